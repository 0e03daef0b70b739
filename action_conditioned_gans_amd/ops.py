"""Operator surface of the hot path: conv2d / deconv2d / batch_norm / lrelu / dna_gather + losses.

Mirrors what the reference's ``ops.py`` and its ``slim.*`` call sites offer (ops.py:19-50,100-120;
models.py:10-21,31-72,80-88), with the same keyword arguments and ``ValueError`` behaviour, but every
function records graph ops (graph.py) that bind to hand-written HIP kernels through the C ABI
(include/acgan_hip.h).  The slim layer contract is kept: ``conv -> (normalizer | +bias) ->
activation``, xavier-uniform weights, zero biases/beta, ``reuse`` sharing by variable name.
Layer-level fusion happens here: BatchNorm + activation is one op, bias + activation is one op.
"""
import contextlib
import ctypes
import functools
import math

import torch

from . import _lib
from . import graph as G
from ._lib import ACG_F32, CONV_DGRAD, CONV_FWD, CONV_WGRAD, SLABS_QUADS, ConvDesc

DNA_KERN_SIZE = 5          # ops.py:12
_ACT_CODE = {None: _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'lrelu': _lib.ACT_LRELU, 'tanh': _lib.ACT_TANH}
_p = G._ptr


# ================================================================================================
# scopes (tf.variable_scope / slim.arg_scope)
# ================================================================================================
_scope_stack = []     # [(name, reuse)]
_arg_stack = []       # [{func: kwargs}]
_seed = [0]
_rng = [None]


def set_random_seed(seed):
    _seed[0] = int(seed)
    _rng[0] = None


def _generator():
    if _rng[0] is None:
        _rng[0] = torch.Generator().manual_seed(_seed[0])
    return _rng[0]


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _scope_stack.append((name, reuse))
    try:
        yield
    finally:
        _scope_stack.pop()


def _scope_name(leaf=None):
    parts = [n for n, _ in _scope_stack if n]
    if leaf:
        parts.append(leaf)
    return '/'.join(parts)


def _scope_reuse():
    for _, r in reversed(_scope_stack):
        if r is not None:
            return bool(r)
    return False


@contextlib.contextmanager
def arg_scope(funcs, **kwargs):
    """slim.arg_scope: default keyword arguments for the listed layer functions."""
    for f in funcs:
        if not getattr(f, '_acg_arg_scoped', False):
            raise ValueError('%s is not decorated with @add_arg_scope' % getattr(f, '__name__', f))
    _arg_stack.append({getattr(f, '_acg_key'): dict(kwargs) for f in funcs})
    try:
        yield
    finally:
        _arg_stack.pop()


def add_arg_scope(fn):
    key = fn.__name__

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        merged = {}
        for frame in _arg_stack:
            merged.update(frame.get(key, {}))
        merged.update(kwargs)
        return fn(*args, **merged)
    wrapper._acg_arg_scoped = True
    wrapper._acg_key = key
    return wrapper


def xavier_initializer():
    """slim.xavier_initializer(uniform=True): U(-l, l), l = sqrt(6/(fan_in+fan_out)), fans = kh*kw*shape[-2|-1]."""
    def init(shape):
        rf = 1
        for s in shape[:-2]:
            rf *= s
        limit = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        return ((torch.rand(shape, generator=_generator(), dtype=torch.float64) * 2 - 1) * limit).float()
    return init


def zeros_initializer():
    return lambda shape: torch.zeros(shape, dtype=torch.float32)


# ================================================================================================
# graph ops
# ================================================================================================
def _new(shape, name, dtype=torch.float32):
    return G.Tensor(G.get_default_graph(), shape, name=name, dtype=dtype)


def act_dtype():
    """Storage type of activation-class tensors in the graph being built (float32, or bfloat16: BASELINE configs 3 / 5)."""
    return G.get_default_graph().act_dtype


def half_mode():
    return act_dtype() == torch.bfloat16


def cpad(c):
    """Channel pitch of an activation with ``c`` channels when it feeds a conv: rounded up to one 16-byte gather."""
    u = G.get_default_graph().cpad
    return -(-c // u) * u


def _new_act(shape, name):
    return _new(shape, name, act_dtype())


def _code(t):
    return _lib.code(t.dtype)


def _code2(a, b):
    return _lib.dtype2(_lib.code(a.dtype), _lib.code(b.dtype))


def _wcopy(w, kind):
    """bf16 pipeline: the bf16 operand copy of filter variable ``w`` [kh,kw,A,B] that a contraction reads - 'rm'
    [taps,A,round8(B)] (conv dgrad, deconv fwd) or 'tr' [taps,B,round8(A)] (conv fwd, deconv dgrad); float32: ``w``."""
    if not half_mode():
        return w
    if getattr(w, 'copies', None) is None:
        g = G.get_default_graph()
        kh, kw, a, b = w.shape
        rm = g.new_state((kh * kw, a, -(-b // 8) * 8), 0.0, w.name + '/bf16_rm', dtype=torch.bfloat16)
        tr = g.new_state((kh * kw, b, -(-a // 8) * 8), 0.0, w.name + '/bf16_tr', dtype=torch.bfloat16)
        w.copies = {'rm': rm, 'tr': tr}
        g.weight_copies.setdefault(w.scope, []).append((w, rm, tr))
    return w.copies[kind]


def prepare_weights_launch(rt, graph, scope):
    """fn(stream) that refreshes every bf16 filter copy of ``scope`` from the float32 master weights in one launch
    (acg_weights_prepare_bf16), or None when the scope has none.  Buffers must be materialised."""
    entries = graph.weight_copies.get(scope) or []
    if not entries:
        return None
    lists = []
    for lo in range(0, len(entries), _lib.PREP_MAX):
        chunk = entries[lo:lo + _lib.PREP_MAX]
        pl = _lib.PrepList()
        for i, (w, rm, tr) in enumerate(chunk):
            kh, kw, a, b = w.shape
            pl.src[i], pl.rm[i], pl.tr[i] = w.buf.data_ptr(), rm.buf.data_ptr(), tr.buf.data_ptr()
            pl.taps[i], pl.a[i], pl.b[i] = kh * kw, a, b
        lists.append((pl, len(chunk)))
    fn = rt.lib.weights_prepare_bf16

    def launch(s):
        for pl, n in lists:
            fn(ctypes.byref(pl), n, s)
    launch._keep = lists
    return launch


def _check_nhwc(x, what):
    if not isinstance(x, G.Tensor) or len(x.shape) != 4:
        raise ValueError('%s expects a rank-4 NHWC tensor, got %r' % (what, x))


def _desc(lib_free_args):
    """Build an acg_conv_desc in Python (same arithmetic as acg_conv_desc_init)."""
    batch, h, w, c, kh, kw, cout, stride, padding = lib_free_args[:9]
    d = ConvDesc()
    d.in_pitch = lib_free_args[9] if len(lib_free_args) > 9 else 0
    d.out_pitch = lib_free_args[10] if len(lib_free_args) > 10 else 0
    d.batch, d.in_h, d.in_w, d.in_c, d.out_c, d.kh, d.kw = batch, h, w, c, cout, kh, kw
    d.stride_h = d.stride_w = stride
    if padding == 'SAME':
        d.out_h, d.out_w = -(-h // stride), -(-w // stride)
        d.pad_top = max((d.out_h - 1) * stride + kh - h, 0) // 2
        d.pad_left = max((d.out_w - 1) * stride + kw - w, 0) // 2
    elif padding == 'VALID':
        if h < kh or w < kw:
            raise ValueError('VALID convolution: kernel %dx%d larger than input %dx%d' % (kh, kw, h, w))
        d.out_h, d.out_w = (h - kh) // stride + 1, (w - kw) // stride + 1
        d.pad_top = d.pad_left = 0
    else:
        raise ValueError('unexpected padding argument')
    return d


class _ConvBase(G.Op):
    """Shared binder: one C-ABI call with (a, b, out, desc, dtype, workspace, n, stream)."""
    which = CONV_FWD

    def _bind(self, rt, entry, a, b, out, accumulate=None):
        lib, d = rt.lib, self.desc
        if a.dtype != (torch.bfloat16 if rt.conv_dtype == _lib.ACG_BF16 else torch.float32):
            raise TypeError('%s: %s operand %r in a %s session' % (self.name, a.dtype, a, 'bf16' if rt.conv_dtype else 'f32'))
        ws, n = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), self.which, rt.conv_dtype))
        self._keep = (ws, d)
        fn = getattr(lib, entry)
        pa, pb, po, dref, pws, dt = _p(a.buf), _p(b.buf), _p(out.buf), ctypes.byref(d), _p(ws), rt.conv_dtype
        if accumulate is None:
            return lambda s: fn(pa, pb, po, dref, dt, pws, n, s)
        return lambda s: fn(pa, pb, po, accumulate, dref, dt, pws, n, s)


class Conv2dOp(_ConvBase):
    """y = conv(x, w) (transposed=False) or TF conv2d_transpose(x, w) (transposed=True; desc is the adjoint conv)."""

    def __init__(self, x, w, desc, transposed, name, out_f32=False):
        g = G.get_default_graph()
        self.desc, self.transposed = desc, transposed
        self.which = CONV_DGRAD if transposed else CONV_FWD
        c = desc.in_c if transposed else desc.out_c
        shape = (desc.batch, desc.in_h, desc.in_w) if transposed else (desc.batch, desc.out_h, desc.out_w)
        cphys = cpad(c) if half_mode() else c          # bf16 activations live at the channel pitch round8(C)
        # out_f32 (bf16 graphs): the result stays float32, at the same pitch - the input of a head BatchNorm (_layer)
        self.out_f32 = bool(out_f32) and half_mode()
        y = _new(shape + (cphys,), name + ':0') if self.out_f32 else _new_act(shape + (cphys,), name + ':0')
        if cphys != c:
            y.valid_c = c
        self.wop = _wcopy(w, 'rm' if transposed else 'tr')
        self.extras = [self.wop]
        self._part = {}
        super().__init__(g, name, [x, w], [y])

    bn_consumer = None      # the layer's BnActOp (set by _layer): takes its statistics, or the split-K slabs, from this op
    bias_consumer = None    # the layer's BiasActOp (set by _layer): bias + activation move into this op's epilogue where they can
    _fused_bias = False     # this program: the epilogue wrote the BiasActOp's output, that op launches nothing
    _slab = None            # (workspace, splits) while this program's BatchNorm sums the slabs itself
    _stats = None           # (partials, blocks per group, rows per block, rows per run) while this program's BatchNorm takes its statistics from the epilogue

    def bind(self, rt):
        x, w = self.inputs
        self._slab = self._stats = None
        self._fused_bias = False
        self.absorbed = ()          # ops whose launch this op's launch replaces in the program being compiled
        bc = self.bias_consumer
        if (bc is not None and rt.epilogue_bias and id(bc) in rt.program_ops and self.transposed and not self.out_f32 and bc.has_bias
                and bc.outputs[0].dtype == torch.float32):
            # models.py:20-21: tanh(conv2d_transpose(x) + b) - bias and activation in the deconv's epilogue, straight into the
            # BiasActOp's output tensor (acg_deconv2d_fwd_bias_act), where the planner runs the layer unsplit on its 128x32 tile
            lib, dt = rt.lib, rt.conv_dtype
            d = ConvDesc()
            ctypes.memmove(ctypes.byref(d), ctypes.byref(self.desc), ctypes.sizeof(ConvDesc))
            d.in_pitch = bc.yp if bc.yp != d.in_c else 0
            if dt == _lib.ACG_BF16:
                d.in_pitch = 0
            if lib.deconv2d_fwd_bias_act_ok(ctypes.byref(d), dt):
                d.in_pitch = bc.yp if bc.yp != d.in_c else 0
                self._keep, self._fused_bias = (d,), True
                self.absorbed = (bc,)
                args = (_p(x.buf), _p(self.wop.buf), _p(bc.inputs[1].buf), _p(bc.outputs[0].buf), ctypes.byref(d), _ACT_CODE[bc.act], bc.leak, dt)
                fn = lib.deconv2d_fwd_bias_act
                return lambda s: fn(*args, s)
        bn = self.bn_consumer
        if self.out_f32:
            lib, d = rt.lib, self.desc
            dt = _lib.dtype2(_lib.ACG_BF16, _lib.ACG_F32)
            ws, n = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), self.which, dt))
            self._keep = (ws, d)
            fn = lib.deconv2d_fwd if self.transposed else lib.conv2d_fwd
            args = (_p(x.buf), _p(self.wop.buf), _p(self.outputs[0].buf), ctypes.byref(d), dt, _p(ws), n)
            return lambda s: fn(*args, s)
        if bn is not None and id(bn) in rt.program_ops and rt.epilogue_stats and type(bn) is BnActOp:
            # followed by its BatchNorm in this program and not split: the epilogue leaves per-tile channel sums, the
            # BatchNorm runs its apply pass alone (acg_(de)conv2d_fwd_stats -> acg_bn_act_fwd_partials): one launch and one
            # read of the activation less
            lib, d, dt = rt.lib, self.desc, rt.conv_dtype
            brows, rrows = ctypes.c_int32(0), ctypes.c_int32(0)
            nblk = lib.conv2d_stats_layout(ctypes.byref(d), self.which, dt, bn.groups, ctypes.byref(brows), ctypes.byref(rrows))
            if nblk > 0:
                size = bn.groups * nblk * 2 * bn.c
                part = self._part.get((rt.device, size))     # one buffer for every program this op is compiled into
                if part is None:
                    part = self._part[(rt.device, size)] = torch.zeros(size, dtype=torch.float32, device=rt.device)
                ws, n = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), self.which, dt))
                self._keep, self._stats = (ws, d, part), (part, nblk, brows.value, rrows.value)
                fn = lib.deconv2d_fwd_stats if self.transposed else lib.conv2d_fwd_stats
                args = (_p(x.buf), _p(self.wop.buf), _p(self.outputs[0].buf), ctypes.byref(d), dt, _p(ws), n, _p(part), bn.groups)
                return lambda s: fn(*args, s)
        if bn is not None and id(bn) in rt.program_ops and rt.slab_handoff:
            # split over K and followed by its BatchNorm in this program: leave the partial slabs, the BatchNorm kernel
            # sums them as it loads (acg_bn_act_fwd_slabs) - one launch less
            lib, d = rt.lib, self.desc
            splits = lib.conv2d_splits(ctypes.byref(d), self.which, rt.conv_dtype)
            layout = bn.slab_layout(rt, splits, False)
            if layout >= 0 and lib.conv2d_slab_layouts(ctypes.byref(d), self.which, rt.conv_dtype) >> layout & 1:
                ws, n = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), self.which, rt.conv_dtype))
                self._keep, self._slab = (ws, d), (ws, splits, layout)
                fn = lib.deconv2d_fwd_slabs if self.transposed else lib.conv2d_fwd_slabs
                pa, pb, dref, pws, dt = _p(x.buf), _p(self.wop.buf), ctypes.byref(d), _p(ws), rt.conv_dtype
                return lambda s: fn(pa, pb, dref, dt, layout, pws, n, s)
        return self._bind(rt, 'deconv2d_fwd' if self.transposed else 'conv2d_fwd', x, self.wop, self.outputs[0])

    def grad(self, gouts, needs, ctx):
        x, w = self.inputs
        dy = gouts[0]
        dx = None
        if needs[0]:
            dx = ConvDgradOp(dy, w, x, self.desc, self.transposed, self.name + '/dgrad').outputs[0]
            dx.valid_c = x.valid_c
        if needs[1] and ctx.wants(w):
            dst, acc = ctx.slot(w)
            wg = ConvWgradOp(x, dy, dst, acc, self.desc, self.transposed, self.name + '/wgrad')
            ctx.wrote(w, wg)
            if dx is not None:          # both consume dy: candidates for ONE launch (Session.pair_bwd, graph._compile)
                dx.op.pair_w = wg
        return [dx, None]


class ConvDgradOp(_ConvBase):
    pair_w = None          # the layer's ConvWgradOp, when both gradients are built
    pair_active = False    # set per compiled program: this op launches both (acg_(de)conv2d_bwd_pair)

    def __init__(self, dy, w, x, desc, transposed, name):
        self.desc, self.transposed = desc, transposed
        self.which = CONV_FWD if transposed else CONV_DGRAD
        self.wop = _wcopy(w, 'tr' if transposed else 'rm')
        self.extras = [self.wop]
        super().__init__(G.get_default_graph(), name, [dy, w], [_new(x.shape, name + ':0', x.dtype)])

    bn_bwd_consumer = None  # the BnActBwdOp that reads this op's output as its dy (set in BnActOp.grad)
    _slab = None

    def limit_channels(self, c):
        """Only the first ``c`` channels of this input gradient are read by anyone (the rest belong to tiled action inputs,
        ConcatActionsOp.grad): acg_conv_desc dgrad_c / adj_dgrad_c - e.g. 128 of d/conv3's 138 columns, two 64-column tiles
        instead of three."""
        if self.transposed:
            self.desc.adj_dgrad_c = int(c)
        else:
            self.desc.dgrad_c = int(c)

    def bind(self, rt):
        dy, w = self.inputs
        w = self.wop
        lib, d, dt = rt.lib, self.desc, rt.conv_dtype
        self._slab = None
        bn = self.bn_bwd_consumer
        layout = -1
        if bn is not None and id(bn) in rt.program_ops and rt.slab_handoff:
            layout = bn.fwd.slab_layout(rt, lib.conv2d_splits(ctypes.byref(d), self.which, dt), True, dy=self.outputs[0])
            if layout >= 0 and not lib.conv2d_slab_layouts(ctypes.byref(d), self.which, dt) >> layout & 1:
                layout = -1
        hand_off = layout >= 0
        if not self.pair_active:
            if hand_off:     # the consuming BatchNorm backward sums the slabs (acg_bn_act_bwd_slabs)
                ws, n = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), self.which, dt))
                self._keep, self._slab = (ws, d), (ws, lib.conv2d_splits(ctypes.byref(d), self.which, dt), layout)
                fn = lib.deconv2d_dgrad_slabs if self.transposed else lib.conv2d_dgrad_slabs
                pa, pb, dref, pws = _p(dy.buf), _p(w.buf), ctypes.byref(d), _p(ws)
                return lambda s: fn(pa, pb, dref, dt, layout, pws, n, s)
            return self._bind(rt, 'deconv2d_dgrad' if self.transposed else 'conv2d_dgrad', dy, w, self.outputs[0])
        # ONE launch for dx and dw: the two contractions share the CUs instead of running one grid after the other
        wg = self.pair_w
        x = wg.inputs[0]
        wsd, nd = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), self.which, dt))
        wsw, nw = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), CONV_WGRAD, dt))
        if hand_off:
            self._slab = (wsd, lib.conv2d_splits(ctypes.byref(d), self.which, dt), layout)
        slabs = 0
        if wg.deferred_to is not None:
            splits = lib.conv2d_splits(ctypes.byref(d), CONV_WGRAD, dt)
            if splits > 1:
                slabs = 1
                wg.deferred_to.pending.append((wsw, wg.outputs[0], splits, wg.accumulate))
        self._keep = (wsd, wsw, d)
        fn = lib.deconv2d_bwd_pair if self.transposed else lib.conv2d_bwd_pair
        args = (_p(dy.buf), _p(w.buf), _p(x.buf), _p(self.outputs[0].buf), None if slabs else _p(wg.outputs[0].buf), wg.accumulate,
                ctypes.byref(d), dt, _p(wsd), nd, _p(wsw), nw, slabs | (2 if hand_off else 0) | (4 if layout == SLABS_QUADS else 0))
        return lambda s: fn(*args, s)


class ConvWgradOp(_ConvBase):
    which = CONV_WGRAD

    def __init__(self, x, dy, dst, accumulate, desc, transposed, name):
        self.desc, self.transposed, self.accumulate = desc, transposed, float(accumulate)
        self.deferred_to = None     # a WgradReduceOp: this op leaves its split-K slabs for that op's single launch
        super().__init__(G.get_default_graph(), name, [x, dy], [dst])

    paired = False          # set per compiled program: the layer's ConvDgradOp launches this contraction too

    def bind(self, rt):
        if self.paired:
            return None
        x, dy = self.inputs
        lib, d = rt.lib, self.desc
        if self.deferred_to is not None:
            splits = lib.conv2d_splits(ctypes.byref(d), CONV_WGRAD, rt.conv_dtype)
            if splits > 1:
                ws, n = rt.workspace(lib.conv2d_workspace_bytes(ctypes.byref(d), CONV_WGRAD, rt.conv_dtype))
                self._keep = (ws, d)
                self.deferred_to.pending.append((ws, self.outputs[0], splits, self.accumulate))
                fn = lib.deconv2d_wgrad_slabs if self.transposed else lib.conv2d_wgrad_slabs
                pa, pb, dref, pws, dt = _p(x.buf), _p(dy.buf), ctypes.byref(d), _p(ws), rt.conv_dtype
                return lambda s: fn(pa, pb, dref, dt, pws, n, s)
        return self._bind(rt, 'deconv2d_wgrad' if self.transposed else 'conv2d_wgrad', x, dy, self.outputs[0],
                          accumulate=self.accumulate)


class WgradReduceOp(G.Op):
    """ONE launch that finishes the split-K weight gradients of several layers (acg_splitk_reduce_many): nothing reads
    a weight gradient before the optimizer update (or the all-reduce of its bucket), and a launch costs ~4-5 us in the
    step's HIP graph whatever its size.  Bit-identical to the per-layer reductions it replaces."""

    joins_side = True

    def __init__(self, wgrads, name):
        g = G.get_default_graph()
        super().__init__(g, name, [], [], control_inputs=wgrads)
        self.side_stream = False
        self.index = max(o.index for o in wgrads) + 0.25      # right behind the last of its layers
        self.pending = []                                     # filled by the ConvWgradOps as they are bound
        for o in wgrads:
            o.deferred_to = self

    def bind(self, rt):
        entries, self.pending = self.pending, []
        self._keep = None                                     # (the optimizer looks here for THIS compile's lists: Adam._step_inc_launch)
        self._bound_for = rt.program_ops                      # the compile these lists belong to: a stale _keep of an earlier program never matches
        self.carries_step_inc = False
        if not entries:
            return None                                       # none of the layers is split at these shapes
        lists = []
        for lo in range(0, len(entries), _lib.REDUCE_MAX):
            chunk = entries[lo:lo + _lib.REDUCE_MAX]
            rl = _lib.ReduceList()
            for i, (ws, dst, splits, acc) in enumerate(chunk):
                rl.slabs[i], rl.out[i], rl.numel[i], rl.splits[i], rl.accumulate[i] = ws.data_ptr(), dst.buf.data_ptr(), dst.numel, splits, acc
            lists.append((rl, len(chunk)))
        self._keep = lists
        fn = rt.lib.splitk_reduce_many

        def launch(s):
            for rl, n in lists:
                fn(ctypes.byref(rl), n, s)
        return launch


class BnActOp(G.Op):
    """slim.batch_norm (batch statistics, beta only) fused with the layer activation."""
    conv_producer = None     # the layer's Conv2dOp (set by _layer): source of the split-K hand-off

    def __init__(self, x, beta, act, leak, eps, groups, name):
        g = G.get_default_graph()
        self.xp = x.shape[-1]                         # channel pitch of x (bf16 conv outputs: round8(C))
        c = x.valid_c or self.xp
        self.act, self.leak, self.eps, self.groups = act, float(leak), float(eps), int(groups)
        self.rows, self.c = x.numel // self.xp, c
        if self.rows % self.groups:
            raise ValueError('batch_norm: %d rows not divisible by %d groups' % (self.rows, self.groups))
        self.mean, self.rstd = _new((groups * c,), name + '/mean'), _new((groups * c,), name + '/rstd')
        if act is None and half_mode():
            # a layer without activation is a head (d/conv6: the logits the losses read): dense float32 output
            y = _new(x.shape[:-1] + (c,), name + ':0')
        else:
            y = _new(x.shape, name + ':0', x.dtype)
            y.valid_c = x.valid_c
        self.yp = y.shape[-1]
        super().__init__(g, name, [x, beta], [y, self.mean, self.rstd])

    # the `flags` argument of this layer's BatchNorm launches in the program being compiled, forward and backward; set per
    # program by Session._compile (an op that runs beside a multi-rank collective takes the two-launch kernels), None = rt.bn_flags
    flags_fwd = flags_bwd = None

    def launch_flags(self, rt, backward):
        own = self.flags_bwd if backward else self.flags_fwd
        return rt.bn_flags if own is None else own

    def slab_layout(self, rt, splits, backward, dy=None):
        """The slab layout this BatchNorm (forward, or its backward reading `dy`) takes from a producer split `splits` ways
        in the program being compiled; -1: none, the producer runs its own reduction (acg_bn_slabs_layout)."""
        if not 1 < splits <= rt.slab_handoff:
            return -1
        code = _code2(self.inputs[0], dy if backward else self.outputs[0])
        layout = rt.lib.bn_slabs_layout(self.rows, self.c, self.xp, self.yp, self.groups, code, 1 if backward else 0, self.launch_flags(rt, backward))
        return layout if (layout == SLABS_QUADS or rt.slab_rows) else -1

    def bind(self, rt):
        lib = rt.lib
        ws, n = rt.state_workspace(lib.bn_workspace_bytes(self.rows, self.c, self.groups))
        self._keep = ws
        x, beta = self.inputs
        y, mean, rstd = self.outputs
        src = self.conv_producer
        stats = src._stats if (src is not None and id(src) in rt.program_ops) else None
        if stats is not None:     # the conv's epilogue left the per-tile channel sums: the apply pass alone
            part, nblk, brows, rrows = stats
            args = (_p(x.buf), _p(beta.buf), _p(part), nblk, brows, rrows, _p(y.buf), _p(mean.buf), _p(rstd.buf), self.rows, self.c, self.xp, self.yp,
                    self.groups, self.eps, _ACT_CODE[self.act], self.leak, _code2(x, y))
            fn = lib.bn_act_fwd_partials
            return lambda s: fn(*args, s)
        slab = src._slab if (src is not None and id(src) in rt.program_ops) else None
        if slab is not None:      # the conv left its split-K slabs: sum them here and write x for the backward pass
            sws, splits, layout = slab
            args = (_p(sws), splits, _p(x.buf), _p(beta.buf), _p(y.buf), _p(mean.buf), _p(rstd.buf), self.rows, self.c, self.xp, self.yp,
                    self.groups, self.eps, _ACT_CODE[self.act], self.leak, _code2(x, y), layout, self.launch_flags(rt, False), _p(ws), n)
            fn = lib.bn_act_fwd_slabs
            return lambda s: fn(*args, s)
        args = (_p(x.buf), _p(beta.buf), _p(y.buf), _p(mean.buf), _p(rstd.buf), self.rows, self.c, self.xp, self.yp, self.groups,
                self.eps, _ACT_CODE[self.act], self.leak, _code2(x, y), self.launch_flags(rt, False), _p(ws), n)
        fn = lib.bn_act_fwd
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        x, beta = self.inputs
        dst, acc = ctx.slot(beta) if (needs[1] and ctx.wants(beta)) else (None, 0.0)
        op = BnActBwdOp(self, gouts[0], dst, acc, self.name + '/bwd')
        if dst is not None:
            ctx.wrote(beta, op)
        if isinstance(gouts[0].op, ConvDgradOp) and gouts[0] is gouts[0].op.outputs[0]:
            gouts[0].op.bn_bwd_consumer, op.dy_producer = op, gouts[0].op     # split-K hand-off candidate (backward)
        return [op.outputs[0] if needs[0] else None, None]


class BnActBwdOp(G.Op):
    dy_producer = None

    def __init__(self, fwd, dy, dbeta_dst, accumulate, name):
        g = G.get_default_graph()
        self.fwd, self.accumulate = fwd, float(accumulate)
        x, beta = fwd.inputs
        if dbeta_dst is None:                      # beta frozen in this pass: private scratch slot
            dbeta_dst = _new((fwd.c,), name + '/dbeta_scratch')
        # (a head layer of a bf16 graph reads a float32 conv output, Conv2dOp out_f32: its gradient goes back as bf16)
        dx = _new(x.shape, name + ':0', act_dtype())
        dx.valid_c = x.valid_c
        super().__init__(g, name, [x, dy, beta, fwd.mean, fwd.rstd], [dx, dbeta_dst])

    def bind(self, rt):
        lib, f = rt.lib, self.fwd
        ws, n = rt.state_workspace(lib.bn_workspace_bytes(f.rows, f.c, f.groups))
        self._keep = ws
        x, dy, beta, mean, rstd = self.inputs
        dx, dbeta = self.outputs
        src = self.dy_producer
        slab = src._slab if (src is not None and id(src) in rt.program_ops) else None
        if slab is not None:      # dy arrives as the producing dgrad's split-K slabs
            sws, splits, layout = slab
            args = (_p(x.buf), _p(sws), splits, _p(beta.buf), _p(mean.buf), _p(rstd.buf), _p(dx.buf), _p(dbeta.buf), self.accumulate,
                    f.rows, f.c, f.xp, f.yp, f.groups, _ACT_CODE[f.act], f.leak, _code2(x, dy), layout, f.launch_flags(rt, True), _p(ws), n)
            fn = lib.bn_act_bwd_slabs
            return lambda s: fn(*args, s)
        args = (_p(x.buf), _p(dy.buf), _p(beta.buf), _p(mean.buf), _p(rstd.buf), _p(dx.buf), _p(dbeta.buf), self.accumulate,
                f.rows, f.c, f.xp, f.yp, f.groups, _ACT_CODE[f.act], f.leak, _bn_bwd_code(x, dy, dx), f.launch_flags(rt, True), _p(ws), n)
        fn = lib.bn_act_bwd
        return lambda s: fn(*args, s)


# ---- synchronised BatchNorm (optional data-parallel mode: statistics of the GLOBAL batch, SURVEY 8(e) caveat 1) -------
class BnMomentsOp(G.Op):
    """This rank's per-group mean / biased variance (acg_bn_moments)."""

    def __init__(self, x, groups, name):
        self.xp = x.shape[-1]
        c = x.valid_c or self.xp
        self.rows, self.c, self.groups = x.numel // self.xp, c, int(groups)
        super().__init__(G.get_default_graph(), name, [x], [_new((groups * 2 * c,), name + ':0')])

    def bind(self, rt):
        lib = rt.lib
        # plain scratch, NOT rt.state_workspace: acg_bn_moments writes its partial sums from byte 0 - there is no state header,
        # and Runtime.check_exchange_flags would read a partial sum as a timeout flag
        ws, n = rt.workspace(lib.bn_workspace_bytes(self.rows, self.c, self.groups))
        self._keep = ws
        x = self.inputs[0]
        args = (_p(x.buf), _p(self.outputs[0].buf), self.rows, self.c, self.xp, self.groups, _code(x), _p(ws), n)
        fn = lib.bn_moments
        return lambda s: fn(*args, s)


class BnMomentsAllReduceOp(G.Op):
    """Host op: combines the ranks' (mean, var) into the moments of the global batch (equal row counts per rank):
    mean = avg(mean_r), var = avg(var_r + mean_r^2) - mean^2.  One small all-reduce per BatchNorm layer."""
    host = True

    def __init__(self, moments, groups, c, name):
        self.groups, self.c = groups, c
        super().__init__(G.get_default_graph(), name, [moments], [_new(moments.shape, name + ':0')])

    def bind(self, rt):
        src, dst, g, c = self.inputs[0], self.outputs[0], self.groups, self.c

        def run():
            m = src.buf.view(g, 2, c)
            t = torch.stack([m[:, 0], m[:, 1] + m[:, 0] * m[:, 0]], dim=1).contiguous()
            rt.comm.all_reduce(t)
            t /= rt.world_size
            out = dst.buf.view(g, 2, c)
            out[:, 0] = t[:, 0]
            out[:, 1] = (t[:, 1] - t[:, 0] * t[:, 0]).clamp_(min=0.0)
        return run


class BnApplyMomentsOp(G.Op):
    """y = act((x - mean) * rstd + beta) with GIVEN (global) moments; saves mean / rstd for backward.  Storage types and
    pitches as BnActOp: bf16 activations at the pitch round8(C), a head layer's float32 input and dense float32 output."""

    def __init__(self, x, beta, gmoments, act, leak, eps, groups, name):
        self.xp = x.shape[-1]
        c = x.valid_c or self.xp
        self.act, self.leak, self.eps, self.groups = act, float(leak), float(eps), int(groups)
        self.rows, self.c = x.numel // self.xp, c
        self.mean, self.rstd = _new((groups * c,), name + '/mean'), _new((groups * c,), name + '/rstd')
        if act is None and half_mode():
            y = _new(x.shape[:-1] + (c,), name + ':0')
        else:
            y = _new(x.shape, name + ':0', x.dtype)
            y.valid_c = x.valid_c
        self.yp = y.shape[-1]
        super().__init__(G.get_default_graph(), name, [x, beta, gmoments], [y, self.mean, self.rstd])

    def bind(self, rt):
        x, beta, gm = self.inputs
        y, mean, rstd = self.outputs
        args = (_p(x.buf), _p(beta.buf), _p(gm.buf), _p(y.buf), _p(mean.buf), _p(rstd.buf), self.rows, self.c, self.xp, self.yp,
                self.groups, self.eps, _ACT_CODE[self.act], self.leak, _code2(x, y))
        fn = rt.lib.bn_act_fwd_moments
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        x, beta, _ = self.inputs
        name = self.name + '/bwd'
        sums = BnBwdSumsOp(self, gouts[0], name + '/sums').outputs[0]
        red = BnSumsAllReduceOp(sums, name + '/allreduce')
        dst, acc = ctx.slot(beta) if (needs[1] and ctx.wants(beta)) else (None, 0.0)
        op = BnBwdApplySumsOp(self, gouts[0], red.outputs[0], red.outputs[1], dst, acc, name)
        if dst is not None:
            ctx.wrote(beta, op)
        return [op.outputs[0] if needs[0] else None, None, None]     # the moments' dependence on x is inside dx


def _bn_bwd_code(x, dy, dx):
    """dtype of a BatchNorm backward call: ACG_DTYPE2(x, dy), or the float32-head code (x, dy float32; dx bf16)."""
    if x.dtype == torch.float32 and dx.dtype == torch.bfloat16:
        return _lib.dtype2(_lib.ACG_F32, _lib.ACG_BF16)
    return _code2(x, dy)


class BnBwdSumsOp(G.Op):
    def __init__(self, fwd, dy, name):
        self.fwd = fwd
        x, beta, _ = fwd.inputs
        super().__init__(G.get_default_graph(), name, [x, dy, beta, fwd.mean, fwd.rstd], [_new((fwd.groups * 2 * fwd.c,), name + ':0')])

    def bind(self, rt):
        lib, f = rt.lib, self.fwd
        ws, n = rt.workspace(lib.bn_workspace_bytes(f.rows, f.c, f.groups))      # plain scratch: see BnMomentsOp.bind
        self._keep = ws
        x, dy, beta, mean, rstd = self.inputs
        args = (_p(x.buf), _p(dy.buf), _p(beta.buf), _p(mean.buf), _p(rstd.buf), _p(self.outputs[0].buf), f.rows, f.c, f.xp, f.yp,
                f.groups, _ACT_CODE[f.act], f.leak, _code2(x, dy), _p(ws), n)
        fn = lib.bn_bwd_sums
        return lambda s: fn(*args, s)


class BnSumsAllReduceOp(G.Op):
    """Host op: keeps this rank's sums (its dbeta share) and all-reduces the sums of dp, dp*xhat over the ranks."""
    host = True

    def __init__(self, sums, name):
        super().__init__(G.get_default_graph(), name, [sums], [_new(sums.shape, name + ':global'), _new(sums.shape, name + ':local')])

    def bind(self, rt):
        src, glob, loc = self.inputs[0], self.outputs[0], self.outputs[1]

        def run():
            loc.buf.copy_(src.buf)
            glob.buf.copy_(src.buf)
            rt.comm.all_reduce(glob.buf)
        return run


class BnBwdApplySumsOp(G.Op):
    def __init__(self, fwd, dy, gsums, lsums, dbeta_dst, accumulate, name):
        self.fwd, self.accumulate = fwd, float(accumulate)
        x, beta, _ = fwd.inputs
        if dbeta_dst is None:
            dbeta_dst = _new((fwd.c,), name + '/dbeta_scratch')
        dx = _new(x.shape, name + ':0', act_dtype())
        dx.valid_c = x.valid_c
        super().__init__(G.get_default_graph(), name, [x, dy, beta, fwd.mean, fwd.rstd, gsums, lsums], [dx, dbeta_dst])

    def bind(self, rt):
        f = self.fwd
        x, dy, beta, mean, rstd, gs, ls = self.inputs
        dx, dbeta = self.outputs
        total = (f.rows // f.groups) * rt.world_size
        args = (_p(x.buf), _p(dy.buf), _p(beta.buf), _p(mean.buf), _p(rstd.buf), _p(gs.buf), _p(ls.buf), total, _p(dx.buf),
                _p(dbeta.buf), self.accumulate, f.rows, f.c, f.xp, f.yp, f.groups, _ACT_CODE[f.act], f.leak, _bn_bwd_code(x, dy, dx))
        fn = rt.lib.bn_act_bwd_sums
        return lambda s: fn(*args, s)


class BiasActOp(G.Op):
    """y = act(x + bias); bias None gives the bare activation.  ``head``: the layer's output is what the losses read
    (frame, state): in a bf16 graph it becomes a dense float32 tensor, the conv output's pad channels dropped."""

    def __init__(self, x, bias, act, leak, name, head=False):
        g = G.get_default_graph()
        self.act, self.leak = act, float(leak)
        self.xp = x.shape[-1]                                  # channel pitch of x
        self.c = x.valid_c or x.shape[-1]
        self.rows = x.numel // self.xp
        to_f32 = head and half_mode()
        y = _new(x.shape[:-1] + (self.c,), name + ':0') if to_f32 else _new(x.shape, name + ':0', x.dtype)
        if not to_f32:
            y.valid_c = x.valid_c
        self.yp = y.shape[-1]
        super().__init__(g, name, [x] + ([bias] if bias is not None else []), [y])
        self.has_bias = bias is not None

    conv_producer = None    # the layer's Conv2dOp (set by _layer)

    def bind(self, rt):
        src = self.conv_producer
        if src is not None and id(src) in rt.program_ops and src._fused_bias:
            return None                               # the deconv's epilogue wrote y (Conv2dOp.bind): nothing to launch
        x, y = self.inputs[0], self.outputs[0]
        pb = _p(self.inputs[1].buf) if self.has_bias else None
        args = (_p(x.buf), pb, _p(y.buf), self.rows, self.c, self.xp, self.yp, _ACT_CODE[self.act], self.leak, _code2(x, y))
        fn = rt.lib.bias_act_fwd
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        dy = gouts[0]
        dst, acc = (None, 0.0)
        if self.has_bias and needs[1] and ctx.wants(self.inputs[1]):
            dst, acc = ctx.slot(self.inputs[1])
        want_dx = needs[0]
        x, y = self.inputs[0], self.outputs[0]
        same = x.dtype == y.dtype and x.shape == y.shape       # dx == dy when there is no activation either
        if dst is None and self.act is None and same:
            return [dy if want_dx else None] + ([None] if self.has_bias else [])
        op = BiasActBwdOp(self, dy, dst, acc, want_dx, self.name + '/bwd')
        if dst is not None:
            ctx.wrote(self.inputs[1], op)
        dx = (dy if (self.act is None and same) else op.dx) if want_dx else None
        return [dx] + ([None] if self.has_bias else [])


class BiasActBwdOp(G.Op):
    def __init__(self, fwd, dy, dbias_dst, accumulate, want_dx, name):
        g = G.get_default_graph()
        self.fwd, self.accumulate = fwd, float(accumulate)
        x, y = fwd.inputs[0], fwd.outputs[0]
        same = x.dtype == y.dtype and x.shape == y.shape
        self.dx = _new(x.shape, name + ':0', x.dtype) if (want_dx and (fwd.act is not None or not same)) else None
        if self.dx is not None:
            self.dx.valid_c = x.valid_c
        self.dbias = dbias_dst
        outs = [t for t in (self.dx, self.dbias) if t is not None]
        super().__init__(g, name, [fwd.outputs[0], dy], outs)

    def bind(self, rt):
        lib, f = rt.lib, self.fwd
        ws, n = rt.workspace(lib.bias_workspace_bytes(f.rows, f.c))
        self._keep = ws
        y, dy = self.inputs
        args = (_p(y.buf), _p(dy.buf), _p(self.dx.buf) if self.dx is not None else None,
                _p(self.dbias.buf) if self.dbias is not None else None, self.accumulate, f.rows, f.c, f.xp, f.yp,
                _ACT_CODE[f.act], f.leak, _code2(f.inputs[0], y), _p(ws), n)
        fn = lib.bias_act_bwd
        return lambda s: fn(*args, s)


class DnaOp(G.Op):
    """softmax(logits + bias) and the k x k gather in one kernel; ``bias`` (may be None) is the bias variable of the
    layer that produced the logits, folded in here instead of a pass of its own over the logits tensor."""

    def __init__(self, logits, image, ksize, name, bias=None):
        self.ksize = int(ksize)
        self.has_bias = bias is not None
        super().__init__(G.get_default_graph(), name, [logits, image] + ([bias] if bias is not None else []),
                         [_new(image.shape, name + ':0')])

    second = None       # (ConcatChannelsOp, its output, channel offset): the frame's second home, written by this kernel too

    def bind(self, rt):
        lg, img = self.inputs[:2]
        b, h, w, c = img.shape
        pb = _p(self.inputs[2].buf) if self.has_bias else None
        p2, pitch2, off2, dt2 = None, 0, 0, 0
        if self.second is not None and id(self.second[0]) in rt.program_ops:      # only where someone reads that tensor
            _, y2, off2 = self.second
            p2, pitch2, dt2 = _p(y2.buf), y2.shape[-1], _code(y2)
        args = (_p(lg.buf), pb, _p(img.buf), _p(self.outputs[0].buf), p2, pitch2, off2, dt2, b, h, w, c, self.ksize, _code(lg))
        fn = rt.lib.dna_fwd
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        if needs[1]:
            raise NotImplementedError('dna_gather: gradient w.r.t. the image is not part of the hot path '
                                      '(the image is a network input, train.py:53-54)')
        dst, acc = (None, 0.0)
        if self.has_bias and needs[2] and ctx.wants(self.inputs[2]):
            dst, acc = ctx.slot(self.inputs[2])
        if not needs[0] and dst is None:
            return [None] * len(self.inputs)
        # dout = (gradient of the frame losses) + (the frame channels of d(discriminator input)): when that sum is an AddOp over
        # a SliceOp of the pitched gradient, the kernel reads the window itself (acg_dna_bwd dout2) and the slice and add
        # launches drop out of every program (nothing else reads them)
        dout, dout2 = gouts[0], None
        if isinstance(dout.op, AddOp) and dout is dout.op.outputs[0]:
            parts = list(dout.op.inputs)
            for k, t in enumerate(parts):
                if isinstance(t.op, SliceOp) and t is t.op.outputs[0] and t.op.c_dst == self.outputs[0].shape[-1] and parts[1 - k].dtype == torch.float32:
                    dout, dout2 = parts[1 - k], (t.op.inputs[0], t.op.c_off)
                    break
        op = DnaBwdOp(self, dout, dst, acc, self.name + '/bwd', dout2=dout2)
        if dst is not None:
            ctx.wrote(self.inputs[2], op)
        return [op.outputs[0] if needs[0] else None, None] + ([None] if self.has_bias else [])


class DnaBwdOp(G.Op):
    def __init__(self, fwd, dout, dbias_dst, accumulate, name, dout2=None):
        self.fwd, self.accumulate, self.dbias = fwd, float(accumulate), dbias_dst
        self.dout2 = dout2                      # (pitched gradient tensor, channel offset) added to dout by the kernel
        lg, img = fwd.inputs[:2]
        dlg = _new(lg.shape, name + ':0', lg.dtype)
        dlg.valid_c = lg.valid_c
        super().__init__(G.get_default_graph(), name, [lg, img, dout] + list(fwd.inputs[2:]) + ([dout2[0]] if dout2 is not None else []),
                         [dlg] + ([dbias_dst] if dbias_dst is not None else []))

    def bind(self, rt):
        lg, img, dout = self.inputs[:3]
        b, h, w, c = img.shape
        pb = _p(self.inputs[3].buf) if self.fwd.has_bias else None
        ws, n = rt.workspace(rt.lib.dna_workspace_bytes(b, h, w, self.fwd.ksize)) if self.dbias is not None else (None, 0)
        self._keep = ws
        p2, pitch2, off2, dt2 = None, 0, 0, 0
        if self.dout2 is not None:
            t2, off2 = self.dout2
            p2, pitch2, dt2 = _p(t2.buf), t2.shape[-1], _code(t2)
        args = (_p(lg.buf), pb, _p(img.buf), _p(dout.buf), p2, pitch2, off2, dt2, _p(self.outputs[0].buf),
                _p(self.dbias.buf) if self.dbias is not None else None,
                self.accumulate, b, h, w, c, self.fwd.ksize, _code(lg), _p(ws) if ws is not None else None, n)
        fn = rt.lib.dna_bwd
        return lambda s: fn(*args, s)


class CdnaOp(G.Op):
    """cdna_transformation after its fully-connected layer (reference ops.py:77-98): normalise the per-sample kernels
    and apply them as a depthwise SAME correlation.  Outputs: the M transformed images (windows of one [M,B,H,W,C]
    buffer the kernel writes in one launch), then the buffer itself and the normalised kernels (saved for backward)."""

    def __init__(self, params, image, masks, ksize, relu_shift, name):
        b, h, w, c = image.shape
        self.masks, self.ksize, self.relu_shift = int(masks), int(ksize), float(relu_shift)
        self.whole = _new((self.masks, b, h, w, c), name + ':whole')
        self.kern_norm = _new((b, self.ksize * self.ksize * self.masks), name + ':kern_norm')
        plane = b * h * w * c
        pieces = [self.whole.view(j * plane, (b, h, w, c), name='%s:%d' % (name, j)) for j in range(self.masks)]
        super().__init__(G.get_default_graph(), name, [params, image], pieces + [self.whole, self.kern_norm])

    def bind(self, rt):
        par, img = self.inputs
        b, h, w, c = img.shape
        args = (_p(par.buf), _p(img.buf), _p(self.whole.buf), _p(self.kern_norm.buf), b, h, w, c, self.masks, self.ksize,
                self.relu_shift, ACG_F32)
        fn = rt.lib.cdna_fwd
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        g = CdnaBwdOp(self, gouts[:self.masks], needs[1], self.name + '/bwd')
        return [g.outputs[0] if needs[0] else None, g.outputs[1] if needs[1] else None]


class CdnaBwdOp(G.Op):
    """Packs the M piece gradients into one [M,B,H,W,C] buffer (pieces nobody differentiated stay zero) and runs the
    backward kernels: d params through the normalisation and the relu, d image on request."""

    def __init__(self, fwd, gpieces, want_dimage, name):
        self.fwd, self.want_dimage = fwd, bool(want_dimage)
        par, img = fwd.inputs
        g = G.get_default_graph()
        self.packed = g.new_state(fwd.whole.shape, 0.0, name + '/packed')
        self.present = [(j, t) for j, t in enumerate(gpieces) if t is not None]
        super().__init__(g, name, [par, img, fwd.kern_norm] + [t for _, t in self.present],
                         [_new(par.shape, name + ':dparams'), _new(img.shape, name + ':dimage')])

    def bind(self, rt):
        par, img, kn = self.inputs[:3]
        b, h, w, c = img.shape
        m, k = self.fwd.masks, self.fwd.ksize
        plane = b * h * w * c
        nbytes = rt.lib.cdna_workspace_bytes(b, h, w, c, m, k)
        ws, nbytes = rt.workspace(nbytes)
        copies = [(_p(t.buf), ctypes.c_void_p(self.packed.buf.data_ptr() + 4 * j * plane)) for j, t in self.present]
        dimg = _p(self.outputs[1].buf) if self.want_dimage else None
        args = (_p(par.buf), _p(kn.buf), _p(img.buf), _p(self.packed.buf), _p(self.outputs[0].buf), dimg, b, h, w, c, m, k,
                self.fwd.relu_shift, ACG_F32, _p(ws), nbytes)
        copy, bwd = rt.lib.slice_channels, rt.lib.cdna_bwd

        def launch(s):
            for src, dst in copies:
                copy(src, dst, 0.0, b * h * w, c, 0, c, ACG_F32, s)
            bwd(*args, s)
        return launch


DGRAD_CHANNEL_LIMIT = True     # A/B switch (bench.py --no-dgrad-limit): input gradients skip the columns of tiled action channels


class ConcatActionsOp(G.Op):
    """tf.tile([B,1,1,A] -> [B,h,w,A]) + tf.concat(axis=3) in one pass (train.py:48-50; models.py:16,38,84).
    The result is stored at a channel pitch rounded up to 4 (138 -> 140, 266 -> 268; zero pad channels) so that
    the consuming conv / deconv gathers it with 16-byte loads."""

    def __init__(self, x, actions, name):
        b, h, w = x.shape[:3]
        c = x.valid_c or x.shape[3]
        if len(actions.shape) != 2 or actions.shape[0] != b:
            raise ValueError('concat_actions: actions must be [batch, A], got %s' % (actions.shape,))
        csum = c + actions.shape[1]
        self.pitch, self.c = cpad(csum), c
        y = _new((b, h, w, self.pitch), name + ':0', x.dtype)
        if self.pitch != csum:
            y.valid_c = csum
        # Producer-side concat (models.py:16,38,84): when x is the output of a fused BatchNorm + activation, that kernel
        # writes its result straight into the concatenated tensor (its y pitch becomes this tensor's pitch), this op only
        # adds the tiled action channels, and backward hands the gradient of the concatenated tensor to the BatchNorm
        # backward as it is (dy at this pitch) - no copy of the feature map in either direction.
        self.in_place = isinstance(x.op, BnActOp) and x is x.op.outputs[0] and x.view_of is None and x.dtype == y.dtype
        g = G.get_default_graph()
        self.fed_inputs = []
        if self.in_place:
            x.op.yp = self.pitch
            x.view_of, x.shape, x.valid_c = (y, 0), y.shape, c
            # ... and when the actions are a fed placeholder (possibly repeated for a joined batch, repeat_batch), the
            # feed copy tiles them into the action channels (Graph.add_feed_alias): then this op launches nothing
            ph = self._fed_source(actions)
            if ph is not None:
                g.add_feed_alias(ph, y, c, tile=(h * w, ph.shape[0]))
                self.fed_inputs = [actions]
        super().__init__(g, name, [x, actions], [y])

    @staticmethod
    def _fed_source(actions):
        """The float32 [B, A] placeholder behind ``actions``: the tensor itself, or repeat_batch(placeholder, 2)."""
        t = actions
        while t.alias_of is not None:
            t = t.alias_of
        if isinstance(t, G.Placeholder):
            return t if (t.dtype == torch.float32 and len(t.shape) == 2 and t.shape[1] == actions.shape[1]) else None
        op = t.op
        if isinstance(op, ConcatChannelsOp) and op.inputs[0] is op.inputs[1] and t is op.outputs[0]:
            src = op.inputs[0]
            while src.alias_of is not None:
                src = src.alias_of
            if (isinstance(src, G.Placeholder) and src.dtype == torch.float32 and len(src.shape) == 2
                    and src.shape[1] == actions.shape[1] and src.shape[0] * 2 == actions.shape[0]):
                return src
        return None

    def bind(self, rt):
        if self.fed_inputs:
            return None
        x, a = self.inputs
        b, h, w = x.shape[:3]
        px = None if self.in_place else _p(x.buf)
        args = (px, _p(a.buf), _p(self.outputs[0].buf), b, h * w, self.c, a.shape[1], self.pitch, _code(self.outputs[0]))
        fn = rt.lib.concat_actions_fwd
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        if needs[1]:
            raise NotImplementedError('concat_actions: actions are inputs, no gradient path')
        x = self.inputs[0]
        if not needs[0]:
            return [None, None]
        src = gouts[0].op
        if DGRAD_CHANNEL_LIMIT and isinstance(src, ConvDgradOp) and gouts[0] is src.outputs[0]:
            src.limit_channels(self.c)      # the action channels are inputs: their gradient columns need not be computed
        if self.in_place:      # d(concat) IS dy of the BatchNorm, read at the concat pitch (a window, not a copy)
            g = gouts[0].view(0, gouts[0].shape, name=self.name + '/bwd:view')
            g.valid_c = self.c
            return [g, None]
        return [SliceOp(gouts[0], 0, x.shape[-1], x.shape, self.name + '/bwd', x.dtype).outputs[0], None]


class ConcatChannelsOp(G.Op):
    """a ++ b on the channel axis; ``pitch`` > ca+cb stores the result with zero pad channels (``valid_c`` set);
    ``act``: the result is an activation (the conv-facing discriminator input: bf16 in a bf16 graph)."""

    def __init__(self, a, b, name, out=None, pitch=0, act=False):
        if a.shape[:-1] != b.shape[:-1]:
            raise ValueError('concat: leading dimensions differ: %s vs %s' % (a.shape, b.shape))
        if a.valid_c or b.valid_c:
            raise ValueError('concat: channel-padded inputs are not supported')
        csum = a.shape[-1] + b.shape[-1]
        shape = a.shape[:-1] + (pitch or csum,)
        if pitch and pitch < csum:
            raise ValueError('concat: pitch %d smaller than %d channels' % (pitch, csum))
        if out is not None and out.shape != shape:
            raise ValueError('concat: out has shape %s, expected %s' % (out.shape, shape))
        self.pitch = pitch if pitch and pitch != csum else 0
        if a.dtype != b.dtype:
            raise ValueError('concat: %s vs %s' % (a.dtype, b.dtype))
        y = out if out is not None else _new(shape, name + ':0', act_dtype() if act else a.dtype)
        if self.pitch:
            y.valid_c = csum
        g = G.get_default_graph()
        # channels that come straight from a fed placeholder are written by the feed copy itself (Graph.add_feed_alias):
        # both discriminator inputs start with the fed current frame, the real one is fed frames only (train.py:64,68)
        fed_a = isinstance(a, G.Placeholder) and a.dtype == torch.float32
        fed_b = fed_a and isinstance(b, G.Placeholder) and b.dtype == torch.float32
        self.fed_inputs = ([a] if fed_a else []) + ([b] if fed_b else [])
        # ... and a generated frame that comes out of the DNA kernel is written here by that kernel (acg_dna_fwd out2), which
        # in the 3 + 3 channels at pitch 8 layout stores the whole pixel - its own image pixel included - as one vector
        self.by_producer = fed_a and not fed_b and isinstance(b.op, DnaOp) and b is b.op.outputs[0] and b.op.second is None
        whole_pixel = self.by_producer and a is b.op.inputs[1] and a.shape[-1] == 3 and b.shape[-1] == 3 and shape[-1] == 8
        if fed_a and not whole_pixel:
            g.add_feed_alias(a, y, 0)
        if fed_b:
            g.add_feed_alias(b, y, a.shape[-1])
        super().__init__(g, name, [a, b], [y])
        if self.by_producer:
            b.op.second = (self, y, a.shape[-1])
            b.op.extras = list(getattr(b.op, 'extras', ())) + [y]

    def bind(self, rt):
        a, b = self.inputs
        if len(self.fed_inputs) == 2 or self.by_producer:
            return None                       # nothing left to launch
        pa = None if self.fed_inputs else _p(a.buf)
        args = (pa, _p(b.buf), _p(self.outputs[0].buf), a.numel // a.shape[-1], a.shape[-1], b.shape[-1], self.pitch,
                _code2(b, self.outputs[0]))
        fn = rt.lib.concat_channels_fwd
        return lambda s: fn(*args, s)

    def grad(self, gouts, needs, ctx):
        a, b = self.inputs
        ga = SliceOp(gouts[0], 0, a.shape[-1], a.shape, self.name + '/bwd_a', a.dtype).outputs[0] if needs[0] else None
        gb = SliceOp(gouts[0], a.shape[-1], b.shape[-1], b.shape, self.name + '/bwd_b', b.dtype).outputs[0] if needs[1] else None
        return [ga, gb]


class JoinOp(G.Op):
    """Batch-axis join with no copy: the parts are windows of ``whole`` that their producers write in place."""

    def __init__(self, parts, whole, name):
        super().__init__(G.get_default_graph(), name, list(parts), [whole])

    def bind(self, rt):
        return None

    def grad(self, gouts, needs, ctx):
        off, out = 0, []
        for part, need in zip(self.inputs, needs):
            out.append(gouts[0].view(off, part.shape) if need else None)
            off += part.numel
        return out


def batch_join(producers, part_shape, name='batch_join', act=False):
    """Join equally shaped tensors along the batch axis without a copy.

    ``producers`` are callables ``f(out)`` that build the op writing one part into the window ``out``.
    Returns ``(whole, parts)``.  ``act``: the joined tensor is an activation (graph activation type)."""
    n = len(producers)
    whole = _new((part_shape[0] * n,) + tuple(part_shape[1:]), name + ':0', act_dtype() if act else torch.float32)
    numel = 1
    for d in part_shape:
        numel *= d
    parts = []
    for k, f in enumerate(producers):
        win = whole.view(k * numel, part_shape, name='%s/part%d' % (name, k))
        f(win)
        parts.append(win)
    JoinOp(parts, whole, name)
    whole.valid_c = parts[0].valid_c
    return whole, parts


class SliceOp(G.Op):
    def __init__(self, src, c_off, c_dst, out_shape, name, dtype=None):
        self.c_off, self.c_dst = int(c_off), int(c_dst)
        super().__init__(G.get_default_graph(), name, [src], [_new(out_shape, name + ':0', dtype or src.dtype)])

    def bind(self, rt):
        src = self.inputs[0]
        cs = src.shape[-1]
        args = (_p(src.buf), _p(self.outputs[0].buf), 0.0, src.numel // cs, cs, self.c_off, self.c_dst, _code2(src, self.outputs[0]))
        fn = rt.lib.slice_channels
        return lambda s: fn(*args, s)


class CopyRowsOp(G.Op):
    """dst <- src, two equally shaped [.., C] tensors of one storage type (acg_slice_channels over all C channels).  The
    look-ahead G step moves the generated frames the pair generator pass parked in the spare rows of the discriminator-input
    buffer to where D(fake) reads them (train.Trainer): a 4 MB copy instead of a second DNA launch.  ``dst`` keeps its
    producer: this op is fetched explicitly by the one program that wants it."""

    def __init__(self, src, dst, name):
        if src.shape != dst.shape or src.dtype != dst.dtype:
            raise ValueError('copy_rows: %r vs %r' % (src, dst))
        super().__init__(G.get_default_graph(), name, [src], [dst])

    def bind(self, rt):
        src, dst = self.inputs[0], self.outputs[0]
        cs = src.shape[-1]
        args = (_p(src.buf), _p(dst.buf), 0.0, src.numel // cs, cs, 0, cs, _code2(src, dst))
        fn = rt.lib.slice_channels
        return lambda s: fn(*args, s)


class AddOp(G.Op):
    def __init__(self, a, b, name='grad_add'):
        if a.numel != b.numel or a.dtype != b.dtype:
            raise ValueError('add: %s %s vs %s %s' % (a.shape, a.dtype, b.shape, b.dtype))
        y = _new(a.shape, name + ':0', a.dtype)
        y.valid_c = a.valid_c
        super().__init__(G.get_default_graph(), name, [a, b], [y])

    def bind(self, rt):
        a, b = self.inputs
        args = (_p(a.buf), _p(b.buf), _p(self.outputs[0].buf), a.numel, _code(a))
        fn = rt.lib.add
        return lambda s: fn(*args, s)


def _add(a, b):
    return AddOp(a, b).outputs[0]


# ================================================================================================
# public layer functions
# ================================================================================================
def relu(x, name='relu'):
    return BiasActOp(x, None, 'relu', 0.0, _scope_name(name)).outputs[0]


def lrelu(x, leak=0.2, name='lrelu'):
    """ops.py:22-26:  0.5(1+leak) x + 0.5(1-leak) |x|."""
    return BiasActOp(x, None, 'lrelu', leak, _scope_name(name)).outputs[0]


def tanh(x, name='tanh'):
    return BiasActOp(x, None, 'tanh', 0.0, _scope_name(name)).outputs[0]


relu._acg_act = ('relu', 0.0)
lrelu._acg_act = ('lrelu', 0.2)
tanh._acg_act = ('tanh', 0.0)


def _act_of(fn):
    """Map an activation_fn argument onto a fusable (kind, leak) or None if it is an arbitrary callable."""
    if fn is None:
        return (None, 0.0)
    if isinstance(fn, functools.partial) and getattr(fn.func, '_acg_act', None) and not fn.args:
        kind, leak = fn.func._acg_act
        return (kind, fn.keywords.get('leak', leak))
    return getattr(fn, '_acg_act', None)


@add_arg_scope
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None, is_training=True,
               reuse=None, scope=None, groups=1):
    """slim.batch_norm as the reference uses it: batch moments over (B,H,W), beta only.

    ``groups`` > 1 normalises equal contiguous chunks of the batch independently (several logical
    batches sharing one launch).  The moving averages of slim exist only as never-updated, never-read
    variables in the reference (UPDATE_OPS is never run), so they are not materialised.
    """
    if scale or not center:
        raise ValueError('batch_norm: only center=True, scale=False (the reference configuration) is implemented')
    if not is_training:
        raise ValueError('batch_norm: the reference never leaves training mode (moving averages are never updated)')
    act = _act_of(activation_fn)
    c = inputs.valid_c or inputs.shape[-1]
    with variable_scope(scope or 'BatchNorm', reuse=reuse):
        beta = G.get_default_graph().get_variable(_scope_name('beta'), (c,), zeros_initializer(), _scope_reuse())
        name = _scope_name()
    fused = act if act is not None else (None, 0.0)
    dp = G.get_default_graph().collections.get('data_parallel')
    if dp is not None and getattr(dp, 'sync_bn', False) and dp.active:
        # statistics of the global batch: moments -> all-reduce (host) -> apply (SURVEY 8(e) caveat 1)
        mom = BnMomentsOp(inputs, groups, name + '/moments').outputs[0]
        gmom = BnMomentsAllReduceOp(mom, groups, c, name + '/moments_allreduce').outputs[0]
        out = BnApplyMomentsOp(inputs, beta, gmom, fused[0], fused[1], epsilon, groups, name).outputs[0]
    else:
        out = BnActOp(inputs, beta, fused[0], fused[1], epsilon, groups, name).outputs[0]
    return out if act is not None else activation_fn(out)


def _layer(inputs, num_outputs, kernel_size, stride, padding, activation_fn, normalizer_fn, normalizer_params,
           weights_initializer, biases_initializer, reuse, scope, transposed, default_scope):
    _check_nhwc(inputs, default_scope)
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
    b, h, w, cphys = inputs.shape
    cin = inputs.valid_c or cphys              # padded channel pitch: the filter sees the logical channels only
    pitch = cphys if cin != cphys else 0
    g = G.get_default_graph()
    if half_mode():
        if inputs.dtype != torch.bfloat16 or cphys != cpad(cin):
            raise ValueError('%s: in a bf16 graph a conv layer reads a bf16 activation stored at the channel pitch round8(C) '
                             '(got %s, %d channels at pitch %d)' % (scope or default_scope, inputs.dtype, cin, cphys))
        pitch = 0                              # implied: the bf16 kernels address every activation at round8(C)
    with variable_scope(scope or default_scope, reuse=reuse):
        share = _scope_reuse()
        winit = weights_initializer or xavier_initializer()
        if transposed:
            if padding != 'SAME':
                raise ValueError('conv2d_transpose: only SAME padding is implemented (the reference uses no other)')
            wshape = (kh, kw, num_outputs, cin)      # the deconv input is the adjoint conv's y: its pitch is out_pitch
            desc = _desc((b, h * stride, w * stride, num_outputs, kh, kw, cin, stride, 'SAME', 0, pitch))
        else:
            wshape = (kh, kw, cin, num_outputs)
            desc = _desc((b, h, w, cin, kh, kw, num_outputs, stride, padding, pitch))
        weights = g.get_variable(_scope_name('weights'), wshape, winit, share)
        name = _scope_name()
        # a BatchNorm'd layer without activation is a head (d/conv6, models.py:87-88): in a bf16 graph its conv output stays
        # float32 - the BatchNorm backward of a 1-channel head cancels to a few per cent of its terms, and one bf16 ulp of its
        # input moved the whole discriminator gradient by 4-11 %
        head = normalizer_fn is not None and activation_fn is None
        out = Conv2dOp(inputs, weights, desc, transposed, name + ('/conv2d_transpose' if transposed else '/conv2d'), out_f32=head).outputs[0]
        act = _act_of(activation_fn)
        if normalizer_fn is not None:
            params = dict(normalizer_params or {})
            conv_op = out.op

            def link(t):      # conv -> its BatchNorm: candidates for the split-K hand-off (Conv2dOp.bind)
                if isinstance(t.op, BnActOp) and t.op.inputs[0] is conv_op.outputs[0]:
                    conv_op.bn_consumer, t.op.conv_producer = t.op, conv_op
                return t
            if getattr(normalizer_fn, '_acg_key', None) == 'batch_norm' and act is not None:
                out = link(normalizer_fn(out, activation_fn=activation_fn, **params))      # fused BN + activation
                return out
            out = link(normalizer_fn(out, **params))
        else:
            bias = g.get_variable(_scope_name('biases'), (num_outputs,), biases_initializer or zeros_initializer(), share)
            # a layer without BatchNorm is a head in the reference nets (frame, state, DNA logits): float32 result
            conv_op = out.op
            if act is not None:
                op = BiasActOp(out, bias, act[0], act[1], name + '/bias_act', head=True)
                conv_op.bias_consumer, op.conv_producer = op, conv_op      # bias + activation may move into the deconv's epilogue
                return op.outputs[0]
            out = BiasActOp(out, bias, None, 0.0, name + '/bias', head=True).outputs[0]
        if activation_fn is not None:
            out = BiasActOp(out, None, act[0], act[1], name + '/act').outputs[0] if act is not None else activation_fn(out)
        return out


@add_arg_scope
def conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', activation_fn=relu, normalizer_fn=None,
           normalizer_params=None, weights_initializer=None, biases_initializer=None, reuse=None, scope=None):
    """slim.conv2d (models.py:12-15,34-37,42-51,82-88): NHWC, HWIO weights, TF SAME/VALID geometry."""
    return _layer(inputs, num_outputs, kernel_size, stride, padding, activation_fn, normalizer_fn, normalizer_params,
                  weights_initializer, biases_initializer, reuse, scope, False, 'Conv')


@add_arg_scope
def deconv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', activation_fn=relu, normalizer_fn=None,
             normalizer_params=None, weights_initializer=None, biases_initializer=None, reuse=None, scope=None):
    """slim.conv2d_transpose (models.py:17-21,39-40,53-59): weights [kh,kw,Cout,Cin], output = input*stride."""
    return _layer(inputs, num_outputs, kernel_size, stride, padding, activation_fn, normalizer_fn, normalizer_params,
                  weights_initializer, biases_initializer, reuse, scope, True, 'Conv2d_transpose')


conv2d_transpose = deconv2d


def dna_gather(logits, image, ksize=DNA_KERN_SIZE, name='dna'):
    """models.py:60-72 in one op: softmax over the k*k logits, then the per-pixel k x k gather of ``image``."""
    _check_nhwc(logits, 'dna_gather')
    _check_nhwc(image, 'dna_gather')
    if logits.shape[:3] != image.shape[:3] or (logits.valid_c or logits.shape[3]) != ksize * ksize:
        raise ValueError('dna_gather: logits %s do not match image %s with ksize %d' % (logits.shape, image.shape, ksize))
    if not 1 <= ksize <= 11:
        raise ValueError('dna_gather: ksize outside 1..11')
    if not 1 <= image.shape[3] <= 4:
        raise ValueError('dna_gather: image channels outside 1..4')
    bias = None
    prod = logits.op
    if isinstance(prod, BiasActOp) and prod.act is None and prod.has_bias:
        # models.py:54-72: the logits come from a layer with a bias and neither BatchNorm nor activation - take its
        # pre-bias output and fold the bias into the softmax kernel; the bias op itself is never fetched and gets pruned
        logits, bias = prod.inputs[0], prod.inputs[1]
    if half_mode() != (logits.dtype == torch.bfloat16):
        raise ValueError('dna_gather: logits are %s in a %s graph' % (logits.dtype, act_dtype()))
    return DnaOp(logits, image, ksize, _scope_name(name), bias=bias).outputs[0]


RELU_SHIFT = 1e-12                                                # ops.py:13


def fully_connected(inputs, num_outputs, activation_fn=None, scope=None, reuse=None):
    """slim.layers.fully_connected on [B, F] (ops.py:70-74): a 1x1 convolution over a 1x1 map - weights
    ``scope/weights`` [1,1,F,out] (slim's [F,out] with two unit axes), ``scope/biases`` [out]."""
    if len(inputs.shape) != 2:
        raise ValueError('fully_connected expects [batch, features], got %s' % (inputs.shape,))
    b, f = inputs.shape
    y = conv2d(inputs.reshape((b, 1, 1, f)), num_outputs, [1, 1], stride=1, padding='SAME', activation_fn=activation_fn,
               normalizer_fn=None, scope=scope, reuse=reuse)
    return y.reshape((b, num_outputs))


def cdna_transformation(prev_image, cdna_input, num_masks, color_channels, ksize=DNA_KERN_SIZE, reuse=None):
    """Reference ops.py:52-98: predict ``num_masks`` k x k kernels per sample from ``cdna_input`` [B, F] with a linear
    layer (scope ``cdna_params``), normalise them and transform ``prev_image`` with each; returns the list of
    ``num_masks`` images [B,H,W,C] (the reference's channel split of the c-major depthwise output, kept as written)."""
    _check_nhwc(prev_image, 'cdna_transformation')
    if prev_image.shape[3] != color_channels:
        raise ValueError('cdna_transformation: color_channels %r does not match image %s' % (color_channels, prev_image.shape))
    if len(cdna_input.shape) != 2 or cdna_input.shape[0] != prev_image.shape[0]:
        raise ValueError('cdna_transformation: cdna_input must be [batch, features], got %s' % (cdna_input.shape,))
    if ksize not in (3, 5, 7) or not 1 <= num_masks <= 32 or not 1 <= color_channels <= 4:
        raise ValueError('cdna_transformation: supports ksize 3/5/7, 1..32 masks, 1..4 colour channels')
    if half_mode():
        raise NotImplementedError('cdna_transformation is float32 only (not on the hot path: no reference model calls it)')
    params = fully_connected(cdna_input, ksize * ksize * num_masks, activation_fn=None, scope='cdna_params', reuse=reuse)
    op = CdnaOp(params, prev_image, num_masks, ksize, RELU_SHIFT, _scope_name('cdna'))
    return list(op.outputs[:num_masks])


def concat_actions(x, actions, name='concat_actions'):
    """x [B,h,w,C] ++ actions [B,A] broadcast over (h,w): the tile+concat of train.py:48-50 / models.py:16,38,84."""
    _check_nhwc(x, 'concat_actions')
    return ConcatActionsOp(x, actions, _scope_name(name)).outputs[0]


def concat(values, axis=3, name='concat', out=None, pitch=0, act=False):
    """tf.concat on the channel axis (train.py:64,68); ``out`` lets the result land in a window of a larger tensor,
    ``pitch`` stores it with zero pad channels up to that channel pitch, ``act`` in the graph's activation type."""
    if axis not in (3, -1) or len(values) != 2:
        raise ValueError('concat: only two tensors on the channel axis are supported')
    return ConcatChannelsOp(values[0], values[1], _scope_name(name), out=out, pitch=pitch, act=act).outputs[0]


def repeat_batch(x, times, name='repeat_batch'):
    """tf.tile along the batch axis for small side inputs (the action vector shared by several sub-batches)."""
    if times != 2:
        raise ValueError('repeat_batch: only times=2 is implemented')
    flat = x.reshape((1, x.numel))
    return ConcatChannelsOp(flat, flat, _scope_name(name)).outputs[0].reshape((x.shape[0] * 2,) + x.shape[1:])


def squeeze(x, name=None):
    """tf.squeeze (models.py:74): drop size-1 dimensions (a storage alias, no kernel)."""
    return x.reshape(tuple(s for s in x.shape if s != 1) or (1,), name=name)


# ================================================================================================
# losses (ops.py:19-50,100-120; train.py:72-85)
# ================================================================================================
class Scalar:
    """A loss value: a fixed linear combination of loss-head outputs, kept symbolic so that
    ``minimize`` can seed every head's fused value+gradient kernel with the right weight."""

    def __init__(self, terms):
        self.terms = list(terms)        # [(head Op, output index, weight)]
        self._tensor = None

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        return Scalar(self.terms + other.terms)

    __radd__ = __add__

    def __mul__(self, k):
        return Scalar([(h, i, w * float(k)) for h, i, w in self.terms])

    __rmul__ = __mul__

    def __truediv__(self, k):
        return self * (1.0 / float(k))

    def __neg__(self):
        return self * -1.0

    def tensor(self):
        if self._tensor is None:
            if len(self.terms) > 4:
                raise ValueError('a loss may combine at most 4 heads')
            self._tensor = CombineOp(self.terms).outputs[0]
        return self._tensor


class CombineOp(G.Op):
    def __init__(self, terms):
        self.terms = terms
        ins = [h.outputs[i] for h, i, _ in terms]
        super().__init__(G.get_default_graph(), 'loss_value', ins, [_new((1,), 'loss_value:0')])

    def bind(self, rt):
        args = [_p(self.outputs[0].buf)]
        for k in range(4):
            if k < len(self.terms):
                args += [_p(self.inputs[k].buf), float(self.terms[k][2])]
            else:
                args += [None, 0.0]
        fn = rt.lib.scalar_combine
        return lambda s: fn(*args, s)


class LossHead(G.Op):
    """A loss head: outputs scalar values; ``seed(weights)`` emits the op writing d(sum w_i out_i)/d input."""

    def seed(self, weights, needs):
        raise NotImplementedError


class FrameLossOp(LossHead):
    """out0 = sum|gen-gt| (tf.norm ord=1, train.py:73), out1 = GDL (ops.py:100-120)."""

    def __init__(self, gen, gt, grad_weights=None, name='frame_loss'):
        if gen.shape != gt.shape:
            raise ValueError('frame loss: shapes differ %s vs %s' % (gen.shape, gt.shape))
        g = G.get_default_graph()
        self.grad_weights = grad_weights
        vals = _new((2,), name + '/values')
        outs = [vals.view(0, (1,), name + '/l1'), vals.view(1, (1,), name + '/gdl')]
        self.vals = vals
        self.dgen = _new(gen.shape, name + '/dgen') if grad_weights else None
        super().__init__(g, name, [gen, gt], outs + ([self.dgen] if self.dgen is not None else []))

    def bind(self, rt):
        gen, gt = self.inputs
        b, h, w, c = gen.shape
        ws, n = rt.workspace(rt.lib.frame_loss_workspace_bytes(gen.numel))
        self._keep = ws
        w1, w2 = self.grad_weights or (0.0, 0.0)
        # the gradient op's own value outputs are read by nobody (loss values come from the forward head, when a program
        # fetches them at all): gradient only - no reductions, no finalize launch
        pout = None if self.dgen is not None else _p(self.outputs[0].buf)
        args = (_p(gen.buf), _p(gt.buf), pout, _p(self.dgen.buf) if self.dgen is not None else None,
                b, h, w, c, float(w1), float(w2), ACG_F32, _p(ws), n)
        fn = rt.lib.frame_loss
        return lambda s: fn(*args, s)

    def seed(self, weights, needs):
        a, b = self.inputs
        if needs[0] and needs[1]:
            raise NotImplementedError('frame loss: gradient w.r.t. both arguments')
        if needs[1]:                       # both terms are symmetric: differentiate w.r.t. the other argument
            a, b = b, a
        op = FrameLossOp(a, b, (weights.get(0, 0.0), weights.get(1, 0.0)), self.name + '/grad')
        return [(a, op.dgen)]


class SmallLossOp(LossHead):
    """One-block heads: 'l2norm' (train.py:77), 'sigmoid_ce' (ops.py:30-31,39-42), 'mean' (ops.py:32-33,44-45)."""

    def __init__(self, kind, x, other=None, label=0.0, grad_scale=None, name=None):
        g = G.get_default_graph()
        self.kind, self.label, self.grad_scale = kind, float(label), grad_scale
        if x.numel > 65536:
            raise ValueError('%s loss: more than 65536 elements' % kind)
        name = name or kind
        self.dx = _new(x.shape, name + '/dx') if grad_scale is not None else None
        ins = [x] + ([other] if other is not None else [])
        super().__init__(g, name, ins, [_new((1,), name + ':0')] + ([self.dx] if self.dx is not None else []))

    def bind(self, rt):
        x = self.inputs[0]
        out, dx = _p(self.outputs[0].buf), (_p(self.dx.buf) if self.dx is not None else None)
        sc = float(self.grad_scale or 0.0)
        if self.kind == 'l2norm':
            args, fn = (_p(x.buf), _p(self.inputs[1].buf), out, dx, x.numel, sc), rt.lib.l2norm_loss
        elif self.kind == 'sigmoid_ce':
            args, fn = (_p(x.buf), self.label, out, dx, x.numel, sc), rt.lib.sigmoid_ce_loss
        else:
            args, fn = (_p(x.buf), out, dx, x.numel, sc), rt.lib.mean_loss
        return lambda s: fn(*args, s)

    def seed(self, weights, needs):
        a = self.inputs[0]
        b = self.inputs[1] if len(self.inputs) > 1 else None
        sign = 1.0
        if b is not None and needs[1]:
            if needs[0]:
                raise NotImplementedError('l2norm: gradient w.r.t. both arguments')
            a, b = b, a                    # ||a-b|| is symmetric
        dp = G.get_default_graph().collections.get('data_parallel')
        if self.kind == 'l2norm' and dp is not None and getattr(dp, 'exact_global_batch', False) and dp.active:
            # ||.||_2 over the GLOBAL batch: this rank's sum of squares -> all-reduce -> gradient with the global norm
            ss = SumSqDiffOp(a, b, self.name + '/sumsq').outputs[0]
            gss = ScalarAllReduceOp(ss, self.name + '/sumsq_allreduce').outputs[0]
            op = L2GlobalGradOp(a, b, gss, sign * weights.get(0, 0.0), self.name + '/grad')
            return [(a, op.dx)]
        op = SmallLossOp(self.kind, a, b, self.label, sign * weights.get(0, 0.0), self.name + '/grad')
        return [(a, op.dx)]


class SumSqDiffOp(G.Op):
    def __init__(self, a, b, name):
        super().__init__(G.get_default_graph(), name, [a, b], [_new((1,), name + ':0')])

    def bind(self, rt):
        a, b = self.inputs
        args, fn = (_p(a.buf), _p(b.buf), _p(self.outputs[0].buf), a.numel), rt.lib.sumsq_diff
        return lambda s: fn(*args, s)


class ScalarAllReduceOp(G.Op):
    """Host op: sum of a small tensor over the ranks."""
    host = True

    def __init__(self, x, name):
        super().__init__(G.get_default_graph(), name, [x], [_new(x.shape, name + ':0')])

    def bind(self, rt):
        src, dst = self.inputs[0], self.outputs[0]

        def run():
            dst.buf.copy_(src.buf)
            rt.comm.all_reduce(dst.buf)
        return run


class L2GlobalGradOp(G.Op):
    def __init__(self, a, b, gss, grad_scale, name):
        self.grad_scale = float(grad_scale)
        self.dx = _new(a.shape, name + '/dx')
        super().__init__(G.get_default_graph(), name, [a, b, gss], [_new((1,), name + ':0'), self.dx])

    def bind(self, rt):
        a, b, gss = self.inputs
        args = (_p(a.buf), _p(b.buf), _p(gss.buf), _p(self.outputs[0].buf), _p(self.dx.buf), a.numel, self.grad_scale)
        fn = rt.lib.l2norm_loss_global
        return lambda s: fn(*args, s)


class PairedLossOp(LossHead):
    """Two 'sigmoid_ce' / 'mean' heads over the two halves of one batched tensor (D(fake) | D(real) run as one
    launch sequence with per-half BatchNorm statistics): out0 is the loss of the first half, out1 of the second."""

    def __init__(self, kind, x, labels=(0.0, 0.0), grad_scales=None, name=None):
        g = G.get_default_graph()
        if x.numel % 2 or x.numel // 2 > 65536:
            raise ValueError('%s loss: halves must be equal and at most 65536 elements' % kind)
        self.kind, self.labels, self.grad_scales = kind, tuple(float(v) for v in labels), grad_scales
        name = name or ('paired_' + kind)
        vals = _new((2,), name + '/values')
        self.dx = _new(x.shape, name + '/dx') if grad_scales is not None else None
        outs = [vals.view(0, (1,), name + '/first'), vals.view(1, (1,), name + '/second')]
        super().__init__(g, name, [x], outs + ([self.dx] if self.dx is not None else []))

    def bind(self, rt):
        x = self.inputs[0]
        half = x.numel // 2
        fn = rt.lib.sigmoid_ce_loss if self.kind == 'sigmoid_ce' else rt.lib.mean_loss
        calls = []
        for k in range(2):
            px = ctypes.c_void_p(x.buf.data_ptr() + 4 * half * k)
            po = ctypes.c_void_p(self.outputs[0].buf.data_ptr() + 4 * k)
            pd = ctypes.c_void_p(self.dx.buf.data_ptr() + 4 * half * k) if self.dx is not None else None
            sc = float(self.grad_scales[k]) if self.grad_scales is not None else 0.0
            calls.append((px, self.labels[k], po, pd, half, sc) if self.kind == 'sigmoid_ce' else (px, po, pd, half, sc))

        def launch(s):
            fn(*calls[0], s)
            fn(*calls[1], s)
        return launch

    def seed(self, weights, needs):
        op = PairedLossOp(self.kind, self.inputs[0], self.labels, (weights.get(0, 0.0), weights.get(1, 0.0)), self.name + '/grad')
        return [(self.inputs[0], op.dx)]


class PsnrOp(LossHead):
    def __init__(self, true, pred, name='psnr'):
        super().__init__(G.get_default_graph(), name, [true, pred], [_new((1,), name + ':0')])

    def bind(self, rt):
        a, b = self.inputs
        ws, n = rt.workspace(rt.lib.frame_loss_workspace_bytes(a.numel))
        self._keep = ws
        args = (_p(a.buf), _p(b.buf), _p(self.outputs[0].buf), a.numel, ACG_F32, _p(ws), n)
        fn = rt.lib.psnr
        return lambda s: fn(*args, s)

    def seed(self, weights, needs):
        raise NotImplementedError('psnr is a metric (ops.py:19-20), not a training loss')


def build_psnr(true, pred):
    """ops.py:19-20."""
    return Scalar([(PsnrOp(true, pred), 0, 1.0)])


def sigmoid_cross_entropy(label, logits, name='sigmoid_ce'):
    """tf.losses.sigmoid_cross_entropy against a constant label, mean over all elements."""
    return Scalar([(SmallLossOp('sigmoid_ce', logits, label=label, name=name), 0, 1.0)])


def reduce_mean(x, name='mean'):
    return Scalar([(SmallLossOp('mean', x, name=name), 0, 1.0)])


def build_g_adv_loss(d_out_gen, arg_loss):
    """ops.py:28-35."""
    if arg_loss == 'bce':
        return sigmoid_cross_entropy(1.0, d_out_gen, 'g_adv_bce')
    elif arg_loss == 'wass':
        return reduce_mean(d_out_gen, 'g_adv_wass')
    else:
        raise ValueError('unexpected loss argument')


def build_d_loss(d_out_direct, d_out_gen, arg_loss, summaries=None):
    """ops.py:37-50 (one-sided label smoothing 0.9).  ``summaries`` receives the two named parts."""
    if arg_loss == 'bce':
        d_direct_loss = sigmoid_cross_entropy(0.9, d_out_direct, 'd_direct_bce')
        d_gen_loss = sigmoid_cross_entropy(0.0, d_out_gen, 'd_gen_bce')
    elif arg_loss == 'wass':
        d_direct_loss = reduce_mean(d_out_direct, 'd_direct_wass')
        d_gen_loss = -reduce_mean(d_out_gen, 'd_gen_wass')
    else:
        raise ValueError('unexpected loss argument')
    if summaries is not None:
        summaries['discriminator_direct_loss'] = d_direct_loss
        summaries['discriminator_gen_loss'] = d_gen_loss
    return d_direct_loss + d_gen_loss


def build_d_loss_batched(d_out_both, arg_loss, summaries=None):
    """ops.py:37-50 for a discriminator run ONCE on [fake ; real] stacked along the batch (BatchNorm per half):
    the first half of ``d_out_both`` is D(fake) = d_out_gen, the second D(real) = d_out_direct."""
    if arg_loss == 'bce':
        head = PairedLossOp('sigmoid_ce', d_out_both, labels=(0.0, 0.9), name='d_bce')
        d_gen_loss, d_direct_loss = Scalar([(head, 0, 1.0)]), Scalar([(head, 1, 1.0)])
    elif arg_loss == 'wass':
        head = PairedLossOp('mean', d_out_both, name='d_wass')
        d_gen_loss, d_direct_loss = Scalar([(head, 0, -1.0)]), Scalar([(head, 1, 1.0)])
    else:
        raise ValueError('unexpected loss argument')
    if summaries is not None:
        summaries['discriminator_direct_loss'] = d_direct_loss
        summaries['discriminator_gen_loss'] = d_gen_loss
    return d_direct_loss + d_gen_loss


def frame_losses(g_out, next_frames):
    """-> (sum|g_out-next_frames|, GDL) from one fused kernel (train.py:73; ops.py:100-120, alpha=1)."""
    cache = G.get_default_graph().collections.setdefault('frame_loss_heads', {})
    key = frozenset((g_out.root().id, next_frames.root().id))
    head = cache.get(key)
    if head is None:
        head = cache[key] = FrameLossOp(g_out, next_frames)
    # the GDL is a SUM over the batch (ops.py:120): averaging the ranks' gradients divides it by the world size relative
    # to one device at the global batch; the exact-global-batch mode scales it back (SURVEY 8(e) caveat 2)
    dp = G.get_default_graph().collections.get('data_parallel')
    gdl_w = float(dp.world_size) if (dp is not None and getattr(dp, 'exact_global_batch', False) and dp.active) else 1.0
    return Scalar([(head, 0, 1.0)]), Scalar([(head, 1, gdl_w)])


def build_gdl(g_out, next_frames, alpha=1):
    """ops.py:100-120 (the loss is symmetric in its arguments; the reference passes them swapped, D10)."""
    if alpha != 1:
        raise ValueError('build_gdl: only alpha=1 (the reference value) is implemented')
    return frame_losses(g_out, next_frames)[1]


def l1_norm(a, b):
    """tf.norm(a-b, ord=1, axis=None) (train.py:73)."""
    return frame_losses(a, b)[0]


def l2_norm(a, b, name='l2norm'):
    """tf.norm(a-b, ord=2, axis=None) (train.py:77)."""
    if a.numel != b.numel:
        raise ValueError('l2_norm: %s vs %s' % (a.shape, b.shape))
    return Scalar([(SmallLossOp('l2norm', a, other=b, name=name), 0, 1.0)])
