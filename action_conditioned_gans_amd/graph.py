"""Static-graph host runtime: the reference's TF-1.0 graph/session contract, MI355X-first.

The reference builds a TensorFlow graph once (Trainer.__init__, train.py:27-112) and then
executes pruned sub-graphs with ``sess.run(fetches, feed_dict)`` (train.py:114-155).  This
module keeps that contract and nothing more:

* ``Tensor`` / ``Variable`` / ``Op`` record a static program whose every buffer is allocated
  once (activations, gradients, workspaces) - no allocator traffic at step time;
* ``Session.run`` prunes to the ops the fetches need (exactly TF's behaviour: the D step
  never back-propagates into G, the G step never runs D(real), a ``test`` fetch runs G only),
  binds each op to a C-ABI call of libacgan_hip.so, and replays the resulting launch list as
  ONE captured HIP graph per fetch signature (segments are cut only around host-side
  collectives);
* all variables of a top-level scope ('g', 'd') live in ONE flat buffer and every optimizer
  owns one flat gradient buffer with the same layout, so weight gradients are written in place
  by the wgrad kernels, the data-parallel all-reduce works on contiguous buckets with no
  packing copy, and a whole network is updated by one fused optimizer launch.

PyTorch is used for device memory, streams, graph capture and torch.distributed only.
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _lib

_ALIGN = 4  # floats: every variable starts 16-byte aligned inside its flat buffer


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Tensor:
    """A statically shaped float32 (or int32) value in the graph; ``buf`` is bound by the session."""

    def __init__(self, graph, shape, name=None, dtype=torch.float32, op=None, init=None):
        self.graph = graph
        self.shape = tuple(int(s) for s in shape)
        self.name = name
        self.dtype = dtype
        self.op = op                # producing Op (None: placeholder, variable, state slot)
        self.init = init            # constant fill value for state slots (optimizer slots, counters)
        self.view_of = None         # (base Tensor, element offset): a window into a flat buffer
        self.alias_of = None        # same storage as another tensor, different shape (reshape)
        self.buf = None
        self.valid_c = None         # logical channel count when the last dimension is a padded channel pitch
        self.id = graph._next_id()

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def view(self, offset, shape, name=None):
        t = Tensor(self.graph, shape, name=name, dtype=self.dtype, op=self.op)
        t.view_of = (self, int(offset))
        return t

    def akey(self):
        """Identity for differentiation: reshapes share their source's gradient, windows do not."""
        t = self
        while t.alias_of is not None:
            t = t.alias_of
        return t.id

    def reshape(self, shape, name=None):
        t = Tensor(self.graph, shape, name=name, dtype=self.dtype, op=self.op)
        assert t.numel == self.numel, 'reshape changes element count'
        t.alias_of = self
        return t

    def root(self):
        t = self
        while t.alias_of is not None or t.view_of is not None:
            t = t.alias_of if t.alias_of is not None else t.view_of[0]
        return t

    def __repr__(self):
        return 'Tensor(%s, %s)' % (self.name, self.shape)


class Placeholder(Tensor):
    pass


class Variable(Tensor):
    """A trainable parameter; ``value`` holds the host initial value until the initializer runs."""

    def __init__(self, graph, name, value, trainable=True):
        super().__init__(graph, value.shape, name=name)
        self.value = value.detach().to(torch.float32).contiguous()
        self.trainable = trainable
        self.scope = name.split('/')[0]


class Op:
    """One node of the static program.  ``bind(rt)`` returns ``fn(stream_ptr)`` that enqueues it."""
    host = False        # True: runs between HIP-graph segments (torch kernels of the exact-global-batch mode)
    side_stream = False  # True: runs on the session's second stream, overlapping the main chain (gradient all-reduce)
    run_last = False    # True: ordered after everything else in a program (weight clip, defect D6)

    def __init__(self, graph, name, inputs, outputs, control_inputs=()):
        self.graph, self.name = graph, name
        self.inputs, self.outputs = list(inputs), list(outputs)
        self.control_inputs = list(control_inputs)
        self.index = float(graph._next_id())
        if graph._side_default and not self.host:
            self.side_stream = True      # built inside Graph.side_branch(): a chain that runs beside the main one
        for o in self.outputs:
            if o.op is None:
                o.op = self
        graph.ops.append(self)

    def bind(self, rt):
        raise NotImplementedError

    def grad(self, gouts, needs, ctx):
        raise NotImplementedError('%s has no gradient' % type(self).__name__)

    def __repr__(self):
        return '%s(%s)' % (type(self).__name__, self.name)


class Graph:
    def __init__(self):
        self.ops = []
        self.variables = {}          # name -> Variable, creation order
        self.state = []              # state slots with constant init (optimizer slots, counters)
        self._id = 0
        self._side_default = False   # inside side_branch(): new ops are flagged side_stream
        self.feed_aliases = {}       # id(placeholder) -> [(tensor, channel offset)]: also written by that placeholder's feed
        self._layouts = {}           # top-level scope -> (offsets dict, total numel, flat Tensor)
        self.collections = {}
        # storage type of activation-class tensors (conv / BatchNorm outputs and their gradients): float32, or
        # bfloat16 for the BASELINE config 3 / 5 pipeline (Session(dtype='bf16') sets it before the graph is built)
        self.act_dtype = torch.float32
        self.weight_copies = {}      # scope -> [(Variable, rm Tensor, tr Tensor)]: bf16 operand copies of the filters

    @property
    def cpad(self):
        """Channel-pitch unit of activation tensors: one 16-byte gather = 4 floats or 8 bf16."""
        return 8 if self.act_dtype == torch.bfloat16 else 4

    def add_feed_alias(self, placeholder, dst, c_off, tile=None):
        """Whenever ``placeholder`` ([..., C] float32) is fed, its value is ALSO written into channels
        [c_off, c_off + C) of ``dst`` (same leading dimensions, any channel pitch, float32 or bf16) - by the same feed copy,
        for every program that reads ``dst`` (ops.ConcatChannelsOp: a concatenation whose inputs are fed needs no launch).
        ``tile`` = (div, mod): row r of ``dst`` takes row (r // div) % mod of the [B, C] placeholder instead - tf.tile of
        the action vector over a feature map, shared by the sub-batches of a joined batch (ops.ConcatActionsOp)."""
        if tile is None and placeholder.shape[:-1] != dst.shape[:-1]:
            raise ValueError('feed alias: %r does not fit %r' % (placeholder, dst))
        if c_off + placeholder.shape[-1] > dst.shape[-1]:
            raise ValueError('feed alias: %r does not fit %r at channel %d' % (placeholder, dst, c_off))
        self.feed_aliases.setdefault(id(placeholder), []).append((dst, int(c_off), tile))

    @contextlib.contextmanager
    def side_branch(self, on=True):
        """Ops built inside (and, through build_gradients, their gradient ops) form a SIDE chain: independent of what the
        main chain does next - the DNA generator's state head next to its frame decoder, models.py:44-51 vs 53-72 - so a
        GPU session enqueues them on its second HIP stream (Session._launch_segment: one fork edge where the chain
        starts, one join in front of the first op that needs a result) and, inside a captured program, they are a
        parallel branch of the HIP graph.  Small latency-bound kernels then hide behind the main chain's large ones."""
        prev, self._side_default = self._side_default, bool(on)
        try:
            yield
        finally:
            self._side_default = prev

    def _next_id(self):
        self._id += 1
        return self._id

    # ---- variables ------------------------------------------------------------------------
    def get_variable(self, name, shape, initializer, reuse):
        """tf.get_variable under variable_scope(reuse=...): create, or fetch when reuse is set."""
        if name in self.variables:
            if not reuse:
                raise ValueError('Variable %s already exists, disallowed. Did you mean to set reuse=True?' % name)
            v = self.variables[name]
            if v.shape != tuple(shape):
                raise ValueError('Trying to share variable %s, but specified shape %s and found shape %s'
                                 % (name, tuple(shape), v.shape))
            return v
        if reuse:
            raise ValueError('Variable %s does not exist, or was not created with get_variable()' % name)
        scope = name.split('/')[0]
        if scope in self._layouts:
            raise ValueError('scope %r is frozen (an optimizer or the initializer already laid it out)' % scope)
        v = Variable(self, name, initializer(tuple(shape)))
        self.variables[name] = v
        return v

    def trainable_variables(self, scope=None):
        """tf.get_collection(TRAINABLE_VARIABLES, scope) (train.py:87-88): prefix match on the name."""
        return [v for n, v in self.variables.items()
                if v.trainable and (scope is None or n == scope or n.startswith(scope.rstrip('/') + '/'))]

    def layout(self, scope):
        """Freeze a top-level scope into one flat buffer; returns (offsets, total, flat Tensor)."""
        if scope not in self._layouts:
            offs, total = {}, 0
            for n, v in self.variables.items():
                if v.scope == scope:
                    offs[n] = total
                    total += -(-v.numel // _ALIGN) * _ALIGN
            if total == 0:
                raise ValueError('no variables in scope %r' % scope)
            flat = Tensor(self, (total,), name=scope + '/flat_params')
            for n, off in offs.items():
                self.variables[n].view_of = (flat, off)
            self._layouts[scope] = (offs, total, flat)
        return self._layouts[scope]

    def new_state(self, shape, init, name, dtype=torch.float32):
        t = Tensor(self, shape, name=name, dtype=dtype, init=init)
        self.state.append(t)
        return t


_default = [Graph()]


def get_default_graph():
    return _default[-1]


def reset_default_graph():
    _default[-1] = Graph()
    return _default[-1]


def placeholder(shape, name=None, dtype=torch.float32, channel_pitch=None, act=False):
    """tf.placeholder.  ``channel_pitch`` stores the last dimension with that pitch (zero pad channels) so that 3-
    or 6-channel images can be gathered with 16-byte loads by the first conv layer; ``act`` stores it in the graph's
    activation type (fed float32 values are rounded on the way in)."""
    if act:
        dtype = get_default_graph().act_dtype
    if channel_pitch and channel_pitch != shape[-1]:
        ph = Placeholder(get_default_graph(), tuple(shape[:-1]) + (channel_pitch,), name=name, dtype=dtype)
        ph.valid_c = shape[-1]
        return ph
    return Placeholder(get_default_graph(), shape, name=name, dtype=dtype)


class InitOp(Op):
    """tf.global_variables_initializer(): lays variables out flat and fills variables + slots."""

    def __init__(self, graph):
        super().__init__(graph, 'init', [], [])


def global_variables_initializer():
    return InitOp(get_default_graph())


# ---- gradient construction -------------------------------------------------------------------
class GradContext:
    """Hands each wgrad-type op its window of the optimizer's flat gradient buffer."""

    def __init__(self, graph, var_list, flat_grad, offsets):
        self.graph = graph
        self.flat_grad = flat_grad
        self.offsets = offsets
        self.var_set = {v.name for v in var_list}
        self.writers = {}            # var name -> list of ops that wrote its gradient

    def wants(self, var):
        return var.name in self.var_set

    def slot(self, var):
        """-> (gradient window Tensor, accumulate flag: 0.0 first write, 1.0 afterwards)."""
        first = var.name not in self.writers
        self.writers.setdefault(var.name, [])
        return self.flat_grad.view(self.offsets[var.name], var.shape, name='grad/' + var.name), (0.0 if first else 1.0)

    def wrote(self, var, op):
        self.writers[var.name].append(op)


def build_gradients(graph, heads, var_list, flat_grad, offsets, add_op):
    """Reverse walk over the recorded ops (what tf.gradients does for train.py:100-102).

    heads: {loss-head Op: {output index: weight}}; each head emits its own seed gradient(s).
    Only paths that reach ``var_list`` are built, so e.g. the D step gets no dgrad through G and
    none for d/conv1.  Returns the GradContext.
    """
    var_names = {v.name for v in var_list}
    last = max(o.index for o in graph.ops)
    reach = set()                                  # tensor ids that depend on a variable in var_list
    fwd_ops = [o for o in graph.ops if o.index <= last]
    for op in fwd_ops:
        if any((isinstance(i, Variable) and i.name in var_names) or i.akey() in reach for i in op.inputs):
            for o in op.outputs:
                reach.add(o.akey())
    pending = {}
    for head, weights in heads.items():
        needs = [(not isinstance(i, Variable)) and i.akey() in reach for i in head.inputs]
        if not any(needs):
            continue
        with graph.side_branch(head.side_stream):
            seeds = list(head.seed(weights, needs))
        for t, g in seeds:
            r = t.akey()
            pending[r] = g if r not in pending else add_op(pending[r], g)
    ctx = GradContext(graph, var_list, flat_grad, offsets)
    for op in reversed(fwd_ops):
        gouts = [pending.get(o.akey()) for o in op.outputs]
        if all(g is None for g in gouts):
            continue
        needs = [(isinstance(i, Variable) and i.name in var_names) or (not isinstance(i, Variable) and i.akey() in reach)
                 for i in op.inputs]
        if not any(needs):
            continue
        with graph.side_branch(op.side_stream):      # the gradient ops of a side chain are a side chain
            gins = op.grad(gouts, needs, ctx)
        for i, g in zip(op.inputs, gins):
            if g is None or isinstance(i, Variable):
                continue
            r = i.akey()
            pending[r] = g if r not in pending else add_op(pending[r], g)
    return ctx


# ---- session ----------------------------------------------------------------------------------------
class _Program:
    def __init__(self, segments, fetch_tensors, feeds):
        self.segments = segments     # list of ('dev', [fn...]) / ('host', fn)
        self.fetch_tensors = fetch_tensors
        self.feeds = feeds
        self.graphs = None           # captured HIP graphs, one per 'dev' segment
        self.eager = False           # always launched eagerly (contains stream-ordered collectives)
        self.runs = 0
        self.missing_feeds = []
        self.used_feeds = frozenset()
        self.alias_copies = {}


class Runtime:
    """What an Op needs to bind itself: the kernel library, the device, buffers, process group."""

    def __init__(self, lib, device, world_size=1, rank=0, process_group=None, conv_dtype=0, comm=None):
        self.lib, self.device = lib, torch.device(device)
        self.conv_dtype = conv_dtype          # ACG_F32 / ACG_BF16 for the conv contractions
        self.world_size, self.rank, self.process_group = world_size, rank, process_group
        self.is_cuda = self.device.type == 'cuda'
        self.side_stream = torch.cuda.Stream(self.device) if self.is_cuda else None
        self.upload_stream = None              # created by the first host-array feed (Session.upload)
        self.program_ops = frozenset()         # ids of the ops of the program being compiled (ops.py: hand-offs between neighbours)
        # split-K slabs summed by the consuming BatchNorm kernel instead of by a launch of their own: bit-identical.  With the
        # resident (one block per channel quad) kernels alone it measured SLOWER (profiles/r2: 16 bytes per row and block, times
        # S slabs); the grid kernels of round 4 read whole rows of every slab and it pays (Session sets it; default on)
        self.slab_handoff = 0
        self.slab_rows = False                 # hand-offs in the ACG_SLABS_ROWS layout too (Session(slab_handoff=True / N))
        self.epilogue_stats = True
        self.epilogue_bias = True
        self.fuse_weight_refresh = True        # bf16: optimizer update + refresh of the bf16 filter copies in one launch (optim.StepOp)
        # `flags` of the BatchNorm entries (acgan_hip.h): ACG_BN_NO_GRID_EXCHANGE where two BatchNorm launches can overlap (Session
        # sets it with side_branches: two partially resident grid-exchange kernels would starve each other)
        self.bn_flags = 0
        self._comm = comm
        self._owns_comm = comm is None          # a communicator handed in (tests sharing one over several sessions) is its owner's to destroy
        self._scratch = {}

    @property
    def comm(self):
        """The gradient transport (comm.py), created on first use: this process's own RCCL communicator on a GPU, the
        given gloo process group on the CPU."""
        if self._comm is None:
            from . import comm as C
            self._comm = C.create(self.device, self.world_size, self.rank, self.process_group)
        return self._comm

    def workspace(self, nbytes):
        """A private zero-initialised scratch buffer (never shared: ops may overlap across streams).  It lives as long
        as the runtime: the launch closures of EVERY compiled program hold its raw address, also after the op is
        bound again for another fetch signature."""
        n = max(int(nbytes), 16)
        buf = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._scratch.setdefault('workspaces', []).append(buf)
        return buf, n

    def state_workspace(self, nbytes):
        """A workspace that is STATE of one call site: acg_bn_act_fwd / _bwd (ABI 6) keep their exchange epoch and a timeout flag
        (uint32 word 2) in it.  Remembered so that `check_exchange_flags` can read the flags."""
        buf, n = self.workspace(nbytes)
        self._scratch.setdefault('state_workspaces', []).append(buf)
        return buf, n

    def check_exchange_flags(self):
        """Raise if a block of a one-launch BatchNorm kernel ever gave up waiting for its peers (acgan_hip.h: the launch then
        finished with what it had - a wrong result - instead of hanging).  One device synchronisation: call it outside timed
        regions - train() does at every log interval and before every checkpoint, bench.py before it prints, Session.close."""
        bufs = [b for b in self._scratch.get('state_workspaces', []) if b.numel() >= 12]
        if not bufs:
            return
        if len(bufs) != self._scratch.get('flag_view_count'):
            # one gather of every call site's word 2 instead of a Python loop of tiny device reads: the addresses are fixed
            self._scratch['flag_views'] = [b[8:12].view(torch.int32) for b in bufs]
            self._scratch['flag_view_count'] = len(bufs)
        flags = torch.cat(self._scratch['flag_views']).cpu()
        if bool((flags != 0).any()):
            bad = [i for i, f in enumerate(flags.tolist()) if f]
            raise _lib.AcgError('BatchNorm grid exchange timed out in %d of %d call sites (first: #%d): the blocks of a one-launch kernel were not '
                           'all resident - results of those launches are wrong' % (len(bad), len(bufs), bad[0]))

    def edge_pool(self, n):
        """n reusable stream-ordering edges (acg_stream_edge: HIP events WITHOUT the system-scope fence of a default
        event, which costs ~20 us per edge on this 8-XCD part)."""
        pool = self._scratch.setdefault('edges', [])
        while len(pool) < n:
            e = ctypes.c_void_p()
            self.lib.stream_edge_create(ctypes.byref(e))
            pool.append(e)
        return pool[:n]

    def stream_ptr(self):
        if self.is_cuda:
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None


class Session:
    """``tf.Session`` stand-in bound to one GPU (one process per GPU)."""

    def __init__(self, device='cuda:0', graph=None, use_hip_graphs=True, lib=None,
                 world_size=1, rank=0, process_group=None, dtype='f32', pair_bwd=True, comm=None, slab_handoff=True, epilogue_stats=True, side_branches=False,
                 epilogue_bias=True, fuse_weight_refresh=True, bn_grid_exchange=None):
        self.graph = graph or get_default_graph()
        dev = torch.device(device)
        if lib is None:
            lib = _lib.get()      # raises when libacgan_hip.so is not built: there is no fallback
            if dev.type != 'cuda':
                raise RuntimeError('the HIP library runs on a GPU device, got %s' % dev)
            if not torch.cuda.is_available():
                raise RuntimeError('no GPU visible: the HIP path cannot run and there is no CPU fallback')
        if dtype not in ('f32', 'bf16'):
            raise ValueError("dtype must be 'f32' or 'bf16' (bf16 activations and matrix-core operands; fp32 master weights, "
                             "weight gradients, statistics and losses)")
        want = torch.bfloat16 if dtype == 'bf16' else torch.float32
        if self.graph.act_dtype != want:
            if self.graph.ops:
                raise ValueError('Session(dtype=%r): the graph was already built for %s activations' % (dtype, self.graph.act_dtype))
            self.graph.act_dtype = want
        self.rt = Runtime(lib, dev, world_size, rank, process_group, _lib.ACG_BF16 if dtype == 'bf16' else _lib.ACG_F32, comm)
        # split-K hand-off: a split layer followed by its BatchNorm leaves its float32 slabs and the BatchNorm kernel sums them while
        # it loads its rows - no reduction launch.  Round 4: ON (True) - the one-launch grid kernels of bn.hip read slabs laid out
        # like the tensor with coalesced rows (+0.8 % config 2, +0.9 % config 3; rounds 2-3, resident kernels only: -4.5 % / level).
        # False / 0: off; 'quads': only layers whose BatchNorm reads the quad slab layout; an int N: only layers split into <= N slabs
        if not (slab_handoff is True or slab_handoff is False or slab_handoff is None or slab_handoff == 'quads'
                or (isinstance(slab_handoff, int) and slab_handoff >= 0)):
            raise ValueError("slab_handoff must be False, True, 'quads' or a non-negative int, got %r" % (slab_handoff,))
        self.rt.slab_handoff = (1 << 30) if (slab_handoff is True or slab_handoff == 'quads') else int(slab_handoff or 0)
        self.rt.slab_rows = slab_handoff != 'quads'
        self.rt.epilogue_stats = bool(epilogue_stats)     # BatchNorm statistics out of the producing conv's epilogue (ops.Conv2dOp.bind)
        self.rt.epilogue_bias = bool(epilogue_bias)       # bias + activation of a transposed head layer in its epilogue (models.py:20-21)
        # the C-oracle stand-in library (CPU tests) implements float32 only: the fused entry is a bf16-pipeline entry
        self.rt.fuse_weight_refresh = bool(fuse_weight_refresh)
        if dev.type == 'cuda':
            torch.cuda.set_device(dev)
        self.use_hip_graphs = use_hip_graphs and dev.type == 'cuda'
        # a layer's input gradient and weight gradient in ONE launch (acg_conv2d_bwd_pair) when they are neighbours in a program
        self.pair_bwd = bool(pair_bwd)
        # side chains (Graph.side_branch) on the second stream.  OFF by default: they then stay where they were built, on
        # the main stream - measured faster on this stack (profiles/r2/m_side_branch_ab.txt: a parallel branch of the HIP
        # graph costs more in fork / join edges than the state head's ~70 us of small kernels hide; bit-identical results)
        self.side_branches = bool(side_branches) and dev.type == 'cuda'
        # bn_grid_exchange=False: never the one-launch grid-exchange BatchNorm kernels (ACG_BN_NO_GRID_EXCHANGE).  Forced with
        # side_branches: a side chain's BatchNorm can run beside a main-chain BatchNorm, and two grid-exchange kernels must
        # never overlap (two partially resident grids starve each other until both time out).  None (default) = the policy of
        # _bn_flags: off where a multi-rank collective shares the device with them, on otherwise; True = on whatever runs beside
        self._bn_grid_exchange = bn_grid_exchange
        self.rt.bn_flags = self._bn_flags()
        self._programs = {}
        self._initialized = False
        self._weights_dirty = True     # bf16 operand copies of the filters are stale (initializer, set_value, restore)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        # an exception is already on its way out: tear down, but do not replace it with a flag error of our own
        self.close(check=exc_type is None)
        return False

    def close(self, check=True):
        """Check the device-side flags (Runtime.check_exchange_flags; ``check=False`` skips that) and tear the gradient transport
        down (ncclCommDestroy) - also when the check raises; the session must not run afterwards."""
        try:
            if check:
                self.rt.check_exchange_flags()
        finally:
            if self.rt._comm is not None:
                comm, self.rt._comm = self.rt._comm, None
                if self.rt._owns_comm:
                    comm.destroy()

    # ---- buffers
    def _materialize(self, t):
        if t.buf is not None:
            return t.buf
        if t.alias_of is not None:
            t.buf = self._materialize(t.alias_of).view(t.shape)
        elif t.view_of is not None:
            base, off = t.view_of
            t.buf = self._materialize(base).view(-1)[off:off + t.numel].view(t.shape)
        else:
            if t.init is not None:
                t.buf = torch.full(t.shape, t.init, dtype=t.dtype, device=self.rt.device)
            else:
                t.buf = torch.zeros(t.shape, dtype=t.dtype, device=self.rt.device)
        return t.buf

    def _initialize(self):
        g = self.graph
        for scope in sorted({v.scope for v in g.variables.values()}):
            g.layout(scope)
        for v in g.variables.values():
            v.buf = None
            self._materialize(v).copy_(v.value)
        for s in g.state:
            buf = self._materialize(s)
            buf.fill_(s.init)
        self._initialized = True
        self._weights_dirty = True

    def upload(self, value):
        return self.upload_many([value])[0]

    def upload_many(self, values):
        """Host arrays (the reference's numpy feed_dict values) -> their float32 device copies, WITHOUT stalling the host or
        the compute stream: the arrays are packed into ONE pinned staging buffer (a ring of 24 slots owned by the session; a
        fresh ``pin_memory()`` per array costs milliseconds on this stack) and go to the device as ONE copy on a copy stream of
        its own; the compute stream waits for that copy by an event.  A pageable ``tensor.to(device)`` per array instead is
        synchronous for the host AND ordered behind the step still running on the compute stream - the host could not prepare
        the next step's inputs while the GPU worked (0.6 ms of a 2.8 ms iteration through the numpy API, round 5).
        Device tensors pass through (converted to float32)."""
        rt = self.rt
        out, host = [None] * len(values), []
        for i, value in enumerate(values):
            if torch.is_tensor(value):
                if value.is_cuda or not rt.is_cuda:
                    out[i] = value.detach().to(rt.device, dtype=torch.float32)
                    continue
                value = value.detach().numpy()
            if not rt.is_cuda:
                out[i] = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
                continue
            host.append((i, np.asarray(value)))
        if not host:
            return out
        main = torch.cuda.current_stream(rt.device)
        mode = os.environ.get('ACG_UPLOAD', 'auto')        # measurement hook: 'pageable' / 'staged' force one path (profiles/r5/k_upload_ab.txt)
        if mode == 'pageable' or (mode == 'auto' and main.query()):
            # the compute stream is idle (a caller that fetches results to the host every step, as the reference's loop does):
            # nothing to overlap with, and the runtime's own pageable copy - which pins large user buffers in place instead of
            # copying them - is then as fast (1.5 MB arrays) or faster (6 MB: 220 vs 198 steps/s at 128 x 128) than staging
            for i, arr in host:
                out[i] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(rt.device, non_blocking=True)
            return out
        if rt.upload_stream is None:
            rt.upload_stream = torch.cuda.Stream(rt.device)
            rt.upload_ring = [[torch.cuda.Event(), None] for _ in range(24)]      # [event, pinned staging buffer]
            rt.upload_count = 0
        # the host may run ahead of the compute stream by at most 24 uploads (a few training iterations): the event recorded
        # on the compute stream 24 uploads ago must have been reached - the slot's staging buffer is then free again, and the
        # device copies in flight stay bounded
        slot = rt.upload_ring[rt.upload_count % len(rt.upload_ring)]
        if rt.upload_count >= len(rt.upload_ring):
            slot[0].synchronize()
        rt.upload_count += 1
        offsets, total = [], 0
        for _, arr in host:
            offsets.append(total)
            total += -(-arr.size // 64) * 64                       # 256-byte segments: every device view is 16-byte aligned
        if slot[1] is None or slot[1].numel() < total:
            slot[1] = torch.empty(total, dtype=torch.float32, pin_memory=True)
        staged = slot[1].numpy()
        with torch.cuda.stream(rt.upload_stream):
            dev = torch.empty(total, dtype=torch.float32, device=rt.device)
            sent = 0
            for k, ((_, arr), off) in enumerate(zip(host, offsets)):
                # numpy's memcpy / cast (torch's threaded host copy_ of 1.5 MB costs milliseconds on a many-core host; slicing a
                # 6 MB array over four threads of our own measured no faster than one memcpy)
                np.copyto(staged[off:off + arr.size].reshape(arr.shape), arr, casting='unsafe')
                end = total if k == len(host) - 1 else offsets[k + 1]
                if end - sent >= (1 << 18) or end == total:        # >= 1 MB staged (or the rest): its DMA runs while the next array is staged
                    dev[sent:end].copy_(slot[1][sent:end], non_blocking=True)
                    sent = end
        main.wait_stream(rt.upload_stream)     # whatever reads the copy next on the compute stream is ordered behind it
        dev.record_stream(main)                # (allocated on the copy stream, used on the compute stream)
        slot[0].record(main)
        for (i, arr), off in zip(host, offsets):
            out[i] = dev[off:off + arr.size].view(arr.shape)
        return out

    # ---- variable access (checkpoint / parity tests)
    def get_value(self, var):
        return self._materialize(var).detach().cpu().clone()

    def set_value(self, var, value):
        self._materialize(var).copy_(torch.as_tensor(np.asarray(value) if not torch.is_tensor(value) else value)
                                     .to(torch.float32).reshape(var.shape))
        self._weights_dirty = True

    def _refresh_weight_copies(self):
        """bf16 pipeline: the filter copies the conv kernels read (acg_weights_prepare_bf16) follow the float32 master
        weights - inside a training program right behind the optimizer update, here after host-side writes."""
        self._weights_dirty = False
        if not self.graph.weight_copies:
            return
        from . import ops as O
        for scope in self.graph.weight_copies:
            for t3 in self.graph.weight_copies[scope]:
                for t in t3:
                    self._materialize(t)
            fn = O.prepare_weights_launch(self.rt, self.graph, scope)
            if fn is not None:
                fn(self.rt.stream_ptr())

    # ---- compile
    @staticmethod
    def _flatten(fetches):
        out = []

        def rec(f):
            if isinstance(f, (list, tuple)):
                for x in f:
                    rec(x)
            else:
                out.append(f)
        rec(fetches)
        return out

    def _bn_flags(self, beside_collective=False):
        """The BatchNorm entries' ``flags`` for an op of this session's programs.  The one-launch kernels exchange their partial
        sums between the blocks of ONE grid and need all of them resident at once; the large tensors take exactly one
        1024-thread block per CU of the whole device.  With more than one rank and ``collectives='side'`` an RCCL ring kernel
        holds CUs on the second stream from the launch of a bucket's all-reduce until the join in front of the optimizer: a
        BatchNorm grid launched in that window cannot become fully resident until the all-reduce is over - correct (the spin is
        bounded far above an all-reduce), but the overlap the side stream exists for is gone.  That this does NOT happen beside
        a real multi-rank ring kernel could not be shown on the one-GPU boxes this was built on, so the BatchNorm launches
        INSIDE that window - the backward passes of the early layers - take the two-launch kernels; every
        forward pass and the backward passes in front of the first all-reduce run beside nothing and keep the one-launch
        kernels, as do one-rank sessions and 'stream' collectives.
        ``bn_grid_exchange``: None = that policy when the graph's data-parallel configuration has more than one rank and side
        collectives; 'not_beside_collectives' = that policy whatever the rank count (measurement / tests on one rank); True =
        one-launch kernels everywhere; False = nowhere."""
        mode = self._bn_grid_exchange
        if self.side_branches or mode is False:
            return _lib.BN_NO_GRID_EXCHANGE
        if mode is None:
            dp = self.graph.collections.get('data_parallel')
            mode = 'not_beside_collectives' if (dp is not None and dp.world_size > 1 and dp.collectives == 'side') else True
        return _lib.BN_NO_GRID_EXCHANGE if (mode == 'not_beside_collectives' and beside_collective) else 0

    def _assign_bn_flags(self, ops):
        """Per BatchNorm launch of the program ``ops``: the flags of _bn_flags, by its position relative to the program's first
        side-stream collective.  Stored on the forward op (ops.BnActOp.flags_fwd / flags_bwd): the producer that hands its
        split-K slabs to a BatchNorm asks that op for the layout and must see what the consuming launch will see."""
        first = next((i for i, o in enumerate(ops) if getattr(o, 'is_collective', False) and o.side_stream), None)
        n_two_launch = 0
        for i, o in enumerate(ops):
            flags = self._bn_flags(first is not None and i > first)
            if hasattr(o, 'flags_fwd'):
                o.flags_fwd = flags
            elif hasattr(getattr(o, 'fwd', None), 'flags_bwd'):
                o.fwd.flags_bwd = flags
            else:
                continue
            n_two_launch += 1 if flags else 0
        return n_two_launch

    def _compile(self, flat_fetches, feeds, skip_ops=()):
        """``skip_ops``: ops whose results are ALREADY in their output tensors (another program of this session left them there:
        Trainer's look-ahead generator pass) - the dependency walk stops at them, they are not launched, their outputs are bound
        as they lie, and no feed alias writes into them either (a concatenation's fed channels are part of its result)."""
        g = self.graph
        self.rt.bn_flags = self._bn_flags()       # (the data-parallel configuration may have been set after the session was made)
        skip = frozenset(id(o) for o in skip_ops)
        skip_outputs = frozenset(id(t) for o in skip_ops for t in o.outputs)
        needed, stack = {}, []
        for f in flat_fetches:
            if isinstance(f, Op):
                stack.append(f)
            elif isinstance(f, Tensor) or hasattr(f, 'tensor'):
                t = f if isinstance(f, Tensor) else f.tensor()
                if t.op is not None:
                    stack.append(t.op)
            else:
                raise TypeError('Fetch argument %r has invalid type %s' % (f, type(f)))
        while stack:
            op = stack.pop()
            if id(op) in needed or id(op) in skip:
                continue
            needed[id(op)] = op
            fed = getattr(op, 'fed_inputs', ())
            for t in op.inputs:
                if t.op is not None and id(t.op) not in needed and not any(t is f for f in fed):
                    stack.append(t.op)
            stack.extend(c for c in op.control_inputs if id(c) not in needed)
        ops = sorted(needed.values(), key=lambda o: (o.run_last, o.index))
        if any(isinstance(o, InitOp) for o in ops):
            return None
        if not self._initialized:
            raise RuntimeError('Attempting to use uninitialized variables: run global_variables_initializer() first')
        ops = self._fold_clips(ops)
        for op in ops:
            for t in op.inputs + op.outputs + list(getattr(op, 'extras', ())):
                self._materialize(t)
        ops = self._hoist_side_chains(ops) if self.side_branches else ops
        self.rt.program_ops = frozenset(id(o) for o in ops)
        for k, op in enumerate(ops):           # pairing is decided per program: both ops fetched, nothing between them
            w = getattr(op, 'pair_w', None)
            active = self.pair_bwd and w is not None and k + 1 < len(ops) and ops[k + 1] is w
            if w is not None:
                op.pair_active, w.paired = active, active
        self.bn_two_launch_ops = max(getattr(self, 'bn_two_launch_ops', 0), self._assign_bn_flags(ops))
        segments, cur = [], []
        for op in ops:
            fn = op.bind(self.rt)
            if fn is None:
                continue
            if op.host:
                if cur:
                    segments.append(('dev', cur))
                    cur = []
                segments.append(('host', fn))
            else:
                cur.append((op, fn))
        if cur:
            segments.append(('dev', cur))
        fetch_tensors = []
        for f in flat_fetches:
            if isinstance(f, Op):
                fetch_tensors.append(None)
            else:
                t = f.tensor() if hasattr(f, 'tensor') else f
                self._materialize(t)
                fetch_tensors.append(t)
        for ph in feeds:
            self._materialize(ph)
        prog = _Program(segments, fetch_tensors, list(feeds))
        # placeholders this program reads (directly, or as the storage a view aliases): a feed for any other placeholder is
        # validated but not copied - Trainer.train_d feeds next_state like the reference does, and the D step never reads it
        read = set()
        for op in ops:
            fed = getattr(op, 'fed_inputs', ())
            for t in list(op.inputs) + list(getattr(op, 'extras', ())):
                if any(t is f for f in fed):
                    continue                      # this op takes the value through a feed alias, not from the placeholder
                read.add(id(t))
                read.add(id(t.root()))
        prog.used_feeds = frozenset(id(ph) for ph in feeds
                                    if id(ph) in read or any(t is not None and t.root() is ph for t in fetch_tensors))
        # TF: "You must feed a value for placeholder tensor ..." - every placeholder the program reads, directly or through
        # a feed alias, has to be in the feed dict
        fed_ids = {id(ph) for ph in feeds}
        needed = {}
        for op in ops:
            for t in op.inputs:
                r = t.root()
                if isinstance(r, Placeholder):
                    needed[id(r)] = r
            # an input that arrives through a feed alias launches nothing and may sit behind an op the walk above pruned
            # (repeat_batch(placeholder) -> ConcatActionsOp): the placeholder behind it has to be fed all the same
            for t in getattr(op, 'fed_inputs', ()):
                src = op._fed_source(t) if hasattr(op, '_fed_source') else t.root()
                if isinstance(src, Placeholder):
                    needed[id(src)] = src
        prog.missing_feeds = sorted(r.name for i, r in needed.items() if i not in fed_ids)      # Session.run raises on these
        # feed aliases whose destination this program reads
        prog.alias_copies = {}
        for ph in feeds:
            for dst, c_off, tile in self.graph.feed_aliases.get(id(ph), ()):
                if id(dst) in skip_outputs:
                    # the destination is the result of a skipped op: whoever computed it wrote these channels too.  (It may be
                    # a window of a tensor this program does read - the pair generator's concatenation, whose first half is the
                    # batch-B instance's - and writing this placeholder's rows there would race with the pair's own feed.)
                    continue
                if id(dst) in read or id(dst.root()) in read:
                    self._materialize(dst)
                    prog.alias_copies.setdefault(id(ph), []).append((dst, c_off, tile))
        prog.eager = any(getattr(op, 'no_graph', False) for op in ops)     # an op that cannot be captured: eager launch list
        return prog

    @staticmethod
    def _hoist_side_chains(ops):
        """Side-chain ops (Graph.side_branch) move to the earliest position their producers allow, keeping their own
        order: the chain is then forked from the main stream as soon as its inputs exist and overlaps everything the main
        chain does after that point, instead of waiting for the main ops that merely precede it in creation order."""
        if not any(o.side_stream and not getattr(o, 'is_collective', False) for o in ops):
            return ops
        out, pos_floor = [], 0
        for op in ops:
            if not op.side_stream or getattr(op, 'is_collective', False) or op.run_last:
                out.append(op)
                continue
            where = {id(o): k for k, o in enumerate(out)}
            deps = [t.op for t in op.inputs if t.op is not None] + list(op.control_inputs)
            p = max([where[id(d)] + 1 for d in deps if id(d) in where] + [pos_floor])
            out.insert(p, op)
            pos_floor = p + 1
        return out

    @staticmethod
    def _fold_clips(ops):
        """Weight clips fetched together with an optimizer step over the same flat buffer are fused
        into that step's kernel (update -> clip; the reference leaves the order undefined, D6)."""
        clips = [o for o in ops if getattr(o, 'is_clip', False)]
        steps = [o for o in ops if getattr(o, 'is_optimizer_step', False)]
        for st in steps:
            st.program_clip = None
        if not clips or not steps:
            return ops
        for st in steps:
            mine = [c for c in clips if c.var.scope == st.scope]
            names = {c.var.name for c in mine}
            bounds = {(c.lo, c.hi) for c in mine}
            if mine and names == set(st.var_names) and len(bounds) == 1:
                st.program_clip = next(iter(bounds))
                ops = [o for o in ops if o not in mine]
        return ops

    def _feed(self, prog, feed_dict):
        """Write the fed values where ``prog`` reads them: the placeholder itself when the program uses it, plus every
        feed alias whose destination the program reads (concatenations written by the feed, tiled action channels).
        Device-resident float32 feeds go through ONE acg_copy_many launch.  Used by run() and profile_ops()."""
        fused = []
        # host arrays (the reference's numpy feed_dict): ONE upload for all of them, each fed ARRAY once however many
        # placeholders take it (upload_many); values no target of this program reads are not uploaded
        uploaded = {}          # id(host value) -> its device copy
        if self.rt.is_cuda:
            todo = {}
            for ph, val in feed_dict.items():
                if (id(ph) in prog.used_feeds or prog.alias_copies.get(id(ph))) and id(val) not in todo:
                    if torch.is_tensor(val):
                        host = (not val.is_cuda) and val.dtype in (torch.float32, torch.float64)
                    else:
                        host = np.asarray(val).dtype in (np.float32, np.float64)
                    if host:
                        todo[id(val)] = val
            if todo:
                uploaded = dict(zip(todo.keys(), self.upload_many(list(todo.values()))))
        for ph, val in feed_dict.items():
            targets = ([(ph, 0, None)] if id(ph) in prog.used_feeds else []) + prog.alias_copies.get(id(ph), [])
            if not targets:
                # fed but not read by this program (train_d feeds next_state like the reference, train.py:137-142): no upload -
                # a host array would cost a staged copy on the stream between two programs
                if tuple(np.shape(val)) != tuple(ph.shape if ph.valid_c is None else ph.shape[:-1] + (ph.valid_c,)):
                    raise ValueError('Cannot feed value of shape %s for %r' % (tuple(np.shape(val)), ph))
                continue
            if id(val) in uploaded:
                # then the same fused device copy as device-resident feeds - a feed may have several destinations (feed aliases),
                # and strided host -> device copies of each of them cost milliseconds.  The Trainer feeds the same frames to two
                # placeholders (the dense one and the channel-padded one the first conv gathers): uploaded once (round 5)
                src = uploaded[id(val)]
            else:
                src = val if torch.is_tensor(val) else torch.from_numpy(np.ascontiguousarray(val))
            dst = ph.buf if ph.valid_c is None else ph.buf[..., :ph.valid_c]     # pad channels stay zero
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError('Cannot feed value of shape %s for %r' % (tuple(src.shape), ph))
            cols = src.shape[-1]
            for t, c_off, tile in targets:
                if (self.rt.is_cuda and src.is_cuda and src.device == t.buf.device and src.dtype == torch.float32
                        and t.dtype in (torch.float32, torch.bfloat16) and src.is_contiguous() and len(fused) < _lib.COPY_MAX):
                    fused.append((src, t, c_off, tile))    # device-resident feeds: one launch for all of them (below)
                else:
                    rows2d = src.reshape(-1, cols)
                    if tile is not None:                   # (r // div) % mod: the tiled action vector
                        idx = (torch.arange(t.numel // t.shape[-1], device=rows2d.device) // tile[0]) % tile[1]
                        rows2d = rows2d[idx]
                    t.buf.view(-1, t.shape[-1])[:, c_off:c_off + cols].copy_(rows2d.to(t.dtype), non_blocking=True)
        if fused:
            cl = _lib.CopyList()
            for i, (src, t, c_off, tile) in enumerate(fused):
                cols = src.shape[-1]
                cl.src[i], cl.dst[i] = src.data_ptr(), t.buf.data_ptr() + c_off * t.buf.element_size()
                cl.rows[i] = src.numel() // cols if tile is None else t.numel // t.shape[-1]
                cl.cols[i], cl.dst_pitch[i], cl.dst_dtype[i] = cols, t.shape[-1], _lib.code(t.dtype)
                cl.src_div[i], cl.src_mod[i] = (0, 0) if tile is None else tile
            self.rt.lib.copy_many(ctypes.byref(cl), len(fused), _lib.ACG_F32, self.rt.stream_ptr())

    # ---- run
    def run(self, fetches, feed_dict=None, device_fetch=False, skip=None):
        """``skip`` (a frozenset of Ops, an extension with no TensorFlow counterpart): ops whose outputs another program of this
        session has already computed into their tensors; the program compiled for this call neither launches them nor anything only
        they need (see _compile).  The caller vouches for those tensors' contents."""
        feed_dict = feed_dict or {}
        single = not isinstance(fetches, (list, tuple))
        flat = self._flatten(fetches)
        if any(isinstance(f, InitOp) for f in flat):
            self._initialize()
            return None if single else [None] * len(flat)
        key = (tuple(id(f) for f in flat), tuple(id(k) for k in feed_dict)) + ((id(skip),) if skip else ())
        prog = self._programs.get(key)
        if prog is None:
            prog = self._compile(flat, list(feed_dict.keys()), tuple(skip) if skip else ())
            prog.skip_ref = skip          # keeps the frozenset (whose id is part of the key) alive
            self._programs[key] = prog
        if prog.missing_feeds:
            raise ValueError('You must feed a value for placeholder tensor(s) %s' % ', '.join(prog.missing_feeds))
        self._feed(prog, feed_dict)
        if self._weights_dirty and prog is not None:
            self._refresh_weight_copies()
        self._execute(prog)
        results = []
        for t in prog.fetch_tensors:
            if t is None:
                results.append(None)
            elif device_fetch:
                results.append(t.buf)
            else:
                results.append(t.buf.detach().float().cpu().numpy().copy())
        if single:
            return results[0]
        return self._unflatten(fetches, iter(results))

    def profile_ops(self, fetches, feed_dict=None, repeats=3, relaunch=None, in_graph=10, skip=None):
        """Instrumented pass of a fetch.  Returns [(op, mean milliseconds)] in program order.  Executes the program
        ``repeats`` times for real (optimizer steps included), every device op bracketed by events recorded on the
        launch stream.  An event pair around ONE eager launch also times the gap the event records themselves open
        (~4 us here, more than some kernels), so the ops accepted by ``relaunch(op)`` - which must be idempotent - are
        timed the way the training step runs them instead: ``in_graph`` launches of the op captured into a small HIP
        graph, replayed once untimed and once between two events; the op's time is that replay / in_graph - kernel(s)
        plus one in-graph launch boundary, on the launch stream."""
        if not self.rt.is_cuda:
            raise RuntimeError('profile_ops needs a GPU session')
        flat = self._flatten(fetches)
        key = (tuple(id(f) for f in flat), tuple(id(k) for k in (feed_dict or {}))) + ((id(skip),) if skip else ())
        if key not in self._programs:
            self.run(fetches, feed_dict, skip=skip)
        prog = self._programs[key]
        self._feed(prog, feed_dict or {})      # exactly what run() writes: used feeds, feed aliases, tiled action channels
        if self._weights_dirty:
            self._refresh_weight_copies()
        stream = torch.cuda.current_stream(self.rt.device)
        records, graphs = [], {}
        for rep in range(repeats):
            for kind, seg in prog.segments:
                if kind == 'host':
                    seg()
                    continue
                sp = self.rt.stream_ptr()
                for op, fn in seg:
                    if relaunch is not None and relaunch(op):
                        fn(sp)                                   # the launch that belongs to the program
                        if id(op) not in graphs:
                            torch.cuda.synchronize(self.rt.device)
                            gr = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(gr, capture_error_mode='thread_local'):
                                csp = self.rt.stream_ptr()
                                for _ in range(in_graph):
                                    fn(csp)
                            gr.replay()
                            graphs[id(op)] = gr
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        graphs[id(op)].replay()
                        e1.record(stream)
                        records.append((op, e0, e1, 1.0 / in_graph))
                    else:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        fn(sp)
                        e1.record(stream)
                        records.append((op, e0, e1, 1.0))
        torch.cuda.synchronize(self.rt.device)
        totals, order = {}, []
        for op, e0, e1, scale in records:
            if id(op) not in totals:
                totals[id(op)] = [op, 0.0]
                order.append(id(op))
            totals[id(op)][1] += e0.elapsed_time(e1) * scale
        return [(totals[i][0], totals[i][1] / repeats) for i in order]

    def _unflatten(self, fetches, it):
        if isinstance(fetches, (list, tuple)):
            return [self._unflatten(f, it) for f in fetches]
        return next(it)

    def _launch_segment(self, seg):
        """Enqueue one device segment.  Ops flagged ``side_stream`` (the gradient all-reduces of ``collectives='side'``;
        the chains built inside Graph.side_branch) go to the session's second HIP stream.  A side op is ordered behind the
        main stream by ONE edge at the point where it first needs something the side stream has not seen (always for the
        first one: that edge also pulls the side stream into a capture), and the side stream is joined back in front of
        the first main op that reads a side result, waits on one (control input) or is flagged ``joins_side`` (the
        optimizer steps and the deferred weight-gradient reduction, which consume what side ops left in buffers).
        Fork and join are captured into the program's HIP graph like any other dependency: a parallel branch."""
        rt = self.rt
        use_side = rt.is_cuda and any(op.side_stream and (self.side_branches or getattr(op, 'is_collective', False)) for op, _ in seg)
        if not use_side:
            sp = rt.stream_ptr()
            for _, fn in seg:
                fn(sp)
            return
        main = torch.cuda.current_stream(rt.device)
        side = rt.side_stream
        sp_main = ctypes.c_void_p(main.cuda_stream)
        sp_side = ctypes.c_void_p(side.cuda_stream)
        pending = set()                 # ids of side-stream ops not yet joined into the main stream
        unseen = set()                  # ids of main-stream ops enqueued since the last fork edge
        forked = False
        edges = iter(rt.edge_pool(2 * len(seg) + 2))
        edge = rt.lib.stream_edge

        def deps(op):
            return [t.op for t in op.inputs if t.op is not None] + list(op.control_inputs)

        def launched(op):
            """ids of every op whose result this launch produced: a paired input gradient also runs its layer's weight
            gradient (ops.ConvDgradOp), a conv whose epilogue applied bias + activation wrote its BiasActOp's output (that
            op's bind returns None, so it never appears in the segment itself: a consumer's deps() name it all the same)."""
            ids = [id(op)]
            if getattr(op, 'pair_active', False):
                ids.append(id(op.pair_w))
            ids.extend(id(a) for a in getattr(op, 'absorbed', ()))
            return tuple(ids)

        def join():
            edge(next(edges), sp_side, sp_main)
            pending.clear()
        for op, fn in seg:
            if op.side_stream and (self.side_branches or getattr(op, 'is_collective', False)):
                if not forked or (unseen and getattr(op, 'is_collective', False)) or any(id(d) in unseen for d in deps(op)):
                    edge(next(edges), sp_main, sp_side)
                    forked = True
                    unseen.clear()
                fn(sp_side)
                pending.update(launched(op))
            else:
                if pending and (getattr(op, 'joins_side', False) or any(id(d) in pending for d in deps(op))):
                    join()
                fn(sp_main)
                unseen.update(launched(op))
        if pending:
            join()

    def _execute(self, prog):
        rt = self.rt
        prog.runs += 1
        if not self.use_hip_graphs or prog.runs == 1 or prog.eager:
            # eager launch list (first run of a program is always eager: it also warms every kernel)
            for kind, seg in prog.segments:
                if kind == 'host':
                    seg()
                else:
                    self._launch_segment(seg)
            return
        if prog.graphs is None:
            graphs = []
            if rt.is_cuda:
                torch.cuda.synchronize(rt.device)
            try:
                for kind, seg in prog.segments:
                    graphs.append(None if kind == 'host' else self.capture_segment(seg))
            except Exception as e:
                # ONLY "this stack cannot capture that launch" falls back to eager launches (said once, per program).  A bad
                # kernel argument (AcgError), an unbalanced fork / join ('capture not joined') or any other failure inside
                # the body is a bug and is raised: with several ranks a silent fallback would leave one rank eager beside
                # replaying peers and show up as nothing but a line on stderr
                if not self._capture_unsupported(e):
                    raise
                import sys
                sys.stderr.write('[acgan] HIP-graph capture is not supported for this program (%s: %s); it is launched '
                                 'eagerly from now on\n' % (type(e).__name__, str(e).splitlines()[0] if str(e) else ''))
                if rt.is_cuda:
                    torch.cuda.synchronize(rt.device)
                    if rt.side_stream is not None and rt.side_stream.is_capturing():
                        raise RuntimeError('the side stream is still capturing after a failed capture: cannot launch eagerly') from e
                prog.eager = True
                for kind, seg in prog.segments:
                    if kind == 'host':
                        seg()
                    else:
                        self._launch_segment(seg)
                return
            prog.graphs = graphs
        for (kind, seg), gr in zip(prog.segments, prog.graphs):
            if kind == 'host':
                seg()
            else:
                gr.replay()

    def capture_segment(self, seg):
        """One device segment captured into a HIP graph; returns an object with ``replay()``.  (Tests replace this to
        exercise the capture-failure path of _execute without a GPU.)"""
        gr = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of OTHER threads (none of ours; a host library's helper thread at worst) neither see nor
        # invalidate this thread's capture
        with torch.cuda.graph(gr, capture_error_mode='thread_local'):
            self._launch_segment(seg)     # current stream = the capture stream
        return gr

    _CAPTURE_UNSUPPORTED = (
        'StreamCaptureUnsupported', 'operation not permitted when stream is capturing',       # hipError 900
        'StreamCaptureInvalidated', 'operation failed due to a previous error during capture',  # 901 (follows a 900)
        'StreamCaptureImplicit', 'disallowed implicit dependency',                             # 906
    )

    @classmethod
    def _capture_unsupported(cls, exc):
        """True only for the HIP runtime's "this call cannot be captured" family.  The sequencing errors of a capture
        (hipErrorStreamCaptureUnjoined / Unmatched / Merge / Isolation / WrongThread: an unbalanced fork or join in
        _launch_segment) and errors of our own library (AcgError: a bad kernel argument) are bugs, not stack limits."""
        if isinstance(exc, _lib.AcgError) or not isinstance(exc, RuntimeError):
            return False
        from .comm import CommError
        if isinstance(exc, CommError):
            # a COLLECTIVE refused while its stream was capturing (RCCL wraps whatever the runtime told it into "unhandled cuda
            # error" / "invalid usage"): the same stack limit seen through RCCL.  The program is launched eagerly instead - if
            # the communicator itself is broken, the first eager ncclAllReduce says so, outside any capture
            return 'ncclAllReduce failed' in str(exc)
        seen = set()
        while exc is not None and id(exc) not in seen:       # (capture_end raising on top of the first error: look at both)
            seen.add(id(exc))
            if isinstance(exc, RuntimeError) and not isinstance(exc, _lib.AcgError) and any(k in str(exc) for k in cls._CAPTURE_UNSUPPORTED):
                return True
            exc = exc.__cause__ or exc.__context__
        return False


@contextlib.contextmanager
def default_graph(graph=None):
    g = graph or Graph()
    _default.append(g)
    try:
        yield g
    finally:
        _default.pop()
