"""Optimizers, the discriminator weight clip and the data-parallel gradient all-reduce.

``AdamOptimizer(lr, name).minimize(loss, var_list)`` / ``RMSPropOptimizer`` keep the reference's call
shape (train.py:91-102) and TensorFlow-1.0's update formulas.  ``minimize``:

1. lays the scope's variables out in one flat parameter buffer and allocates one flat gradient buffer
   with the same layout (graph.py), so wgrad kernels write gradients in place;
2. builds the backward ops (graph.build_gradients) - only along paths that reach ``var_list``;
3. with world_size > 1, cuts the flat gradient buffer into contiguous buckets in backward-completion
   order and inserts one RCCL all-reduce per bucket right after the op that completes it - on the
   compute stream in program order, or on a side HIP stream overlapping the rest of backward
   (``collectives='side'``; see DataParallel) - captured into the step's HIP graph either way; the
   optimizer step runs behind all of them;
4. appends ONE fused update launch over the flat buffers (+ the weight clip when it is fetched with it).
"""
import torch

from . import graph as G
from . import ops as O

_p = G._ptr


def _f32(v):
    """TF-1.0 holds optimizer hyper-parameters as float32 constants."""
    import numpy as np
    return float(np.float32(v))


class DataParallel:
    """Per-graph data-parallel configuration (one process per GPU; this process's own RCCL communicator, comm.py)."""

    def __init__(self, world_size=1, n_buckets=None, force=False, sync_bn=False, exact_global_batch=False, collectives='stream'):
        self.world_size, self.force = int(world_size), bool(force)
        # Every bucket's all-reduce is a plain ncclAllReduce on a HIP stream, captured into the step's HIP graph:
        # 'stream': on the compute stream, in program order right behind the op that completes the bucket;
        # 'side':   on the session's second stream, behind one fork edge, overlapping the rest of backward; one join
        #           edge in front of the optimizer update (north_star: "overlapped with backward on a side HIP stream").
        if collectives not in ('stream', 'side'):
            raise ValueError("collectives must be 'stream' or 'side'")
        self.collectives = collectives
        # buckets per optimizer: an in-order all-reduce hides nothing, so ONE large message per optimizer is best there;
        # the side-stream form sends the layers that finish first (the last-created, for D 69 % of the bytes) while the
        # rest of backward still runs, in two messages
        self.n_buckets = int(n_buckets) if n_buckets else (1 if collectives == 'stream' else 2)
        # exact_global_batch: the run reproduces ONE device at the global batch - BatchNorm over the global batch,
        # the GDL sum scaled by the world size, the state-loss norm taken over all ranks (SURVEY 8(e) caveats 1-3)
        self.exact_global_batch = bool(exact_global_batch)
        self.sync_bn = bool(sync_bn) or self.exact_global_batch   # BatchNorm over the GLOBAL batch (ops.batch_norm)

    @property
    def active(self):
        return self.world_size > 1 or self.force


def set_data_parallel(world_size, n_buckets=None, graph=None, force=False, sync_bn=False, exact_global_batch=False, collectives='stream'):
    """``force`` inserts the bucketed all-reduce even at world_size 1 (a one-rank communicator): lets a single GPU
    exercise the collective / side-stream / graph-capture machinery the multi-GPU runs depend on."""
    (graph or G.get_default_graph()).collections['data_parallel'] = DataParallel(world_size, n_buckets, force, sync_bn, exact_global_batch, collectives)


def _dp(graph):
    return graph.collections.get('data_parallel') or DataParallel(1)


class AllReduceOp(G.Op):
    """Sum-all-reduce of one contiguous gradient bucket, in place: ``ncclAllReduce`` on the stream the launch list
    hands it - the compute stream (collectives='stream') or the session's side stream (collectives='side',
    ``side_stream`` makes graph._launch_segment fork and join around it).  A device op like any kernel launch."""

    is_collective = True     # never part of a hoisted side chain; forks behind EVERY main op enqueued so far (its bucket's writers)
    joins_side = True        # on the main stream: side-chain weight gradients of its bucket must have landed

    def __init__(self, flat_grad, start, end, after, name, side):
        super().__init__(flat_grad.graph, name, [], [], control_inputs=after)
        self.flat_grad, self.start, self.end = flat_grad, start, end
        self.index = max(o.index for o in after) + 0.5     # right behind the op that completes the bucket
        self.side_stream = bool(side)

    def bind(self, rt):
        comm = rt.comm
        view = self.flat_grad.buf[self.start:self.end]
        self.no_graph = rt.is_cuda and not comm.capturable
        if hasattr(comm, 'all_reduce_ptr'):
            from . import comm as C
            ptr, n, call = _p(view), view.numel(), comm.all_reduce_ptr
            return lambda s: call(ptr, n, C.NCCL_FLOAT32, C.NCCL_SUM, s)
        return lambda s: comm.all_reduce(view)


class StepOp(G.Op):
    """One fused optimizer launch over the flat buffers of a scope."""
    is_optimizer_step = True
    joins_side = True        # reads every gradient of its scope, whichever stream produced it

    def __init__(self, opt, scope, var_names, flat_param, flat_grad, slots, deps, grad_scale):
        super().__init__(flat_param.graph, opt.name + '/update', [flat_param, flat_grad] + slots, [], control_inputs=deps)
        self.opt, self.scope, self.var_names, self.grad_scale = opt, scope, list(var_names), grad_scale
        self.program_clip = None
        self.extras = [t for t3 in flat_param.graph.weight_copies.get(scope, []) for t in t3[1:]]

    def bind(self, rt):
        # bf16 pipeline: the conv kernels read bf16 copies of the filters, which follow the update.  Round 4: update and
        # refresh are ONE launch (acg_opt_step_prepare_bf16: the blocks that write a filter's copies update its elements) where
        # every copy of the scope fits one list; else two launches as before
        fused = self.opt._bind_step_prepared(rt, self, self.program_clip) if rt.fuse_weight_refresh else None
        if fused is not None:
            return fused
        step = self.opt._bind_step(rt, self, self.program_clip)
        prep = O.prepare_weights_launch(rt, self.graph, self.scope)
        if prep is None:
            return step

        def launch(s):
            step(s)
            prep(s)
        return launch


class ClipOp(G.Op):
    """p.assign(tf.clip_by_value(p, lo, hi)) (train.py:89).  Ordered after the update of the same
    program (defect D6), and folded into the optimizer kernel when it covers the optimizer's scope."""
    joins_side = True
    is_clip = True
    run_last = True

    def __init__(self, var, lo, hi):
        super().__init__(var.graph, 'clip/' + var.name, [var], [])
        self.var, self.lo, self.hi = var, _f32(lo), _f32(hi)

    def bind(self, rt):
        args = (_p(self.var.buf), self.var.numel, self.lo, self.hi)
        fn = rt.lib.clip
        if getattr(self.var, 'copies', None) is None:
            return lambda s: fn(*args, s)
        prep = O.prepare_weights_launch(rt, self.graph, self.var.scope)    # a standalone clip changes a filter: refresh

        def launch(s):
            fn(*args, s)
            prep(s)
        return launch


def clip_by_value_assign(var, lo, hi):
    return ClipOp(var, lo, hi)


class Optimizer:
    def __init__(self, learning_rate, name):
        self.lr, self.name = _f32(learning_rate), name

    def _make_slots(self, graph, total):
        raise NotImplementedError

    def _bind_step(self, rt, step_op, clip):
        raise NotImplementedError

    def _opt_args(self, rt, op, clip):
        """-> (kind, (lr, beta1 | decay, beta2, eps), slot1, slot2 | None, step counter | None, launch that must precede | None)"""
        raise NotImplementedError

    def _bind_step_prepared(self, rt, op, clip):
        """One launch for the update of the scope and the refresh of its bf16 filter copies, or None (no copies: a float32
        graph; more filters than one list holds)."""
        import ctypes
        from . import _lib
        entries = op.graph.weight_copies.get(op.scope) or []
        if not entries or len(entries) > _lib.PREP_MAX:
            return None
        kind, fields, s1, s2, step, before = self._opt_args(rt, op, clip)
        pl = _lib.PrepList()
        for i, (w, rm, tr) in enumerate(entries):
            kh, kw, a, b = w.shape
            pl.src[i], pl.rm[i], pl.tr[i] = w.buf.data_ptr(), rm.buf.data_ptr(), tr.buf.data_ptr()
            pl.taps[i], pl.a[i], pl.b[i] = kh * kw, a, b
        lo, hi = clip if clip else (0.0, 0.0)
        oa = _lib.OptArgs(kind, fields[0], fields[1], fields[2], fields[3], op.grad_scale, 1 if clip else 0, lo, hi)
        p, g = op.inputs[0], op.inputs[1]
        args = (_p(p.buf), _p(g.buf), _p(s1.buf), _p(s2.buf) if s2 is not None else None, _p(step.buf) if step is not None else None,
                p.numel, ctypes.byref(oa), ctypes.byref(pl), len(entries))
        fn = rt.lib.opt_step_prepare_bf16

        def launch(s):
            if before is not None:
                before(s)
            fn(*args, s)
        launch._keep = (oa, pl)
        return launch

    def minimize(self, loss, var_list=None):
        if not isinstance(loss, O.Scalar):
            raise TypeError('minimize expects a loss built from this package\'s loss functions')
        g = G.get_default_graph()
        if var_list is None:
            var_list = g.trainable_variables()
        if not var_list:
            raise ValueError('No variables to optimize.')
        scopes = {v.scope for v in var_list}
        if len(scopes) != 1:
            raise ValueError('minimize: var_list must come from one top-level scope, got %s' % sorted(scopes))
        scope = scopes.pop()
        offsets, total, flat_param = g.layout(scope)
        flat_grad = g.new_state((total,), 0.0, self.name + '/flat_grad')
        heads = {}
        for head, idx, w in loss.terms:
            heads.setdefault(head, {})
            heads[head][idx] = heads[head].get(idx, 0.0) + w
        ctx = G.build_gradients(g, heads, var_list, flat_grad, offsets, O._add)
        missing = [v.name for v in var_list if v.name not in ctx.writers]
        if len(missing) == len(var_list):
            raise ValueError('No gradients provided for any variable: %s' % missing)
        dp = _dp(g)
        buckets = self._buckets(dp, var_list, offsets, total) if dp.active else [(0, total)]
        reduce_ops = self._defer_wgrad_reductions(var_list, ctx, offsets, buckets)
        writers = [op for ops_ in ctx.writers.values() for op in ops_]
        deps = list(dict.fromkeys(writers))
        if dp.active:
            deps += self._insert_allreduce(var_list, ctx, flat_grad, offsets, buckets, dp.collectives == 'side')
        slots = self._make_slots(g, total)
        step_op = StepOp(self, scope, [v.name for v in var_list], flat_param, flat_grad, slots, deps, 1.0 / dp.world_size)
        step_op.reduce_ops = reduce_ops        # one of them can carry the step counter's increment (_step_inc_launch)
        return step_op

    @staticmethod
    def _buckets(dp, var_list, offsets, total):
        # Buckets are contiguous windows of the flat gradient buffer, cut from the END of the layout
        # (the last-created layers finish their wgrad first), each reduced as soon as its last writer ran.
        order = sorted(var_list, key=lambda v: offsets[v.name])
        target = max(total // max(dp.n_buckets, 1), 1)
        buckets, hi, acc = [], total, 0
        for v in reversed(order):
            acc = hi - offsets[v.name]
            if acc >= target:
                buckets.append((offsets[v.name], hi))
                hi, acc = offsets[v.name], 0
        if hi > 0:
            buckets.append((0, hi))
        return buckets

    def _defer_wgrad_reductions(self, var_list, ctx, offsets, buckets):
        """Per bucket (one bucket = the whole buffer without data parallelism), the conv weight gradients with a single
        writer hand their split-K slab reduction to ONE WgradReduceOp, which becomes the gradient's writer."""
        made = []
        for k, (lo, hi) in enumerate(buckets):
            group = [ctx.writers[v.name][0] for v in var_list
                     if lo <= offsets[v.name] < hi and len(ctx.writers.get(v.name, ())) == 1
                     and isinstance(ctx.writers[v.name][0], O.ConvWgradOp)]
            if len(group) < 2:
                continue
            red = O.WgradReduceOp(group, '%s/wgrad_reduce_%d' % (self.name, k))
            made.append(red)
            for v in var_list:
                if ctx.writers.get(v.name) and ctx.writers[v.name][0] in group:
                    ctx.writers[v.name] = [red]
        return made

    def _insert_allreduce(self, var_list, ctx, flat_grad, offsets, buckets, side):
        order = sorted(var_list, key=lambda v: offsets[v.name])
        reduces = []
        for k, (lo, hi_) in enumerate(buckets):
            after = list(dict.fromkeys(op for v in order if lo <= offsets[v.name] < hi_ for op in ctx.writers.get(v.name, [])))
            if not after:
                continue
            reduces.append(AllReduceOp(flat_grad, lo, hi_, after, '%s/allreduce_%d' % (self.name, k), side))
        return reduces


class AdamOptimizer(Optimizer):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, name='Adam'):
        super().__init__(learning_rate, name)
        self.b1, self.b2, self.eps = _f32(beta1), _f32(beta2), _f32(epsilon)

    def _make_slots(self, graph, total):
        return [graph.new_state((total,), 0.0, self.name + '/m'), graph.new_state((total,), 0.0, self.name + '/v'),
                graph.new_state((1,), 0, self.name + '/step', dtype=torch.int32)]

    @staticmethod
    def _step_inc_launch(rt, op, step):
        """The launch that advances the device step counter in front of the update, or None when a deferred weight-gradient
        reduction of this optimizer runs in the same program: that launch (acg_splitk_reduce_many, always ahead of the update on
        the main stream) then carries the increment in its acg_reduce_list::step_inc - one tiny launch less per step."""
        for red in getattr(op, 'reduce_ops', ()):
            lists = getattr(red, '_keep', None)
            # the hand-off is explicit: the reduce op must have been bound for THIS compile (WgradReduceOp.bind stamps the program
            # it built its lists for - a stale list of an earlier program, or a reduce op bound after this step op, never matches,
            # and the counter then gets its own launch below), and one list carries ONE counter
            if id(red) in rt.program_ops and lists and not red.side_stream and getattr(red, '_bound_for', None) is rt.program_ops:
                if red.carries_step_inc:
                    raise RuntimeError('%s already advances another optimizer step counter' % red.name)
                lists[0][0].step_inc = step.buf.data_ptr()
                red.carries_step_inc = True
                return None
        inc, ps = rt.lib.step_inc, _p(step.buf)
        return lambda s: inc(ps, s)

    def _bind_step(self, rt, op, clip):
        p, g, m, v, step = op.inputs
        lo, hi = clip if clip else (0.0, 0.0)
        adam, before = rt.lib.adam_step, self._step_inc_launch(rt, op, step)
        args = (_p(p.buf), _p(g.buf), _p(m.buf), _p(v.buf), _p(step.buf), p.numel, self.lr, self.b1, self.b2, self.eps,
                op.grad_scale, 1 if clip else 0, lo, hi)

        def launch(s):
            if before is not None:
                before(s)
            adam(*args, s)
        return launch

    def _opt_args(self, rt, op, clip):
        p, g, m, v, step = op.inputs
        return 0, (self.lr, self.b1, self.b2, self.eps), m, v, step, self._step_inc_launch(rt, op, step)


class RMSPropOptimizer(Optimizer):
    """tf.train.RMSPropOptimizer (momentum 0): ms starts at ONE; p -= lr*g/sqrt(ms+eps)."""

    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, name='RMSProp'):
        super().__init__(learning_rate, name)
        if momentum != 0.0:
            raise ValueError('RMSPropOptimizer: momentum is not used by the reference and is not implemented')
        self.decay, self.eps = _f32(decay), _f32(epsilon)

    def _make_slots(self, graph, total):
        return [graph.new_state((total,), 1.0, self.name + '/ms')]

    def _bind_step(self, rt, op, clip):
        p, g, ms = op.inputs
        lo, hi = clip if clip else (0.0, 0.0)
        args = (_p(p.buf), _p(g.buf), _p(ms.buf), p.numel, self.lr, self.decay, self.eps, op.grad_scale,
                1 if clip else 0, lo, hi)
        fn = rt.lib.rmsprop_step
        return lambda s: fn(*args, s)

    def _opt_args(self, rt, op, clip):
        p, g, ms = op.inputs
        return 1, (self.lr, self.decay, 0.0, self.eps), ms, None, None, None
