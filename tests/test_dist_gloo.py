"""Data-parallel path on CPU: world_size 2 over gloo, kernels from the C-oracle stand-in library.

Checks the construction the multi-GPU bench relies on (one process per GPU, bucketed all-reduce of the
flat gradient buffer placed behind the wgrads that complete each bucket, 1/world_size folded into the
fused optimizer step): after one D step the reduced buffer equals the SUM of the two ranks'
independent gradients, the weights equal a hand-computed RMSProp+clip update with the MEAN gradient,
and both ranks hold identical weights after a further G step."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASE = 'dna_k6_bce_rmsprop'


def _inputs(rank):
    rng = np.random.default_rng(100 + rank)
    return (rng.uniform(-1, 1, (2, 64, 64, 3)).astype(np.float32), rng.uniform(-1, 1, (2, 64, 64, 3)).astype(np.float32),
            rng.standard_normal((2, 10)).astype(np.float32), rng.standard_normal((2, 5)).astype(np.float32))


def _flat(t):
    return t.buf.detach().clone().cpu().numpy()


def _worker(rank, world, port, outdir, collectives='stream', lookahead=False):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    import train_cases as TC
    from oracle import cbind
    from action_conditioned_gans_amd import graph as G
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    sess, tr = TC.build_trainer(lambda **kw: G.Session(device='cpu', lib=cbind.load(), world_size=world, rank=rank, **kw),
                                CASE, world_size=world, collectives=collectives)
    x, y, a, s = _inputs(rank)
    out = {'d_param0': _flat(tr.d_opt_op.inputs[0])}
    tr.train_d(x, y, a, next_g=(x, a) if lookahead else None)     # look-ahead: this step's generator pass covers the G step's samples too
    out['d_grad'] = _flat(tr.d_opt_op.inputs[1])
    out['d_param1'] = _flat(tr.d_opt_op.inputs[0])
    tr.train_g(x, y, a, s)
    out['g_grad'] = _flat(tr.g_opt_op.inputs[1])
    out['g_param1'] = _flat(tr.g_opt_op.inputs[0])
    if world > 1:
        kinds = [type(o).__name__ for o in G.get_default_graph().ops]
        out['n_allreduce'] = np.array(kinds.count('AllReduceOp'))
        segs = [k for k, _ in sess._programs[next(iter(sess._programs))].segments]
        out['n_host_segments'] = np.array(segs.count('host'))
        out['eager'] = np.array(int(sess._programs[next(iter(sess._programs))].eager))
        # round 5: which BatchNorm launches take the two-launch kernels (Session._bn_flags): flags of the G step's program,
        # the last one compiled - forward launches, and backward launches by their position relative to the first all-reduce
        ops = [o for _, seg in sess._programs[list(sess._programs)[-1]].segments for o, _ in seg]
        first = next((i for i, o in enumerate(ops) if getattr(o, 'is_collective', False) and o.side_stream), len(ops))
        bwd = [(i, o.fwd.flags_bwd) for i, o in enumerate(ops) if hasattr(getattr(o, 'fwd', None), 'flags_bwd')]
        out['bn_fwd_flags'] = np.array([o.flags_fwd for o in ops if hasattr(o, 'flags_fwd')])
        out['bn_bwd_before'] = np.array([f for i, f in bwd if i < first])
        out['bn_bwd_after'] = np.array([f for i, f in bwd if i > first])
    if lookahead:
        out['n_programs'] = np.array(len(sess._programs))
        out['skipped'] = np.array(sum(1 for p in sess._programs.values() if getattr(p, 'skip_ref', None)))
    np.savez(os.path.join(outdir, 'w%d_r%d%s.npz' % (world, rank, '_la' if lookahead else '')), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
@pytest.mark.parametrize('collectives', ['stream', 'side'])
def test_two_rank_allreduce_matches_mean_gradient_update(collectives):
    """Both ways of issuing the bucketed all-reduce: one message per optimizer in program order, and two buckets flagged
    for the side stream (on the CPU stand-in both run in program order; the GPU forks / joins around them)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, collectives), nprocs=2, join=True)
        for r in (0, 1):                      # independent single-process references on each shard
            mp.spawn(_single, args=(r, d), nprocs=1, join=True)
        dp = [dict(np.load(os.path.join(d, 'w2_r%d.npz' % r))) for r in (0, 1)]
        ref = [dict(np.load(os.path.join(d, 'w1_r%d.npz' % r))) for r in (0, 1)]
    # the reduced buffer holds the SUM of per-rank gradients, identical on both ranks
    want_sum = ref[0]['d_grad'].astype(np.float64) + ref[1]['d_grad']
    scale = np.abs(want_sum).max()
    for r in (0, 1):
        assert np.abs(dp[r]['d_grad'] - want_sum).max() <= 1e-6 * scale
    assert np.array_equal(dp[0]['d_grad'], dp[1]['d_grad'])
    # RMSProp (ms starts at 1) + clip with the MEAN gradient (TF formulas)
    f32 = lambda v: float(np.float32(v))
    g = want_sum * 0.5
    ms = f32(0.9) * 1.0 + (1 - f32(0.9)) * g * g
    want_p = np.clip(dp[0]['d_param0'].astype(np.float64) - f32(5e-5) * g / np.sqrt(ms + f32(1e-10)), f32(-0.01), f32(0.01))
    for r in (0, 1):
        assert np.abs(dp[r]['d_param1'] - want_p).max() <= 2e-7
    # replicas stay bit-identical through the following G step
    assert np.array_equal(dp[0]['g_param1'], dp[1]['g_param1'])
    assert np.array_equal(dp[0]['g_grad'], dp[1]['g_grad'])
    # the all-reduces are device ops of the launch list (no host segments), in both modes
    assert int(dp[0]['n_host_segments']) == 0 and int(dp[0]['eager']) == 0
    # BatchNorm kernels beside a multi-rank ring kernel (Session._bn_flags): with side-stream collectives the backward launches
    # behind the first bucket's all-reduce take the two-launch path (flag 1), everything in front of it - all forward passes,
    # the late layers' backward passes - and every launch under in-order collectives keeps the one-launch kernels
    assert len(dp[0]['bn_fwd_flags']) >= 6 and not dp[0]['bn_fwd_flags'].any() and not dp[0]['bn_bwd_before'].any()
    if collectives == 'side':          # two buckets per optimizer
        assert int(dp[0]['n_allreduce']) >= 4
        assert len(dp[0]['bn_bwd_before']) >= 1 and len(dp[0]['bn_bwd_after']) >= 3 and dp[0]['bn_bwd_after'].all()
    else:
        assert len(dp[0]['bn_bwd_after']) == 0
        assert 2 <= int(dp[0]['n_allreduce']) <= 3          # one message per optimizer: D, G (and G pre-training)


@pytest.mark.timeout(900)
def test_two_rank_lookahead_step_matches_the_plain_data_parallel_step():
    """Data parallel + the look-ahead generator pass (round 5): the pair pass is rank-local forward work, the bucketed all-reduces
    sit where they sat.  Two gloo ranks, side-stream buckets: the reduced gradient buffers and the updated parameters of the
    look-ahead step equal those of the plain two-rank step to rounding, the replicas stay bit-identical, and both programs of the
    look-ahead step were compiled with a skip set."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, 'side', False), nprocs=2, join=True)
        mp.spawn(_worker, args=(2, _free_port(), d, 'side', True), nprocs=2, join=True)
        plain = [dict(np.load(os.path.join(d, 'w2_r%d.npz' % r))) for r in (0, 1)]
        ahead = [dict(np.load(os.path.join(d, 'w2_r%d_la.npz' % r))) for r in (0, 1)]
    for k in ('d_grad', 'g_grad', 'd_param1', 'g_param1'):
        assert np.array_equal(ahead[0][k], ahead[1][k]), k                       # replicas in step
        scale = max(float(np.abs(plain[0][k]).max()), 1e-6)
        assert float(np.abs(ahead[0][k].astype(np.float64) - plain[0][k]).max()) <= 2e-5 * scale, k
    assert int(ahead[0]['n_programs']) == 2 and int(ahead[0]['skipped']) == 2
    assert int(ahead[0]['n_host_segments']) == 0 and int(ahead[0]['eager']) == 0


def _single(_, rank, outdir):
    _worker(rank, 1, 0, outdir)


# ---- synchronised BatchNorm: two ranks x batch 2 == one process x batch 4 -------------------------------------------
SYNC_CASE = 'c1_plain_l1'       # plain generator, L1 only: the loss is a per-sample mean, so with global-batch BatchNorm
                                # statistics data parallel must reproduce the single-process global batch exactly


def _global_inputs():
    rng = np.random.default_rng(77)
    return (rng.uniform(-1, 1, (4, 64, 64, 3)).astype(np.float32), rng.uniform(-1, 1, (4, 64, 64, 3)).astype(np.float32),
            rng.standard_normal((4, 10)).astype(np.float32), rng.standard_normal((4, 5)).astype(np.float32))


def _sync_worker(rank, world, port, outdir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    import train_cases as TC
    from oracle import cbind
    from action_conditioned_gans_amd import graph as G
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    sess, tr = TC.build_trainer(lambda **kw: G.Session(device='cpu', lib=cbind.load(), world_size=world, rank=rank, **kw),
                                SYNC_CASE, world_size=world, sync_bn=True, batch=4 // world)
    lo, hi = rank * (4 // world), (rank + 1) * (4 // world)
    x, y, a, s = (t[lo:hi] for t in _global_inputs())
    frame = sess.run(tr.g_next_frame, tr._feed(x, y, a, s))
    tr.pretrain_g(x, y, a, s)
    out = {'frame': frame, 'grad': _flat(tr.g_pretrain_opt_op.inputs[1]), 'param': _flat(tr.g_pretrain_opt_op.inputs[0])}
    if world > 1:
        kinds = [type(o).__name__ for o in G.get_default_graph().ops]
        out['n_moment_reduces'] = np.array(kinds.count('BnMomentsAllReduceOp'))
        out['n_sum_reduces'] = np.array(kinds.count('BnSumsAllReduceOp'))
    np.savez(os.path.join(outdir, 'sync_w%d_r%d.npz' % (world, rank)), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sync_batchnorm_reproduces_the_global_batch():
    """SURVEY 8(e) caveat 1: with set_data_parallel(..., sync_bn=True) every BatchNorm uses the statistics of the
    global batch (one small all-reduce per layer and direction).  Two ranks with two samples each must then give the
    frames, the (averaged) gradient and the updated weights of ONE process running all four samples."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_sync_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        mp.spawn(_sync_worker, args=(1, 0, d), nprocs=1, join=True)
        dp = [dict(np.load(os.path.join(d, 'sync_w2_r%d.npz' % r))) for r in (0, 1)]
        ref = dict(np.load(os.path.join(d, 'sync_w1_r0.npz')))
    assert dp[0]['n_moment_reduces'] >= 7 and dp[0]['n_sum_reduces'] >= 7          # the plain generator has 7 BatchNorm layers
    frames = np.concatenate([dp[0]['frame'], dp[1]['frame']])
    assert np.abs(frames - ref['frame']).max() <= 2e-5 * max(np.abs(ref['frame']).max(), 1.0)
    mean_grad = dp[0]['grad'].astype(np.float64) / 2.0                              # the buffer holds the SUM over ranks
    assert np.array_equal(dp[0]['grad'], dp[1]['grad'])
    scale = np.abs(ref['grad']).max()
    assert np.abs(mean_grad - ref['grad']).max() <= 2e-4 * scale, np.abs(mean_grad - ref['grad']).max() / scale
    assert np.array_equal(dp[0]['param'], dp[1]['param'])


# ---- exact global batch: SyncBN + GDL scaling + global state-loss norm, the full adversarial DNA step ----------------
EXACT_CASE = 'dna_k6_bce_rmsprop'


def _exact_worker(rank, world, port, outdir, exact=True):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    import train_cases as TC
    from oracle import cbind
    from action_conditioned_gans_amd import graph as G
    if world > 1:
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    sess, tr = TC.build_trainer(lambda **kw: G.Session(device='cpu', lib=cbind.load(), world_size=world, rank=rank, **kw),
                                EXACT_CASE, world_size=world, exact=exact, batch=4 // world)
    lo, hi = rank * (4 // world), (rank + 1) * (4 // world)
    x, y, a, s = (t[lo:hi] for t in _global_inputs())
    tr.train_d(x, y, a)
    out = {'d_grad': _flat(tr.d_opt_op.inputs[1]), 'd_param': _flat(tr.d_opt_op.inputs[0])}
    tr.train_g(x, y, a, s)
    out['g_grad'] = _flat(tr.g_opt_op.inputs[1])
    out['g_param'] = _flat(tr.g_opt_op.inputs[0])
    np.savez(os.path.join(outdir, 'exact%d_w%d_r%d.npz' % (int(exact), world, rank)), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_exact_global_batch_mode_reproduces_one_device():
    """SURVEY 8(e) caveats 1-3 together (set_data_parallel(..., exact_global_batch=True)): BatchNorm over the global
    batch, the GDL sum scaled by the world size, the state-loss norm over all ranks.  One adversarial D step + G step of
    the DNA model on two ranks x two samples must match ONE process on the four samples: averaged gradients and
    RMSProp-updated weights (the D input of the G step already depends on the first update)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_exact_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        mp.spawn(_exact_worker, args=(1, 0, d), nprocs=1, join=True)
        mp.spawn(_exact_worker, args=(2, _free_port(), d, False), nprocs=2, join=True)      # control: conventional DP
        dp = [dict(np.load(os.path.join(d, 'exact1_w2_r%d.npz' % r))) for r in (0, 1)]
        ref = dict(np.load(os.path.join(d, 'exact1_w1_r0.npz')))
        plain = dict(np.load(os.path.join(d, 'exact0_w2_r0.npz')))
    for key in ('d_grad', 'g_grad'):
        assert np.array_equal(dp[0][key], dp[1][key])
        mean = dp[0][key].astype(np.float64) / 2.0
        scale = np.abs(ref[key]).max()
        assert np.abs(mean - ref[key]).max() <= 5e-4 * scale, (key, np.abs(mean - ref[key]).max() / scale)
    for key in ('d_param', 'g_param'):
        assert np.array_equal(dp[0][key], dp[1][key])
        assert np.abs(dp[0][key] - ref[key]).max() <= 1e-5 * max(np.abs(ref[key]).max(), 1.0), key
    # the control shows the test bites: conventional data parallel (per-replica BatchNorm, unscaled GDL, local norm) is
    # a different, if equally legitimate, computation
    off = np.abs(plain['g_grad'].astype(np.float64) / 2.0 - ref['g_grad']).max() / np.abs(ref['g_grad']).max()
    assert off > 1e-2, off


# ---- bootstrap of the RCCL communicator (comm.py): the 128-byte unique id travels over CPU channels only ------------
def _uid_worker(rank, world, port, outdir, via):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from action_conditioned_gans_amd import comm as C
    # the same pack -> share -> unpack path RcclCommunicator.__init__ takes, on an id shaped like a real one: an 8-byte
    # magic, then a sockaddr_in (AF_INET = 02 00 ...) - NUL bytes from byte 9 on (a c_char field would cut it there)
    mine = C._UniqueId()
    if rank == 0:
        C.unpack_unique_id(_UID_PATTERN, mine)
    uid = C.pack_unique_id(mine) if rank == 0 else None
    if via == 'gloo':
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
        got = C._share_unique_id(uid, world, rank, None)
        dist.barrier()
        dist.destroy_process_group()
    else:       # no process group at all: a TCPStore at MASTER_ADDR / MASTER_PORT
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        os.environ.pop('TORCHELASTIC_USE_AGENT_STORE', None)
        got = C._share_unique_id(uid, world, rank, None)
    C.unpack_unique_id(got, mine)
    with open(os.path.join(outdir, 'uid_%s_%d' % (via, rank)), 'wb') as f:
        f.write(C.pack_unique_id(mine))


_UID_PATTERN = bytes([0x5a, 0x7e, 0x11, 0xc3, 0x9d, 0x02, 0xee, 0x41, 0x02, 0x00, 0x9c, 0x40, 127, 0, 0, 1] + [0] * 8
                     + [(7 * i) & 0xff if i % 5 else 0 for i in range(104)])
assert len(_UID_PATTERN) == 128 and _UID_PATTERN.index(0) == 9


def test_unique_id_pack_unpack_keeps_all_128_bytes():
    """`bytes(uid.internal)` on a c_char field stopped at the first NUL.  The struct now round-trips
    whole, and a short or non-bytes delivery is a CommError instead of a 128-byte read out of a short object."""
    from action_conditioned_gans_amd import comm as C
    uid = C.unpack_unique_id(_UID_PATTERN)
    assert C.pack_unique_id(uid) == _UID_PATTERN
    assert bytes(bytearray(uid.internal)) == _UID_PATTERN
    for bad in (_UID_PATTERN[:9], _UID_PATTERN + b'x', None, 'x' * 128):
        with pytest.raises(C.CommError):
            C.unpack_unique_id(bad)


@pytest.mark.parametrize('via', ['gloo', 'store'])
def test_rccl_unique_id_bootstrap_over_cpu_channels(via):
    """What every rank needs before ncclCommInitRank: rank 0's unique id, delivered through an initialised gloo group
    or, without any process group, a TCPStore - never through ProcessGroupNCCL."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_uid_worker, args=(2, _free_port(), d, via), nprocs=2, join=True)
        for r in (0, 1):
            assert open(os.path.join(d, 'uid_%s_%d' % (via, r)), 'rb').read() == _UID_PATTERN, (via, r)


# ---- capture failure on ONE rank of two (graph._execute's fallback): no deadlock, identical weights ------------------
class _FakeGraph:
    """Stands in for a captured HIP graph on the CPU: replaying it launches the segment."""

    def __init__(self, sess, seg):
        self.sess, self.seg, self.replays = sess, seg, 0

    def replay(self):
        self.replays += 1
        self.sess._launch_segment(self.seg)


def _capfail_worker(rank, world, port, outdir, fail_rank):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    import train_cases as TC
    from oracle import cbind
    from action_conditioned_gans_amd import graph as G
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    sess, tr = TC.build_trainer(lambda **kw: G.Session(device='cpu', lib=cbind.load(), world_size=world, rank=rank, **kw),
                                CASE, world_size=world, collectives='side')
    captured = []
    if fail_rank is not None:
        sess.use_hip_graphs = True               # the CPU session never captures by itself: route _execute through the hook

        def capture(seg):
            if rank == fail_rank:
                raise RuntimeError('HIP error: operation not permitted when stream is capturing')
            captured.append(_FakeGraph(sess, seg))
            return captured[-1]
        sess.capture_segment = capture
    x, y, a, s = _inputs(rank)
    for _ in range(3):                            # run 1 eager, run 2 captures (or falls back), run 3 replays (or stays eager)
        tr.train_d(x, y, a)
        tr.train_g(x, y, a, s)
    progs = list(sess._programs.values())
    out = {'d_param': _flat(tr.d_opt_op.inputs[0]), 'g_param': _flat(tr.g_opt_op.inputs[0]),
           'eager': np.array([int(p.eager) for p in progs]), 'replays': np.array(sum(g.replays for g in captured))}
    np.savez(os.path.join(outdir, 'capfail%s_r%d.npz' % ('' if fail_rank is None else '_f%d' % fail_rank, rank)), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_capture_failure_on_one_rank_keeps_the_ranks_in_step():
    """rank 1's programs cannot be captured (the runtime refuses: hipErrorStreamCaptureUnsupported)
    while rank 0 captures and replays.  The all-reduces are device ops of the launch list either way, so the eager rank
    and the replaying rank issue the same collectives in the same order: no deadlock, bit-identical weights on both
    ranks, equal to a run in which nobody captures."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_capfail_worker, args=(2, _free_port(), d, 1), nprocs=2, join=True)
        mp.spawn(_capfail_worker, args=(2, _free_port(), d, None), nprocs=2, join=True)
        got = [dict(np.load(os.path.join(d, 'capfail_f1_r%d.npz' % r))) for r in (0, 1)]
        ref = [dict(np.load(os.path.join(d, 'capfail_r%d.npz' % r))) for r in (0, 1)]
    assert got[1]['eager'].all() and not got[0]['eager'].any()       # rank 1 fell back, rank 0 did not
    assert int(got[0]['replays']) >= 2 and int(got[1]['replays']) == 0
    for key in ('d_param', 'g_param'):
        assert np.array_equal(got[0][key], got[1][key]), key
        assert np.array_equal(got[0][key], ref[0][key]), key
        assert np.array_equal(ref[0][key], ref[1][key]), key


def _drift_worker(rank, world, port, outdir, drift):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    import bench
    import train_cases as TC
    from oracle import cbind
    from action_conditioned_gans_amd import _lib, graph as G
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    sess, tr = TC.build_trainer(lambda **kw: G.Session(device='cpu', lib=cbind.load(), world_size=world, rank=rank, **kw),
                                CASE, world_size=world, collectives='stream')
    x, y, a, s = _inputs(rank)
    tr.train_d(x, y, a)
    tr.train_g(x, y, a, s)
    if drift and rank == 1:
        tr.g_opt_op.inputs[0].buf[17] += 1e-3                 # one weight of one rank off by a little
    try:
        verdict = 'in sync' if bench._require_weights_in_sync(sess, tr, world) else 'no check'
    except _lib.AcgError as e:
        verdict = 'raised: ' + str(e)[:60]
    with open(os.path.join(outdir, 'sync_%d_r%d.txt' % (drift, rank)), 'w') as f:
        f.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_bench_refuses_a_result_when_the_ranks_drifted_apart():
    """bench.py, N > 1: before the result line every rank's trainable weights are compared over the control group (checksums
    of the optimizers' flat parameter buffers).  Two ranks after one data-parallel D + G step on DIFFERENT samples hold the
    same bits -> in sync; one weight of one rank nudged -> AcgError on BOTH ranks (no line, non-zero exit)."""
    with tempfile.TemporaryDirectory() as d:
        for drift in (0, 1):
            mp.spawn(_drift_worker, args=(2, _free_port(), d, drift), nprocs=2, join=True)
        out = {(drift, r): open(os.path.join(d, 'sync_%d_r%d.txt' % (drift, r))).read() for drift in (0, 1) for r in (0, 1)}
    assert out[(0, 0)] == out[(0, 1)] == 'in sync', out
    assert out[(1, 0)].startswith('raised: data parallel') and out[(1, 1)].startswith('raised: data parallel'), out
