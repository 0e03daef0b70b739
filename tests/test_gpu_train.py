"""GPU parity of the whole training step (host runtime over libacgan_hip.so) against the fp64-oracle
golden vectors, plus HIP-graph replay equivalence.  Tolerance: 1e-3 rel (north_star) on frames and
losses; gradients are compared per variable by L2 norm at 10x that."""
import os

import numpy as np
import pytest
import torch

import train_cases as TC

from action_conditioned_gans_amd import graph as G

pytestmark = pytest.mark.gpu


def gpu_session(**kw):
    return G.Session(device='cuda:0', **kw)


@pytest.mark.parametrize('name', sorted(TC.MG.CASES))
def test_trainer_matches_golden(name):
    TC.case_golden(gpu_session, name, 1e-3)


@pytest.mark.parametrize('name', ['c2_dna_bce_adam', 'plain_adv_bce_rmsprop', 'c1_plain_l1'])
def test_pretrain_step_matches_golden(name):
    """Trainer.pretrain_g (train.py:114-121) against the oracle's pre-training step: loss, elementwise gradient samples and
    the generator weights after the update."""
    TC.case_pretrain_golden(gpu_session, name, 1e-3)


@pytest.mark.parametrize('name', ['c2_dna_bce_adam', 'c4_dna_wass_rmsprop'])
def test_hip_graph_replay_equals_eager(name):
    """Run 1 is eager, run 2 captures, run 3+ replays: the weights must match an all-eager session bit for bit."""
    x, y, a, s = TC.MG.inputs(2)
    finals = []
    for use_graphs in (False, True):
        sess, tr = TC.build_trainer(gpu_session, name, use_hip_graphs=use_graphs)
        for _ in range(4):
            tr.train_d(x, y, a)
            frames = tr.train_g(x, y, a, s)
        torch.cuda.synchronize()
        finals.append(({n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()}, frames))
        if use_graphs:
            assert all(p.graphs is not None for p in sess._programs.values() if p.runs >= 2)
    (pe, fe), (pg, fg) = finals
    for n in pe:
        assert torch.equal(pe[n], pg[n]), n
    assert np.array_equal(fe, fg)


def test_adam_step_counter_rides_on_the_deferred_reduction():
    """ABI 7: where an optimizer's deferred weight-gradient reduction runs in the program, that launch carries the increment of
    Adam's device step counter (no step_inc launch of its own); the counter must advance exactly once per update whichever launch
    does it (the update's bias correction - weights vs the golden vectors - is pinned by test_step_matches_golden)."""
    from action_conditioned_gans_amd import ops as O
    sess, tr = TC.build_trainer(gpu_session, 'c2_dna_bce_adam', batch=32)
    x, y, a, s = TC.MG.inputs(2)
    xs, ys, as_, ss = np.tile(x, (16, 1, 1, 1)), np.tile(y, (16, 1, 1, 1)), np.tile(a, (16, 1)), np.tile(s, (16, 1))
    n = 5                                   # eager, capture, replays
    for _ in range(n):
        tr.train_d(xs, ys, as_)
    for _ in range(n - 2):
        tr.train_g(xs, ys, as_, ss)
    torch.cuda.synchronize()
    g = G.get_default_graph()
    steps = sorted(int(op.inputs[-1].buf[0]) for op in g.ops if getattr(op, 'is_optimizer_step', False) and op.inputs[-1].buf is not None)
    assert [v for v in steps if v] == [n - 2, n], steps          # (the pre-training optimizer never ran)
    carried = [op.name for op in g.ops if isinstance(op, O.WgradReduceOp) and getattr(op, '_keep', None) and op._keep[0][0].step_inc]
    assert len(carried) >= 2, 'at batch 32 both optimizers have split weight gradients: their reductions should carry the increment (%s)' % carried


def test_batch32_step_is_finite_and_learns():
    """BASELINE config 2 shapes (B=32, DNA k=5, bce, Adam): a few steps run, stay finite, and the pretrain
    loss goes down on a fixed batch."""
    from action_conditioned_gans_amd import optim, train as T
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = gpu_session()
    tr = T.Trainer(sess, True, 'bce', 'adam', True, batch_size=32, img_size=64, ksize=5)
    sess.run(G.global_variables_initializer())
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (32, 64, 64, 3)).astype(np.float32)
    y = np.roll(x, 1, axis=2)
    a = rng.standard_normal((32, 10)).astype(np.float32)
    s = rng.standard_normal((32, 5)).astype(np.float32)
    losses = [tr.pretrain_g(x, y, a, s) for _ in range(8)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    for _ in range(3):
        summ = tr.train_d(x, y, a, summarize=True)
        frames = tr.train_g(x, y, a, s)
    assert all(np.isfinite(v) for v in summ.values()), summ
    assert np.isfinite(frames).all()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('loss,opt', [('bce', 'adam'), ('wass', 'rmsprop')], ids=['config2', 'config4'])
def test_config2_full_size_matches_oracle_live(loss, opt):
    """BASELINE config 2 at its FULL size (batch 32, 64x64x3, DNA k=5, bce, Adam, fp32 - the bench workload) and config
    4's losses / optimizer (Wasserstein + RMSProp + clip; there also the weights after the two updates): the evaluation pass, one D step and one G step against the fp64 oracle run live on the host, 1e-3 - the predicted
    frame and state, the three losses and the per-variable gradient norms of both steps (every conv / BatchNorm /
    DNA kernel at the shapes and launch geometries the benchmark times, paired launches and deferred reductions
    included)."""
    import torch
    from oracle import models as OM
    from oracle.trainer import OracleTrainer
    from action_conditioned_gans_amd import optim, train as T
    B, S, K = 32, 64, 5
    params = OM.init_params(True, batch=B, img=S, ksize=K, seed=9, dtype=torch.float32)
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = gpu_session()
    tr = T.Trainer(sess, True, loss, opt, True, batch_size=B, img_size=S, ksize=K)
    sess.run(G.global_variables_initializer())
    g = G.get_default_graph()
    for n, v in g.variables.items():
        sess.set_value(v, params[n])
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    y = np.clip(np.roll(x, 2, axis=2) + 0.05 * rng.standard_normal(x.shape).astype(np.float32), -1, 1)
    a = rng.standard_normal((B, 10)).astype(np.float32)
    s = rng.standard_normal((B, 5)).astype(np.float32)
    td = lambda t: torch.from_numpy(t).double()     # noqa: E731
    torch.set_num_threads(16)
    ot = OracleTrainer({k: v.double() for k, v in params.items()}, True, loss, opt, True, K)
    frame, state, _ = tr.test(x, y, a)
    oframe, ostate, _ = ot.test(td(x), td(y), td(a))
    assert TC.rel(frame, oframe.numpy()) <= 1e-3 and TC.rel(state, ostate.numpy()) <= 1e-3
    dsumm = tr.train_d(x, y, a, summarize=True)
    od = ot.train_d(td(x), td(y), td(a), return_all=True)
    assert abs(dsumm['discriminator_loss'] - float(od['d_loss'])) <= 1e-3 * max(abs(float(od['d_loss'])), 1.0)
    TC.check_norms(TC.flat_grad_norms(sess, tr.d_opt_op), {'dgrad_norm/' + k: v.norm() for k, v in ot.last_grads.items()},
                   'dgrad_norm/', 1e-3, 'D grad (batch 32)')
    res = sess.run([tr.g_opt_op, tr.g_loss], tr._feed(x, y, a, s))
    og = ot.train_g(td(x), td(y), td(a), td(s), return_all=True)
    assert abs(res[1][0] - float(og['g_loss'])) <= 1e-3 * abs(float(og['g_loss']))
    TC.check_norms(TC.flat_grad_norms(sess, tr.g_opt_op), {'ggrad_norm/' + k: v.norm() for k, v in ot.last_grads.items()},
                   'ggrad_norm/', 1e-3, 'G grad (batch 32)')
    if opt == 'rmsprop':                                   # weights after 1 D (+ clip) + 1 G update
        for n, v in g.variables.items():
            got, want = sess.get_value(v).double(), ot.p[n]
            assert (got - want).abs().max().item() <= 1e-3 * max(want.abs().max().item(), 1e-3) + 2e-6, n


def test_config5_shapes_128x128_k11_match_oracle():
    """BASELINE config 5 geometry (128x128, 11x11 DNA kernel; fp32 here): action tile H/16 = 8, D logits 4x4,
    state head 8x8 VALID (64 taps).  Checked live against the fp64 oracle at batch 2."""
    import torch
    from oracle import models as OM
    from oracle.trainer import OracleTrainer
    from action_conditioned_gans_amd import optim, train as T
    B, S, K = 2, 128, 11
    params = OM.init_params(True, batch=B, img=S, ksize=K, seed=5, dtype=torch.float32)
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = gpu_session()
    tr = T.Trainer(sess, True, 'bce', 'rmsprop', True, batch_size=B, img_size=S, ksize=K)
    sess.run(G.global_variables_initializer())
    g = G.get_default_graph()
    assert set(params) == set(g.variables)
    for n, v in g.variables.items():
        assert tuple(params[n].shape) == v.shape, n
        sess.set_value(v, params[n])
    assert tr.d_out_gen.shape == (B, 4, 4, 1) and g.variables['g/sconv5/weights'].shape == (8, 8, 16, 5)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    y = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    a = rng.standard_normal((B, 10)).astype(np.float32)
    s = rng.standard_normal((B, 5)).astype(np.float32)
    td = lambda t: torch.from_numpy(t).double()
    ot = OracleTrainer({k: v.double() for k, v in params.items()}, True, 'bce', 'rmsprop', True, K)
    frame, state, summ = tr.test(x, y, a)
    oframe, ostate, opsnr = ot.test(td(x), td(y), td(a))
    assert TC.rel(frame, oframe.numpy()) <= 1e-3 and TC.rel(state, ostate.numpy()) <= 1e-3
    dsumm = tr.train_d(x, y, a, summarize=True)
    od = ot.train_d(td(x), td(y), td(a), return_all=True)
    assert abs(dsumm['discriminator_loss'] - float(od['d_loss'])) <= 1e-3
    res = sess.run([tr.g_opt_op, tr.g_loss], tr._feed(x, y, a, s))
    og = ot.train_g(td(x), td(y), td(a), td(s), return_all=True)
    assert abs(res[1][0] - float(og['g_loss'])) <= 1e-3 * abs(float(og['g_loss']))
    for n, v in g.variables.items():                       # RMSProp: weights after 1 D + 1 G step
        got, want = sess.get_value(v).double(), ot.p[n]
        assert (got - want).abs().max().item() <= 1e-3 * max(want.abs().max().item(), 1e-3) + 2e-6, n


@pytest.mark.parametrize('name', ['c2_dna_bce_adam', 'plain_adv_bce_rmsprop'])
def test_rollout_matches_oracle(name):
    """the recursive multi-step rollout (Trainer.test_sequence, train.py:157-176 and the eval
    block at :285-298) - prediction and predicted state fed back for T-1 steps - against the fp64 oracle's rollout.
    Errors compound through the recursion, so the bar is the north_star 1e-3 on the LAST frame as well."""
    from oracle.trainer import OracleTrainer
    from oracle import models as OM
    adv, loss, opt, dna, batch, ksize = TC.MG.CASES[name]
    sess, tr = TC.build_trainer(gpu_session, name)
    rng = np.random.default_rng(5)
    T_ = 5
    frames = rng.uniform(-1, 1, (batch, T_, 64, 64, 3)).astype(np.float32)
    acts = rng.standard_normal((batch, T_, 10)).astype(np.float32)
    pred, summ = tr.test_sequence(frames, frames, acts)
    params = OM.init_params(dna, batch=batch, ksize=ksize, seed=TC.MG.PARAM_SEED, dtype=torch.float32)
    ot = OracleTrainer({k: v.double() for k, v in params.items()}, adv, loss, opt, dna, ksize)
    want, psnrs = ot.test_sequence(torch.from_numpy(frames).double(), torch.from_numpy(frames).double(),
                                   torch.from_numpy(acts).double())
    assert pred.shape == (batch, T_ - 1, 64, 64, 3)
    for j in range(T_ - 1):
        assert TC.rel(pred[:, j], want[:, j].numpy()) <= 1e-3, j
    assert abs(summ['g_psnr'] - psnrs[0]) <= 1e-3 * abs(psnrs[0])
    # the default keeps the prediction and the predicted state on the device between steps (round 5); the reference's own
    # loop - every frame through the host - gives the same bits
    pred_host, summ_host = tr.test_sequence(frames, frames, acts, device_loop=False)
    assert np.array_equal(pred, pred_host) and summ == summ_host
    one, _ = tr.test_sequence(frames, frames, acts, steps=1)
    assert np.array_equal(one, pred[:, :1])


def test_test_sequence_literal_matches_reference_indexing():
    """Trainer.test_sequence(literal=True) is the reference's method as written (train.py:157-176): six steps, step j
    commanded by actions[:, 2 j] and fed next_frame[:, 2 j], second result current_frame[1:7] - against the oracle's
    restatement of the same lines."""
    from oracle.trainer import OracleTrainer
    from oracle import models as OM
    name = 'c2_dna_bce_adam'
    adv, loss, opt, dna, batch, ksize = TC.MG.CASES[name]
    sess, tr = TC.build_trainer(gpu_session, name, batch=8)
    rng = np.random.default_rng(6)
    frames = rng.uniform(-1, 1, (8, 11, 64, 64, 3)).astype(np.float32)
    acts = rng.standard_normal((8, 11, 10)).astype(np.float32)
    pred, tail = tr.test_sequence(frames, frames, acts, literal=True)
    params = OM.init_params(dna, batch=8, ksize=ksize, seed=TC.MG.PARAM_SEED, dtype=torch.float32)
    ot = OracleTrainer({k: v.double() for k, v in params.items()}, adv, loss, opt, dna, ksize)
    want, wtail = ot.test_sequence(torch.from_numpy(frames).double(), torch.from_numpy(frames).double(), torch.from_numpy(acts).double(),
                                   steps='literal')
    assert pred.shape == (8, 6, 64, 64, 3) and tail.shape == (6, 64, 64, 3)
    for j in range(6):
        assert TC.rel(pred[:, j], want[:, j].numpy()) <= 1e-3, j
    assert TC.rel(tail, wtail.numpy()) <= 1e-3


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_checkpoint_roundtrip_on_the_gpu(tmp_path, dtype):
    """Saver (train.py:215,274; test.py:29-30) on CUDA sessions: weights and optimizer slots (Adam m, v, step counter)
    are saved, restored into a FRESH session, and the resumed run continues bit-identically - also through the
    captured HIP graphs and, in a bf16 session, the refreshed bf16 filter copies."""
    from action_conditioned_gans_amd.saver import Saver, latest_checkpoint
    x, y, a, s = TC.MG.inputs(2)
    sess, tr = TC.build_trainer(gpu_session, 'c2_dna_bce_adam', dtype=dtype)
    for _ in range(3):
        tr.train_d(x, y, a)
        tr.train_g(x, y, a, s)
    Saver().save(sess, str(tmp_path / 'model100'))
    assert latest_checkpoint(str(tmp_path)) == str(tmp_path / 'model100')
    for _ in range(3):
        tr.train_d(x, y, a)
        frames = tr.train_g(x, y, a, s)
    want = {n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()}
    sess2, tr2 = TC.build_trainer(gpu_session, 'c2_dna_bce_adam', dtype=dtype)
    Saver().restore(sess2, str(tmp_path / 'model100'))
    for _ in range(3):
        tr2.train_d(x, y, a)
        frames2 = tr2.train_g(x, y, a, s)
    for n, v in G.get_default_graph().variables.items():
        assert torch.equal(sess2.get_value(v), want[n]), n
    assert np.array_equal(frames, frames2)


def test_device_resident_feeds_equal_host_feeds():
    """sess.run fed with CUDA tensors (one fused copy launch, incl. the channel-padded image placeholder) must give
    exactly what the same values give when fed as numpy arrays."""
    x, y, a, s = TC.MG.inputs(2)
    sess, tr = TC.build_trainer(gpu_session, 'c2_dna_bce_adam')
    host = sess.run([tr.g_next_frame, tr.g_loss], tr._feed(x, y, a, s))
    dev = lambda t: torch.from_numpy(t).cuda()
    devr = sess.run([tr.g_next_frame, tr.g_loss], tr._feed(dev(x), dev(y), dev(a), dev(s)))
    assert np.array_equal(host[0], devr[0]) and np.array_equal(host[1], devr[1])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_training_loop_runs_wass_rmsprop_n_critic(dtype):
    """train() end to end on synthetic sequences: pretrain iterations, then n_critic=5 D steps per G step
    (train.py:217-263) with weight clip; weights stay finite and inside the clip range.  Both pipelines (`--dtype`)."""
    from action_conditioned_gans_amd import train as T
    tr = T.train('synthetic', None, None, None, None, True, 'wass', 'rmsprop', True, batch_size=4, seq_len=8,
                 train_iter=6, pretrain_iter=2, device='cuda:0', quiet=True, dtype=dtype)
    for v in tr.d_vars:
        val = tr.sess.get_value(v)
        assert torch.isfinite(val).all() and val.abs().max().item() <= 0.01 + 1e-7, v.name
    for v in tr.g_vars:
        assert torch.isfinite(tr.sess.get_value(v)).all(), v.name


# Declared tolerances of the bf16 pipeline (BASELINE configs 3 and 5) against the fp64 oracle that rounds the SAME
# tensors to bfloat16 (oracle.tf_ops.bf16_storage: conv operands and outputs, BatchNorm + activation outputs, the
# concatenated maps and the gradients of all of them; float32 master weights, weight gradients, statistics and
# losses).  What is left between the two is summation order (fp32 vs fp64 accumulation) and the one-ulp bf16 rounding
# flips it causes (2^-8 relative each), compounding through ~10 layers:
# Round 3 (profiles/r3/bf16_vs_oracle_*.txt, the per-variable table this test prints): every weight gradient's norm is
# within 0.5 % (betas, 100x smaller vectors, within 2.5 %) and every cosine >= 0.992 at batch 32 / >= 0.983 at batch 2,
# after the head layer d/conv6 stopped rounding its conv output to bf16 (round 2: 3.6 % / 0.984).  The cosine floor is
# the chaos of storage rounding amplified by the BatchNorm'd head (test_epilogue_statistics_match_the_statistics_pass).
BF16_TOL = {'frame': 1.0e-2,      # max abs error of the predicted frame / frame scale        (measured 0.71e-2 / 0.37e-2)
            'state': 1.5e-2,      # predicted state (a 5-vector behind three strided convs)    (0.83e-2 / 1.0e-2)
            'loss': 1e-3,         # D and G loss values, relative (5e-3 until round 5)         (<= 4e-5)
            'grad_norm': 3e-2,    # per-variable gradient L2 norm, relative                    (2.5e-2 / 1.8e-2, betas)
            'grad_cos': 0.98}     # per-variable gradient direction: cosine with the oracle's  (0.9923 / 0.9837)


def _bf16_step_vs_oracle(B, S, K, loss, opt, seed):
    from oracle import models as OM, tf_ops as OT
    from oracle.trainer import OracleTrainer
    from action_conditioned_gans_amd import optim, train as T
    params = OM.init_params(True, batch=2, img=S, ksize=K, seed=seed, dtype=torch.float32)
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = gpu_session(dtype='bf16')
    tr = T.Trainer(sess, True, loss, opt, True, batch_size=B, img_size=S, ksize=K)
    sess.run(G.global_variables_initializer())
    g = G.get_default_graph()
    assert g.act_dtype == torch.bfloat16 and tr.d_out_gen.dtype == torch.float32 and tr.g_out.dtype == torch.float32
    for n, v in g.variables.items():
        sess.set_value(v, params[n])
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    y = np.clip(np.roll(x, 2, axis=2) + 0.05 * rng.standard_normal(x.shape).astype(np.float32), -1, 1)
    a = rng.standard_normal((B, 10)).astype(np.float32)
    s = rng.standard_normal((B, 5)).astype(np.float32)
    td = lambda t: torch.from_numpy(t).double()     # noqa: E731
    torch.set_num_threads(16)
    ot = OracleTrainer({k: v.double() for k, v in params.items()}, True, loss, opt, True, K)
    report = {}

    def grads_of(step_op):
        offs, _, _ = g.layout(step_op.scope)
        flat = step_op.inputs[1].buf.detach().double().cpu()
        return {n: flat[o:o + g.variables[n].numel].reshape(g.variables[n].shape) for n, o in offs.items()}

    lines = []

    def check_grads(got, want, what):
        scale = max(float(v.norm()) for v in want.values())
        worst_n, worst_c = 0.0, 1.0
        failures = []
        for n, w in want.items():
            wn, gn = float(w.norm()), float(got[n].norm())
            if wn < 1e-6 * scale:
                assert gn <= 1e-3 * scale, (what, n, gn)
                continue
            cos = float((w * got[n]).sum() / (wn * gn))
            lines.append('%-8s %-36s |g| %.4e  norm err %+.4f  cos %.5f' % (what, n, wn, (gn - wn) / wn, cos))
            worst_n, worst_c = max(worst_n, abs(gn - wn) / wn), min(worst_c, cos)
            if abs(gn - wn) > BF16_TOL['grad_norm'] * wn + 1e-5 * scale or cos < BF16_TOL['grad_cos']:
                failures.append((what, n, gn, wn, cos))
        report[what] = (worst_n, worst_c)
        return failures

    with OT.bf16_storage():
        frame, state, _ = tr.test(x, y, a)
        oframe, ostate, _ = ot.test(td(x), td(y), td(a))
        report['frame'], report['state'] = TC.rel(frame, oframe.numpy()), TC.rel(state, ostate.numpy())
        assert report['frame'] <= BF16_TOL['frame'] and report['state'] <= BF16_TOL['state'], report
        dsumm = tr.train_d(x, y, a, summarize=True)
        got_d = grads_of(tr.d_opt_op)
        od = ot.train_d(td(x), td(y), td(a), return_all=True)
        report['d_loss'] = abs(dsumm['discriminator_loss'] - float(od['d_loss'])) / max(abs(float(od['d_loss'])), 1.0)
        assert report['d_loss'] <= BF16_TOL['loss'], report
        bad = check_grads(got_d, ot.last_grads, 'D grad')
        res = sess.run([tr.g_opt_op, tr.g_loss], tr._feed(x, y, a, s))
        got_g = grads_of(tr.g_opt_op)
        og = ot.train_g(td(x), td(y), td(a), td(s), return_all=True)
        report['g_loss'] = abs(res[1][0] - float(og['g_loss'])) / abs(float(og['g_loss']))
        assert report['g_loss'] <= BF16_TOL['loss'], report
        bad += check_grads(got_g, ot.last_grads, 'G grad')
    # the per-variable table: printed always, kept when ACG_BF16_REPORT names a directory
    table = '\n'.join(lines)
    print(table)
    if os.environ.get('ACG_BF16_REPORT'):
        with open(os.path.join(os.environ['ACG_BF16_REPORT'], 'bf16_vs_oracle_b%d_s%d_k%d.txt' % (B, S, K)), 'w') as f:
            f.write('# bf16 pipeline vs fp64 oracle with the same tensors rounded to bf16: per-variable gradient norm error and cosine\n')
            f.write(table + '\n' + repr(report) + '\n')
    assert not bad, bad
    print('bf16 vs oracle (B=%d, %dx%d, k=%d):' % (B, S, S, K), {k: (tuple(round(float(t), 5) for t in v) if isinstance(v, tuple) else round(float(v), 5)) for k, v in report.items()})


@pytest.mark.timeout(900)
def test_bf16_config3_step_matches_bf16_oracle():
    """BASELINE config 3 at its per-GPU size (batch 32, 64x64x3, DNA k=5, bce, Adam) in the bf16 pipeline - bf16
    activations in HBM, bf16 matrix cores, fp32 accumulation and master weights - against the fp64 oracle with the same
    tensors rounded to bf16, run live on the host: evaluation frame and state, D and G losses, and every gradient of
    one D step and one G step by norm and direction (BF16_TOL)."""
    _bf16_step_vs_oracle(32, 64, 5, 'bce', 'adam', seed=9)


@pytest.mark.timeout(900)
def test_bf16_config5_geometry_matches_bf16_oracle():
    """BASELINE config 5 geometry in its stated arithmetic: 128x128, 11x11 DNA kernel (121 logits at a pitch of 128),
    bf16, at batch 2 (the oracle's fp64 128x128 step is what bounds the size)."""
    _bf16_step_vs_oracle(2, 128, 11, 'bce', 'rmsprop', seed=5)


def test_bf16_replay_and_weight_copies():
    """bf16 session mechanics: the bf16 filter copies follow the master weights (set_value, optimizer step inside the
    captured program), and HIP-graph replay equals eager execution bit for bit."""
    x, y, a, s = TC.MG.inputs(2)
    finals = []
    for use_graphs in (False, True):
        sess, tr = TC.build_trainer(gpu_session, 'c2_dna_bce_adam', use_hip_graphs=use_graphs, dtype='bf16')
        for _ in range(4):
            tr.train_d(x, y, a)
            frames = tr.train_g(x, y, a, s)
        torch.cuda.synchronize()
        g = G.get_default_graph()
        for scope, entries in g.weight_copies.items():
            for w, rm, tr_ in entries:
                kh, kw, ca, cb = w.shape
                want = w.buf.detach().to(torch.bfloat16).reshape(kh * kw, ca, cb)
                assert torch.equal(rm.buf[:, :, :cb], want) and torch.equal(tr_.buf[:, :, :ca], want.permute(0, 2, 1)), w.name
        finals.append(({n: sess.get_value(v) for n, v in g.variables.items()}, frames))
    (pe, fe), (pg, fg) = finals
    for n in pe:
        assert torch.equal(pe[n], pg[n]), n
    assert np.array_equal(fe, fg)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_splitk_handoff_to_batchnorm(dtype):
    """Session(slab_handoff=...): small layers are split over K and, instead of a reduction launch, the layer's BatchNorm
    sums the slabs as it loads its input (forward: acg_bn_act_fwd_slabs, which also writes x for the backward pass;
    backward: acg_bn_act_bwd_slabs on the input-gradient slabs).  'quads': only where the resident BatchNorm kernels read
    the quad slab layout (none at this batch since round 4: the grid kernels take these layers and ask for rows); True (the
    default): every split layer, in the layout acg_bn_slabs_layout asks for.  Same summation order and
    rounding of the sums; the BatchNorm arithmetic behind them is a different kernel instantiation (FMA contraction may
    differ), so weights after four D + G steps must agree with the separate-reduction run to rounding level (1e-5 of
    the weight scale), and the hand-off must have been taken."""
    from action_conditioned_gans_amd import ops as O
    x, y, a, s = TC.MG.inputs(2)
    finals = []
    for handoff in (False, 'quads', True):
        sess, tr = TC.build_trainer(gpu_session, 'dna_k6_bce_rmsprop', batch=8, dtype=dtype, slab_handoff=handoff)
        xs, ys = np.tile(x, (4, 1, 1, 1)), np.tile(y, (4, 1, 1, 1))
        as_, ss = np.tile(a, (4, 1)), np.tile(s, (4, 1))
        one_step = None
        for it in range(4):
            tr.train_d(xs, ys, as_)
            frames = tr.train_g(xs, ys, as_, ss)
            if it == 0:         # the state after ONE D + G step: where a kernel bug shows before the trajectory's chaos does
                one_step = ({n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()}, np.array(frames, copy=True))
        torch.cuda.synchronize()
        g = G.get_default_graph()
        fwd = [o._slab[2] for o in g.ops if isinstance(o, O.Conv2dOp) and o._slab is not None]
        bwd = [o._slab[2] for o in g.ops if isinstance(o, O.ConvDgradOp) and o._slab is not None]
        assert (len(fwd) >= 4 and len(bwd) >= 3) if handoff is True else (handoff == 'quads' or (not fwd and not bwd)), (handoff, fwd, bwd)
        assert handoff != 'quads' or set(fwd + bwd) <= {1}, (fwd, bwd)
        finals.append(({n: sess.get_value(v) for n, v in g.variables.items()}, frames, one_step))
    p0, f0 = finals[0][:2]
    nrel = lambda got, want: float(np.linalg.norm(np.asarray(got, np.float64) - np.asarray(want, np.float64)) / np.linalg.norm(np.asarray(want, np.float64)))  # noqa: E731
    if dtype == 'f32':
        for p1, f1, _ in finals[1:]:
            for n in p0:
                d = float((p0[n].double() - p1[n].double()).abs().max())
                assert d <= 1e-5 * max(float(p0[n].abs().max()), 1e-3), (n, d)
            assert TC.rel(f1, f0) <= 1e-4
        return
    # bf16, the bug-detecting bar: after ONE D + G step the hand-off run and the separate-reduction run
    # differ by storage rounding only - frames within 5e-3 (measured 2.5e-3), every filter within 2e-2 of its scale (measured: up
    # to 1.2e-2, g/tconv1, whose xavier bound is 0.025)
    w1, fr1 = finals[0][2]
    for _, _, (w1h, fr1h) in finals[1:]:
        assert nrel(fr1h, fr1) <= 5e-3, ('frames after one step', nrel(fr1h, fr1))
        for n in w1:
            if not n.endswith('weights'):
                continue        # a beta / bias after ONE step IS its first RMSProp update (+-lr-sized, sign-like): no scale to compare against
            d = float((w1h[n].double() - w1[n].double()).abs().max())
            assert d <= 2e-2 * max(float(w1[n].abs().max()), 1e-3), ('after one step', n, d)
    # bf16 over FOUR steps: RMSProp's first steps are lr * g / sqrt(0.1 g^2), sign-like, so an element whose cancellation-heavy gradient sits near 0
    # moves a whole step either way once one bf16 ulp flips upstream: four steps amplify ANY rounding-level change of the arithmetic
    # to percents (measured on this case: frames 2.5e-3 apart after one step, 6.5e-2 after four - and the float32 run of the same
    # steps is just as far from both, 7.7e-2 / 8.1e-2).  The yardstick is therefore that float32 run: the hand-off may move the bf16
    # trajectory no further than bf16 arithmetic itself moved it (x 1.5); bit-level agreement of the hand-off kernels with the
    # separate reduction is pinned at op level (test_gpu_ops.py) and to 1e-5 by the float32 variant of this test.
    sess, tr = TC.build_trainer(gpu_session, 'dna_k6_bce_rmsprop', batch=8, dtype='f32', slab_handoff=False)
    for _ in range(4):
        tr.train_d(xs, ys, as_)
        frames32 = tr.train_g(xs, ys, as_, ss)
    p32 = {n: sess.get_value(v).cpu().numpy() for n, v in G.get_default_graph().variables.items()}
    yard_f = nrel(f0, frames32)
    assert 1e-2 < yard_f < 0.2, yard_f        # the premise: bf16 vs float32 after four steps is percents, not rounding
    for p1, f1, _ in finals[1:]:
        assert nrel(f1, f0) <= 1.5 * yard_f, (nrel(f1, f0), yard_f)
        for n in p0:
            assert bool(torch.isfinite(p1[n]).all()), n
            if n.endswith('weights'):
                d, yard = nrel(p1[n].cpu().numpy(), p0[n].cpu().numpy()), nrel(p0[n].cpu().numpy(), p32[n])
                assert d <= 1.5 * max(yard, 1e-3), (n, d, yard)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_epilogue_statistics_match_the_statistics_pass(dtype):
    """Session(epilogue_stats=True), the default: BatchNorm statistics come out of the producing conv's epilogue
    (acg_(de)conv2d_fwd_stats -> acg_bn_act_fwd_partials) instead of a pass over the activation.  Same values summed in a
    different order (per row tile, then float64), so the gradients and frames of the first D + G step must agree with the
    statistics-pass run to rounding level, and the fused path must actually have been taken."""
    from action_conditioned_gans_amd import ops as O
    # eight DISTINCT samples (round 2 tiled two samples four times: BatchNorm over 8 distinct values per channel in the
    # 2x2 layers - an ill-conditioned normalisation that amplified one bf16 ulp into 4-11 % of the discriminator gradient)
    xs, ys, as_, ss = TC.MG.inputs(8)
    finals = []
    for fused in (False, True):
        sess, tr = TC.build_trainer(gpu_session, 'c2_dna_bce_adam', batch=8, dtype=dtype, epilogue_stats=fused)
        tr.train_d(xs, ys, as_)
        frames = tr.train_g(xs, ys, as_, ss)
        torch.cuda.synchronize()
        g = G.get_default_graph()
        n = sum(1 for o in g.ops if isinstance(o, O.Conv2dOp) and o._stats is not None)
        assert (n >= 3) if fused else (n == 0), (fused, n)      # batch 8: most small layers are split over K and keep the pass
        grads = [op.inputs[1].buf.detach().double().cpu().clone() for op in (tr.d_opt_op, tr.g_opt_op)]
        finals.append((grads, frames))
    (g0, f0), (g1, f1) = finals
    # the gradients of the first D and G step (the weights behind them are an Adam step apart whatever the gradient's size, so
    # they are the wrong thing to compare), whole-buffer relative error.
    # float32: the discriminator's loss is smooth -> accumulation-order level.  The generator's loss is not: |G - y| and the
    # GDL (ops.py:100-120) have kinks, and ONE element of the 8 x 64 x 64 x 3 frame gradient changing sign moves the whole
    # gradient by 2 / sqrt(1.5e6) = 1.6e-3 - which a rounding-level change of the frame does to the odd element that sits
    # within 1e-7 of a kink (measured with eight distinct samples: D 3.7e-6, G 1.7e-3 = one flip); allow three.
    # bf16: storage rounding is chaotic - a last-bit change of one statistic re-rounds 3e-5 of that layer's elements by a
    # whole bf16 ulp, the next layer's inputs then differ by 2e-5 everywhere, which re-rounds 0.5 % of ITS outputs, and after
    # three or four layers two runs differ by the full bf16 rounding noise (~3e-3 per element) everywhere.  The discriminator
    # then amplifies it: its last layer is BatchNorm'd without activation (models.py:87-88), dy of the bce loss is nearly
    # linear in the normalised logit, and BatchNorm backward removes exactly the constant and linear parts - what is left is
    # ~1/30 of dy, so a 2e-3 relative perturbation of the normalised logits is 5-10 % of the gradient.  That is the floor of
    # ANY two bf16 runs of this model (the bf16-vs-oracle tests above sit at cosine 0.9967 = 8 %); measured here with eight
    # distinct samples: D 6.3e-2, G 4.6e-2 (G: ~200 of the frame-loss kinks flip under a 1e-4 change of the frame).
    # Round 4: the bf16 discriminator bar is no longer a number picked to pass.  The oracle's bf16-storage emulation is run
    # twice on THESE inputs and weights, the second time with d/conv1's conv output perturbed by 1e-7 (what another summation
    # order does): that difference - 4.2e-2 here - is the model's own sensitivity, independent of any kernel
    # (tests/bf16_head_sensitivity.py; profiles/r4/n_bf16_head_sensitivity.txt shows that it does not move when the head's
    # neighbourhood is kept in float32, only when whole layers of D leave bf16).  Two HIP runs differ at EVERY BatchNorm of G and
    # D, not at one: they may differ by at most twice that.
    tols = {'f32': (2e-5, 5e-3), 'bf16': (0.1, 0.1)}[dtype]
    if dtype == 'bf16':
        import bf16_head_sensitivity as HS
        from oracle import models as OM
        floor = HS.run_to_run_floor(OM.init_params(True, batch=8, ksize=5, seed=TC.MG.PARAM_SEED, dtype=torch.float32), xs, ys, as_)
        print('oracle run-to-run floor of the bf16 D gradient on these inputs: %.3g' % floor)
        assert 1e-2 < floor < 6e-2, floor
        tols = (2.0 * floor, 0.1)
    for a0, a1, who, tol in zip(g0, g1, ('d', 'g'), tols):
        err = float((a0 - a1).norm() / a0.norm())
        print('epilogue statistics vs pass, %s, %s gradient: relative difference %.3g' % (dtype, who, err))
        assert err <= tol, (who, err)
    assert TC.rel(f1, f0) <= (1e-4 if dtype == 'f32' else 3e-2)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_side_branch_is_bit_identical(dtype):
    """The DNA generator's state head runs as a side chain (Graph.side_branch): on the session's second HIP stream, a
    parallel branch of the step's HIP graph, hoisted to where its inputs exist.  Same kernels on the same data, only
    the schedule differs: weights and frames after three D + G steps (eager, capture, replay) must equal those of a
    session that keeps everything on one stream, bit for bit - a missing fork or join edge shows up here.  (Round 5: a session
    with side chains never takes the grid-exchange BatchNorm kernels - two of them must not overlap - so the one-stream
    reference session is told the same, bn_grid_exchange=False; against the default kernels the results agree to rounding.)"""
    x, y, a, s = TC.MG.inputs(2)
    xs, ys = np.tile(x, (4, 1, 1, 1)), np.tile(y, (4, 1, 1, 1))
    as_, ss = np.tile(a, (4, 1)), np.tile(s, (4, 1))
    finals = []
    for side in (False, True):
        sess, tr = TC.build_trainer(gpu_session, 'dna_k6_bce_rmsprop', batch=8, dtype=dtype, side_branches=side, bn_grid_exchange=False)
        assert sess.rt.bn_flags == 1
        for _ in range(3):
            tr.train_d(xs, ys, as_)
            frames = tr.train_g(xs, ys, as_, ss)
        torch.cuda.synchronize()
        sess.rt.check_exchange_flags()
        g = G.get_default_graph()
        n_side = sum(1 for o in g.ops if o.side_stream)
        assert n_side >= 10, n_side          # sconv3-5 with BatchNorm / bias, the state loss, and their gradient ops
        finals.append(({n: sess.get_value(v) for n, v in g.variables.items()}, frames))
    (p0, f0), (p1, f1) = finals
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n
    assert np.array_equal(f0, f1)


def test_data_parallel_machinery_on_one_rank():
    """The multi-GPU path cannot be launched from here, so drive everything but the peers on ONE rank: this process's
    own RCCL communicator of size 1 (comm.py), the per-bucket ncclAllReduce captured into the step's HIP graphs - on the
    compute stream and on the side stream (fork / join edges) - the 1/world scale in the optimizer, and the
    exact-global-batch mode (SyncBN, global state-loss norm).  Weights must equal the plain single-GPU run bit for bit.
    Runs in a child process (tests/dp_one_rank.py) that ends with ncclCommDestroy and a normal exit: any abort fails."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'dp_one_rank.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DP_ONE_RANK_OK' in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[:3000], r.stderr[-2000:])


def _push_like_stream(batch, steps, seed=3):
    """A learnable synthetic stream (the bench's white noise has nothing to learn): smooth random images (8x8 noise
    upsampled to 64x64) that move horizontally by -2..2 pixels, the shift encoded in the first action component; the state
    is a fixed linear function of the action.  -> list of (x, y, action||state, next_state)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(steps):
        low = rng.uniform(-1, 1, (batch, 8, 8, 3)).astype(np.float32)
        x = np.clip(np.kron(low, np.ones((1, 8, 8, 1), np.float32)) + 0.05 * rng.standard_normal((batch, 64, 64, 3)).astype(np.float32), -1, 1)
        shift = rng.integers(-2, 3, batch)
        y = np.stack([np.roll(x[i], int(shift[i]), axis=1) for i in range(batch)])
        a = rng.standard_normal((batch, 10)).astype(np.float32) * 0.1
        a[:, 0] = shift / 2.0
        s = (a[:, :5] * 0.5 + 0.1).astype(np.float32)
        out.append((x, y, a, s))
    return out


@pytest.mark.timeout(900)
def test_bf16_training_tracks_fp32_over_200_steps():
    """the bf16 pipeline (BASELINE configs 3 / 5) against the float32 one over a whole trajectory, not
    one step.  200 generator pre-training iterations (train.py:114-121, the deterministic part of the reference loop:
    0.05 L1 / B + state loss, Adam) on the same learnable stream from the same initial weights, then 40 adversarial
    D + G iterations.  Declared band: the pre-training loss averaged over the last 50 iterations within 8 % of the
    float32 run, every 20-iteration window within 12 %, and the PSNR on held-out batches (three late checkpoints) within
    0.75 dB, both runs having actually learned (loss down by > 25 %, PSNR up by > 1 dB); every weight finite after the
    adversarial phase and D's weights inside the clip.  (The per-iteration loss follows the batch - 22 to 66 on this stream -
    and single iterations of the two runs differ by up to 12 %: Adam at lr 1e-3 amplifies a last-bit difference, so ANY
    reordering of a BatchNorm's partial sums moves the bf16 trajectory by that much.  A 5 % band on 20 iterations sat inside
    one standard error of that mean and failed on a change that only permuted the order of stride classes.)"""
    from action_conditioned_gans_amd import optim, train as T
    B, n_pre, n_adv = 16, 200, 40
    stream = _push_like_stream(B, n_pre + n_adv)
    held = _push_like_stream(B, 4, seed=11)
    curves = {}
    for dtype in ('f32', 'bf16'):
        sess, tr = TC.build_trainer(gpu_session, 'c2_dna_bce_adam', batch=B, dtype=dtype)

        def psnr():
            vals = []
            for x, y, a, s in held:
                frame = tr.test(x, y, a)[0]
                vals.append(10.0 * np.log10(1.0 / np.mean((frame - y) ** 2)))
            return float(np.mean(vals))
        p0 = psnr()
        # (Trainer.pretrain_g returns g_loss, which carries the adversarial and GDL terms; what it MINIMISES is g_l2_loss)
        losses, late = [], []
        for i in range(n_pre):
            losses.append(float(np.asarray(sess.run([tr.g_pretrain_opt_op, tr.g_l2_loss], tr._feed(*stream[i]))[1]).reshape(-1)[0]))
            if i + 1 in (n_pre - 40, n_pre - 20, n_pre):      # Adam at lr 1e-3 oscillates from step to step: three late checkpoints
                late.append(psnr())
        p1 = float(np.mean(late))
        for i in range(n_pre, n_pre + n_adv):
            x, y, a, s = stream[i]
            tr.train_d(x, y, a)
            tr.train_g(x, y, a, s)
        torch.cuda.synchronize()
        for v in tr.g_vars + tr.d_vars:
            assert torch.isfinite(sess.get_value(v)).all(), (dtype, v.name)
        for v in tr.d_vars:
            assert sess.get_value(v).abs().max().item() <= 0.01 + 1e-7, (dtype, v.name)
        curves[dtype] = (np.array(losses), p0, p1, psnr())
        print('%s: pre-training loss %.4f -> %.4f (mean of first / last 20), held-out PSNR %.2f -> %.2f dB, %.2f dB after %d adversarial iterations'
              % (dtype, np.mean(losses[:20]), np.mean(losses[-20:]), p0, p1, curves[dtype][3], n_adv))
    (l32, a0, a1, _), (l16, b0, b1, _) = curves['f32'], curves['bf16']
    for l, q0, q1, who in ((l32, a0, a1, 'f32'), (l16, b0, b1, 'bf16')):
        assert np.mean(l[-20:]) < 0.75 * np.mean(l[:20]), (who, 'did not learn', np.mean(l[:20]), np.mean(l[-20:]))
        assert q1 > q0 + 1.0, (who, 'PSNR did not improve', q0, q1)
    assert abs(np.mean(l16[-50:]) - np.mean(l32[-50:])) <= 0.08 * np.mean(l32[-50:]), (np.mean(l16[-50:]), np.mean(l32[-50:]))
    assert abs(b1 - a1) <= 0.75, (b1, a1)
    # the two trajectories stay together all along, not just at the end: windowed means within 12 %
    for lo in range(0, n_pre, 20):
        m32, m16 = np.mean(l32[lo:lo + 20]), np.mean(l16[lo:lo + 20])
        assert abs(m16 - m32) <= 0.12 * m32, (lo, m16, m32)


@pytest.mark.timeout(900)
def test_config5_full_per_gpu_size_replay_equals_eager():
    """BASELINE config 5 at its full per-GPU size - batch 32, 128x128x3, 11x11 DNA kernel, bf16 - which the oracle cannot
    reach (its fp64 step is checked at batch 2): the size-independent properties instead.  Every conv takes the plan it
    has in the bench (128x128 and 128x32 bf16 tiles, unsplit layers with epilogue statistics, the BatchNorm finalize
    launch behind more than 512 partial blocks); three D + G steps launched eagerly and three replayed from captured
    HIP graphs must leave bit-identical weights and frames, all finite, D inside its clip."""
    from action_conditioned_gans_amd import optim, train as T
    rng = np.random.default_rng(31)
    B, S = 32, 128
    x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    y = np.clip(np.roll(x, 3, axis=2) + 0.05 * rng.standard_normal(x.shape).astype(np.float32), -1, 1)
    a = rng.standard_normal((B, 10)).astype(np.float32)
    s = rng.standard_normal((B, 5)).astype(np.float32)
    finals = []
    for use_graphs in (False, True):
        G.reset_default_graph()
        optim.set_data_parallel(1)
        sess = gpu_session(dtype='bf16', use_hip_graphs=use_graphs)
        tr = T.Trainer(sess, True, 'bce', 'adam', True, batch_size=B, img_size=S, ksize=11)
        sess.run(G.global_variables_initializer())
        for _ in range(3):
            tr.train_d(x, y, a)
            frames = tr.train_g(x, y, a, s)
        torch.cuda.synchronize()
        g = G.get_default_graph()
        progs = list(sess._programs.values())
        assert all((p.graphs is not None) == use_graphs for p in progs if p is not None and p.runs >= 2)
        finals.append(({n: sess.get_value(v) for n, v in g.variables.items()}, frames))
        del sess, tr
        torch.cuda.empty_cache()
    (pe, fe), (pg, fg) = finals
    assert np.isfinite(fe).all() and np.array_equal(fe, fg)
    for n in pe:
        assert torch.isfinite(pe[n]).all(), n
        assert torch.equal(pe[n], pg[n]), n
        if n.startswith('d/'):
            assert pe[n].abs().max().item() <= 0.01 + 1e-7, n


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_plain_generator_bias_tanh_in_the_deconv_epilogue(dtype):
    """models.py:20-21: the plain generator's frame is tanh(conv2d_transpose + b).  At batch 16 the last deconv runs unsplit, so its
    bias and tanh move into its epilogue (Conv2dOp._fused_bias; one launch and one pass over the frame less): frames and the
    weights after one pre-training step must match the run that keeps the separate bias_act launch."""
    from action_conditioned_gans_amd import ops as O, optim, train as T
    rng = np.random.default_rng(5)
    B = 16
    x, y = rng.uniform(-1, 1, (B, 64, 64, 3)).astype(np.float32), rng.uniform(-1, 1, (B, 64, 64, 3)).astype(np.float32)
    a, s = rng.standard_normal((B, 10)).astype(np.float32), rng.standard_normal((B, 5)).astype(np.float32)
    outs = []
    for fuse in (False, True):
        G.reset_default_graph()
        optim.set_data_parallel(1)
        sess = gpu_session(dtype=dtype, epilogue_bias=fuse)
        tr = T.Trainer(sess, False, 'bce', 'adam', False, batch_size=B)
        sess.run(G.global_variables_initializer())
        frame = tr.test(x, y, a)[0]
        tr.pretrain_g(x, y, a, s)
        torch.cuda.synchronize()
        g = G.get_default_graph()
        n = sum(1 for o in g.ops if isinstance(o, O.Conv2dOp) and o._fused_bias)
        assert n == (1 if fuse else 0), (fuse, n)
        outs.append((frame, {k: sess.get_value(v) for k, v in g.variables.items()}))
    (f0, w0), (f1, w1) = outs
    tol = 2e-6 if dtype == 'f32' else 1e-2      # bf16: the separate op reads the deconv output ROUNDED to bf16, the epilogue its float32 accumulator
    assert float(np.abs(f1 - f0).max()) <= tol, float(np.abs(f1 - f0).max())
    for k in w0:
        assert torch.isfinite(w1[k]).all(), k


@pytest.mark.timeout(900)
def test_lookahead_step_matches_oracle_live_at_full_size():
    """The bench's step since round 5 - Trainer.train_d(..., next_g=...) then train_g - at BASELINE config 2's size (batch 32,
    RMSProp so that the updated weights are comparable): the D step on samples A runs the generator on the pair batch
    [B ; A] (batch 64, BatchNorm statistics per half), the G step on samples B starts behind that pass.  Against the fp64
    oracle doing what train.py:241-263 does - train_d(A), then train_g(B) - at 1e-3: the generated frames the G step
    returns, both steps' per-variable gradient norms, every weight after the two updates; then two more iterations (capture,
    replay) with the oracle stepping along, compared as far as the arithmetic allows (see below)."""
    import torch
    from oracle import models as OM
    from oracle.trainer import OracleTrainer
    from action_conditioned_gans_amd import optim, train as T
    B, S, K = 32, 64, 5
    params = OM.init_params(True, batch=B, img=S, ksize=K, seed=9, dtype=torch.float32)
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = gpu_session()
    tr = T.Trainer(sess, True, 'bce', 'rmsprop', True, batch_size=B, img_size=S, ksize=K)
    assert tr.lookahead
    sess.run(G.global_variables_initializer())
    g = G.get_default_graph()
    for n, v in g.variables.items():
        sess.set_value(v, params[n])
    rng = np.random.default_rng(33)
    mk = lambda: (rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32), rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32),     # noqa: E731
                  rng.standard_normal((B, 10)).astype(np.float32), rng.standard_normal((B, 5)).astype(np.float32))
    (xa, ya, aa, _), (xb, yb, ab, sb) = mk(), mk()
    td = lambda t: torch.from_numpy(t).double()     # noqa: E731
    torch.set_num_threads(16)
    ot = OracleTrainer({k: v.double() for k, v in params.items()}, True, 'bce', 'rmsprop', True, K)
    # Iteration 0 is compared in full.  Later iterations only on the frames, and only the next one: this training is CHAOTIC at
    # float32 rounding level - two plain runs that differ in nothing but the summation order of one reduction are 2e-4 / 2e-3 /
    # 8e-3 / 2e-2 apart in their frames after 2 / 3 / 4 / 5 iterations and 1e-2 in their G gradients after two (one sign flip in
    # the kinked L1 / GDL frame losses moves the whole G gradient by 1.6e-3; profiles/r5/e_lookahead_divergence.txt) - so a
    # per-variable 1e-3 bar against ANY other arithmetic cannot hold beyond the first step, look-ahead or not.
    for it in range(3):
        tr.train_d(xa, ya, aa, next_g=(xb, ab))
        ot.train_d(td(xa), td(ya), td(aa))
        if it == 0:
            TC.check_norms(TC.flat_grad_norms(sess, tr.d_opt_op), {'dgrad_norm/' + k: v.norm() for k, v in ot.last_grads.items()},
                           'dgrad_norm/', 1e-3, 'D grad (look-ahead)')
        frames = tr.train_g(xb, yb, ab, sb)
        oframe = ot.train_g(td(xb), td(yb), td(ab), td(sb))
        if it <= 1:
            assert TC.rel(frames, oframe.numpy()) <= 1e-3, (it, TC.rel(frames, oframe.numpy()))
        if it == 0:
            TC.check_norms(TC.flat_grad_norms(sess, tr.g_opt_op), {'ggrad_norm/' + k: v.norm() for k, v in ot.last_grads.items()},
                           'ggrad_norm/', 1e-3, 'G grad (look-ahead)')
            for n, v in g.variables.items():           # every filter's UPDATE over the first iteration (D step + clip, G step)
                if not n.endswith('weights'):
                    continue
                got, want, init = sess.get_value(v).double(), ot.p[n], params[n].double()
                upd = (want - init).norm().item()
                # (the update vector, not single elements: RMSProp's step saturates at lr * sqrt(10) for |g| >> 3, so where a kinked
                # loss or a ReLU at 0 moves one gradient element, that element's update moves by up to a whole step)
                assert (got - want).norm().item() <= 2e-2 * upd + 1e-9, (n, (got - want).norm().item(), upd)
        assert np.isfinite(frames).all()
    progs = sorted(sum(len(seg) for kind, seg in p.segments if kind == 'dev') for p in sess._programs.values())
    assert len(progs) == 2 and all(p.graphs is not None for p in sess._programs.values()), progs     # both look-ahead programs were captured and replayed
    sess.close()


@pytest.mark.parametrize('dtype,dna,ksize', [('f32', True, 5), ('f32', False, 5), ('bf16', True, 5), ('f32', True, 6), ('bf16', True, 11)])
def test_lookahead_equals_the_plain_call_path_on_the_gpu(dtype, dna, ksize):
    """Look-ahead on / off from the same weights on distinct D-step and G-step samples, batch 8, four iterations (eager,
    capture, two replays).  After ONE iteration the two agree to rounding: float32 frames 1e-5, D gradient 1e-4, G gradient
    5e-3 (a kink of the L1 / GDL losses moves it by 1.6e-3), filters 2e-5 of their scale; bf16 (storage rounding) frames 5e-3,
    filters 2e-2.  Over more iterations the training is chaotic at rounding level (profiles/r5/e_lookahead_divergence.txt: a plain
    run that differs only in the summation order of the split-K reductions drifts as fast as the look-ahead run): finiteness and a
    sanity bound only."""
    from action_conditioned_gans_amd import optim, train as T
    x, y, a, s = TC.MG.inputs(8)
    xb, yb, ab, sb = [np.ascontiguousarray(np.roll(t, 3, axis=0)[::-1]) for t in (y, x, a, s)]
    nrel = lambda got, want: float(np.linalg.norm(np.asarray(got, np.float64) - np.asarray(want, np.float64)) / np.linalg.norm(np.asarray(want, np.float64)))  # noqa: E731

    def run(use, **kw):
        G.reset_default_graph()
        optim.set_data_parallel(1)
        sess = gpu_session(dtype=dtype, **kw)
        tr = T.Trainer(sess, True, 'bce', 'rmsprop', dna, batch_size=8, ksize=ksize)       # (k >= 6: the 16-lane-row DNA kernel writes the pair's pixels)
        sess.run(G.global_variables_initializer())
        grad = lambda name: [sess._materialize(t).clone().cpu().double() for t in G.get_default_graph().state if t.name == name][0]   # noqa: E731
        first = None
        for it in range(4):
            tr.train_d(x, y, a, next_g=(xb, ab) if use else None)
            dg = grad('d_opt/flat_grad') if it == 0 else None
            frames = tr.train_g(xb, yb, ab, sb)
            if it == 0:
                first = (np.array(frames, copy=True), dg, grad('g_opt/flat_grad'), {n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()})
        final = {n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()}
        if use:
            assert all(p.graphs is not None for p in sess._programs.values())      # the look-ahead programs were captured and replayed
        sess.close()
        return first, frames, final
    (f0, dg0, gg0, w0), fl0, wl0 = run(False)
    (f1, dg1, gg1, w1), fl1, wl1 = run(True)
    if dtype == 'f32':
        assert nrel(f1, f0) <= 1e-5 and nrel(dg1, dg0) <= 1e-4 and nrel(gg1, gg0) <= 5e-3, (nrel(f1, f0), nrel(dg1, dg0), nrel(gg1, gg0))
    else:
        assert nrel(f1, f0) <= 5e-3, nrel(f1, f0)
    init = {n: v.value.double() for n, v in G.get_default_graph().variables.items()}      # (same seed: the same initial weights in every run)
    for n in w0:
        if n.endswith('weights'):      # (a beta after one step is its first update: no scale to compare against)
            if dtype == 'f32':         # the UPDATE vectors agree (single elements can move by a whole saturated RMSProp step at a kink)
                upd = (w0[n].double() - init[n]).norm().item()
                assert (w1[n].double() - w0[n].double()).norm().item() <= 2e-2 * upd + 1e-9, (n, upd)
            else:
                d = float((w1[n].double() - w0[n].double()).abs().max())
                assert d <= 2e-2 * max(float(w0[n].abs().max()), 1e-3), (n, d)
        assert bool(torch.isfinite(wl1[n]).all()), n
    assert np.isfinite(fl1).all()
    # after four iterations: the same trajectory as far as a chaotic training allows - two PLAIN runs that differ only in a
    # summation order are anywhere between 1e-4 and 1e-1 apart by then, depending on which kinks of the L1 / GDL losses flip
    # (tests/lookahead_divergence.py -> profiles/r5/e_lookahead_divergence.txt), so this is a sanity bound, not a parity bar
    assert nrel(fl1, fl0) <= 0.5, nrel(fl1, fl0)


def test_lookahead_pairs_successive_discriminator_steps_on_the_gpu():
    """n_critic = 3 under --loss wass on the GPU (float32, batch 8, RMSProp + clip): D steps paired (D1 -> D2), (D3 -> G) through
    train_d(next_d=) against the plain call path - after ONE iteration (three D updates, one G update) the generated frames agree
    to 1e-5 and the filters' update vectors to 2e-2; three more iterations run through capture and replay and stay finite.  Three
    programs: the pair-pass D step, the generator-free D step, the generator-free G step."""
    from action_conditioned_gans_amd import optim, train as T
    rng = np.random.default_rng(5)
    B = 8
    mk = lambda: (rng.uniform(-1, 1, (B, 64, 64, 3)).astype(np.float32), rng.uniform(-1, 1, (B, 64, 64, 3)).astype(np.float32),     # noqa: E731
                  rng.standard_normal((B, 10)).astype(np.float32))
    ds, g_in = [mk() for _ in range(3)], mk() + (rng.standard_normal((B, 5)).astype(np.float32),)

    def run(use):
        G.reset_default_graph()
        optim.set_data_parallel(1)
        sess = gpu_session()
        tr = T.Trainer(sess, True, 'wass', 'rmsprop', True, batch_size=B, ksize=5)
        sess.run(G.global_variables_initializer())
        first = None
        for it in range(4):
            carried = False
            for j, (x, y, a) in enumerate(ds):
                if use and not carried:
                    tr.train_d(x, y, a, next_d=(g_in[0], g_in[2]) if j == len(ds) - 1 else (ds[j + 1][0], ds[j + 1][2]))
                    carried = True
                else:
                    tr.train_d(x, y, a)
                    carried = False
            frames = tr.train_g(*g_in)
            if it == 0:
                first = (np.array(frames, copy=True), {n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()})
        init = {n: v.value.double() for n, v in G.get_default_graph().variables.items()}
        nprog = len(sess._programs)
        sess.close()
        return first, frames, init, nprog
    (f0, w0), fl0, init, n0 = run(False)
    (f1, w1), fl1, _, n1 = run(True)
    assert (n0, n1) == (2, 3)
    assert float(np.linalg.norm(f1 - f0) / np.linalg.norm(f0)) <= 1e-5
    for n in w0:
        if n.endswith('weights') and n.startswith('g/'):      # (D's filters sit ON the clip bounds after a wass step: their "update" is the clip)
            upd = (w0[n].double() - init[n]).norm().item()
            assert (w1[n].double() - w0[n].double()).norm().item() <= 2e-2 * upd + 1e-9, (n, upd)
        assert bool(torch.isfinite(w1[n]).all()), n
    assert np.isfinite(fl1).all()


def test_host_feeds_go_through_the_staged_upload():
    """Session.upload_many (round 5): the host arrays of a feed are packed into one pinned staging buffer and copied on a copy
    stream of their own.  The device copies must hold the arrays' values (float64 and integer inputs converted, non-contiguous
    ones gathered), every view 16-byte aligned; device tensors pass through; more uploads than the ring has slots (24) reuse
    staging buffers only after the compute stream is past their last reader - checked by keeping the stream busy with work
    that reads each upload after a delay."""
    G.reset_default_graph()
    sess = gpu_session()
    rng = np.random.default_rng(11)
    a = rng.standard_normal((3, 5, 7)).astype(np.float32)
    b = rng.standard_normal((4, 10))                               # float64
    c = np.arange(24, dtype=np.int64).reshape(2, 12)
    d = rng.standard_normal((8, 6)).astype(np.float32)[:, ::2]     # strided
    e = torch.full((3,), 2.5, device='cuda:0', dtype=torch.float64)
    out = sess.upload_many([a, b, c, d, e])
    for got, want in zip(out, (a, b, c, d)):
        assert got.is_cuda and got.dtype == torch.float32 and got.data_ptr() % 16 == 0
        assert np.array_equal(got.cpu().numpy(), np.asarray(want, np.float32))
    assert out[4].dtype == torch.float32 and float(out[4][0]) == 2.5
    assert torch.equal(sess.upload(a), out[0])
    # ring reuse under a busy compute stream: 80 uploads of distinct data, each consumed by queued work
    big = torch.zeros(1 << 24, device='cuda:0')
    sums, want = [], []
    for k in range(80):
        x = np.full((1 << 16,), float(k), np.float32)
        dev = sess.upload(x)
        for _ in range(4):
            big.add_(1.0)                                          # keeps the compute stream behind the host
        sums.append(dev.double().sum())
        want.append(float(k) * (1 << 16))
    torch.cuda.synchronize()
    assert [float(s) for s in sums] == want
    sess.close()
