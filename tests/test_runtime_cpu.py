"""CPU suite for the host runtime (graph building, fetch pruning, gradient wiring, flat buffers,
optimizer sequencing, scopes).  The kernels come from the C oracle through the SAME C ABI - injected
here, by the test, as a stand-in device library; the product itself has no such path."""
import ctypes

import numpy as np
import pytest
import torch

import train_cases as TC
from oracle import cbind

from action_conditioned_gans_amd import _lib
from action_conditioned_gans_amd import graph as G
from action_conditioned_gans_amd import models as M
from action_conditioned_gans_amd import ops as O
from action_conditioned_gans_amd import optim
from action_conditioned_gans_amd import train as T


def cpu_session(**kw):
    return G.Session(device='cpu', lib=cbind.load(), **kw)


@pytest.mark.parametrize('name', ['c1_plain_l1', 'c2_dna_bce_adam', 'c4_dna_wass_rmsprop'])
def test_trainer_matches_golden(name):
    TC.case_golden(cpu_session, name, 1e-5)


@pytest.mark.parametrize('name', ['c2_dna_bce_adam', 'plain_adv_bce_rmsprop'])
def test_pretrain_step_matches_golden(name):
    TC.case_pretrain_golden(cpu_session, name, 1e-5)


def test_oracle_still_matches_golden():
    """Pins the fp64 restatement itself: regenerating a fixture must reproduce the committed file."""
    import make_golden as MG
    for name in ('c2_dna_bce_adam',):
        fresh, gold = MG.make_case(name), TC.golden(name)
        assert set(fresh) == set(gold)
        for k in gold:
            assert TC.rel(fresh[k], gold[k]) <= 1e-9 or abs(float(np.max(np.abs(fresh[k] - gold[k])))) <= 1e-12, k


def test_fetch_pruning_matches_tf_semantics():
    """what each sess.run computes, and - as important - what it does not."""
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam')
    fd = tr._feed(*TC.MG.inputs(2))
    names = lambda fetch: [n for _, _, n in TC.program_op_names(sess, fetch, fd)]
    kinds = lambda fetch: [k for _, k, _ in TC.program_op_names(sess, fetch, fd)]
    test_ops = names([tr.g_next_frame])
    assert test_ops and all(n.startswith('g/') for n in test_ops), test_ops          # generator forward only
    d_ops = names([tr.d_opt_op, tr.clip_d])
    assert not any(n.startswith('g/') and ('/dgrad' in n or '/wgrad' in n or '/bwd' in n) for n in d_ops)  # no backprop into G
    assert 'd/conv1/conv2d/dgrad' not in d_ops                                       # images need no gradient
    # batched D step: D runs once on [fake ; real] (BatchNorm per half), so each D variable has ONE wgrad
    assert sum(n.endswith('/wgrad') and n.startswith('d/conv2/') for n in d_ops) == 1
    assert tr.d_out_both.shape == (4, 2, 2, 1)
    g_kinds, g_ops = kinds([tr.g_opt_op, tr.g_next_frame]), names([tr.g_opt_op, tr.g_next_frame])
    assert not any(n.startswith('d/') and n.endswith('/wgrad') for n in g_ops)        # D weights are frozen in the G step
    assert 'd/conv1/conv2d/dgrad' in g_ops                                            # gradient flows through D into G
    assert sum(n == 'd/conv1/conv2d' for n in g_ops) == 1                             # D(real) is pruned
    assert 'StepOp' in g_kinds


def test_unbatched_d_step_matches_golden_and_accumulates_two_wgrads():
    """The reference's literal structure (two D calls sharing variables): second wgrad accumulates in place."""
    import make_golden as MG
    adv, loss, opt, dna, batch, ksize = MG.CASES['c2_dna_bce_adam']
    gold = TC.golden('c2_dna_bce_adam')
    from action_conditioned_gans_amd import train as T
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = cpu_session()
    tr = T.Trainer(sess, adv, loss, opt, dna, batch_size=batch, ksize=ksize, batched_d=False)
    sess.run(G.global_variables_initializer())
    from oracle import models as OM
    params = OM.init_params(dna, batch=batch, ksize=ksize, seed=MG.PARAM_SEED, dtype=torch.float32)
    for n, v in G.get_default_graph().variables.items():
        sess.set_value(v, params[n])
    x, y, a, s = MG.inputs(batch)
    d_ops = [n for _, _, n in TC.program_op_names(sess, [tr.d_opt_op, tr.clip_d], tr._feed(x, y, a))]
    assert sum(n.endswith('/wgrad') and n.startswith('d/conv2/') for n in d_ops) == 2
    summ = tr.train_d(x, y, a, summarize=True)
    assert abs(summ['discriminator_loss'] - gold['d_loss']) <= 1e-5
    TC.check_norms(TC.flat_grad_norms(sess, tr.d_opt_op), gold, 'dgrad_norm/', 1e-4, 'D grad (unbatched)')


def test_weight_gradient_reductions_share_one_launch():
    """The split-K slab reductions of the conv weight gradients are deferred to ONE op per optimizer step - per
    all-reduce bucket under data parallelism, each bucket's all-reduce ordered behind its reduction."""
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam')
    x, y, a, s = TC.MG.inputs(2)
    ops_ = TC.program_op_names(sess, [tr.g_opt_op, tr.g_next_frame], tr._feed(x, y, a, s))
    red = [o for o in ops_ if o[1] == 'WgradReduceOp']
    wg = [o for o in ops_ if o[1] == 'ConvWgradOp']
    upd = [o for o in ops_ if o[1] == 'StepOp']
    assert len(red) == 1 and len(wg) >= 10 and len(upd) == 1
    assert max(o[0] for o in wg) < red[0][0] < upd[0][0]
    prog_ops = [op for kind, seg in sess._compile(sess._flatten([tr.g_opt_op]), []).segments if kind == 'dev' for op, _ in seg]
    r = [op for op in prog_ops if isinstance(op, O.WgradReduceOp)][0]
    assert len(r._keep) == 1 and r._keep[0][1] == len(wg)       # the oracle splits every layer at batch 2: all deferred
    # data parallel: one reduction per bucket, the bucket's all-reduce right behind it
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam', world_size=2, collectives='side')     # two buckets per optimizer
    ops_ = TC.program_op_names(sess, [tr.g_opt_op, tr.g_next_frame], tr._feed(x, y, a, s))
    red = [o for o in ops_ if o[1] == 'WgradReduceOp']
    ar = [o for o in ops_ if o[1] == 'AllReduceOp']
    assert len(red) == len(ar) == 2
    assert all(op.side_stream for op in G.get_default_graph().ops if isinstance(op, optim.AllReduceOp))
    for k, r_ in enumerate(red):
        later = [o for o in ar if o[0] > r_[0]]
        assert later and min(o[0] for o in later) - r_[0] == 0.5, (k, r_, ar)
    # in-order collectives on the compute stream: ONE bucket per optimizer, its all-reduce behind the single reduction
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam', world_size=2)
    ops_ = TC.program_op_names(sess, [tr.g_opt_op, tr.g_next_frame], tr._feed(x, y, a, s))
    red = [o for o in ops_ if o[1] == 'WgradReduceOp']
    ar = [o for o in ops_ if o[1] == 'AllReduceOp']
    assert len(red) == 1 and len(ar) == 1 and ar[0][0] - red[0][0] == 0.5


def test_clip_is_fused_after_the_update():
    """Defect D6: the reference leaves update/clip unordered; here clip follows the update in-kernel."""
    sess, tr = TC.build_trainer(cpu_session, 'c4_dna_wass_rmsprop')
    x, y, a, s = TC.MG.inputs(2)
    tr.train_d(x, y, a)
    for v in tr.d_vars:
        val = sess.get_value(v)
        assert val.abs().max().item() <= 0.01 + 1e-9, v.name
    prog = [p for k, p in sess._programs.items()][-1]
    assert tr.d_opt_op.program_clip is not None
    # clip fetched alone still works (standalone kernel)
    sess.set_value(tr.d_vars[0], torch.full(tr.d_vars[0].shape, 0.5))
    sess.run(tr.clip_d)
    assert sess.get_value(tr.d_vars[0]).max().item() <= 0.01 + 1e-9


def test_variable_sharing_and_errors():
    G.reset_default_graph()
    x = G.placeholder((2, 64, 64, 6))
    a = G.placeholder((2, 10))
    M.build_discriminator(x, a, reuse=False)
    n = len(G.get_default_graph().variables)
    M.build_discriminator(x, a, reuse=True)
    assert len(G.get_default_graph().variables) == n == 12                     # 12 tensors in d/
    with pytest.raises(ValueError):
        M.build_discriminator(x, a, reuse=False)                              # already exists
    G.reset_default_graph()
    with pytest.raises(ValueError):
        M.build_discriminator(x, a, reuse=True)                               # does not exist yet
    with pytest.raises(ValueError):
        O.build_g_adv_loss(x, 'hinge')                                        # ops.py:35
    with pytest.raises(ValueError):
        O.build_d_loss(x, x, 'hinge')                                         # ops.py:47
    with pytest.raises(ValueError):
        O.conv2d(G.placeholder((2, 3, 3, 4)), 5, [4, 4], padding='VALID', scope='too_big')
    with pytest.raises(ValueError):
        O.conv2d(G.placeholder((2, 8, 8, 4)), 5, [3, 3], padding='REFLECT', scope='bad_pad')


def test_variable_inventory_matches_survey_appendix_c():
    for dna, n_g, p_g in ((True, 22, 2871694), (False, 16, 8676611)):
        G.reset_default_graph()
        x = G.placeholder((2, 64, 64, 3))
        a = G.placeholder((2, 10))
        if dna:
            frame, state = M.build_generator_transform(x, a, batch_size=2)
            assert state.shape == (2, 5)
        else:
            frame = M.build_generator(x, a)
        assert frame.shape == (2, 64, 64, 3)
        logits = M.build_discriminator(O.concat([x, frame]), a)
        assert logits.shape == (2, 2, 2, 1)
        g = G.get_default_graph()
        gv, dv = g.trainable_variables('g'), g.trainable_variables('d')
        assert (len(gv), sum(v.numel for v in gv)) == (n_g, p_g)
        assert (len(dv), sum(v.numel for v in dv)) == (12, 4755137)
    # train.py:54 builds the DNA generator with ksize=6 -> tconv4 [5,5,36,128]
    G.reset_default_graph()
    M.build_generator_transform(G.placeholder((2, 64, 64, 3)), G.placeholder((2, 10)), ksize=6)
    assert G.get_default_graph().variables['g/tconv4/weights'].shape == (5, 5, 36, 128)


def test_uninitialized_and_missing_library():
    G.reset_default_graph()
    x = G.placeholder((1, 8, 8, 3))
    y = O.conv2d(x, 4, [3, 3], scope='c')
    sess = cpu_session()
    with pytest.raises(RuntimeError):
        sess.run(y, {x: np.zeros((1, 8, 8, 3), np.float32)})                  # initializer not run
    with pytest.raises(RuntimeError):
        G.Session(device='cpu')                                               # product path: no CPU fallback


def test_checkpoint_roundtrip_and_rollout(tmp_path):
    """Saver (train.py:215,274; test.py:29-30): weights AND optimizer slots survive a save/restore, so a resumed
    run continues bit-identically; test_sequence is the recursive rollout of train.py:157-176."""
    from action_conditioned_gans_amd.saver import Saver, latest_checkpoint
    x, y, a, s = TC.MG.inputs(2)
    sess, tr = TC.build_trainer(cpu_session, 'c4_dna_wass_rmsprop')
    tr.train_d(x, y, a)
    tr.train_g(x, y, a, s)
    path = Saver().save(sess, str(tmp_path / 'model100'))
    Saver().save(sess, str(tmp_path / 'model20'))
    assert latest_checkpoint(str(tmp_path)) == str(tmp_path / 'model100') and path.endswith('model100.npz')
    tr.train_d(x, y, a)
    tr.train_g(x, y, a, s)
    want = {n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()}
    sess2, tr2 = TC.build_trainer(cpu_session, 'c4_dna_wass_rmsprop')
    Saver().restore(sess2, str(tmp_path / 'model100'))
    tr2.train_d(x, y, a)
    tr2.train_g(x, y, a, s)
    for n, v in G.get_default_graph().variables.items():
        assert torch.equal(sess2.get_value(v), want[n]), n
    rng = np.random.default_rng(3)
    frames = rng.uniform(-1, 1, (2, 4, 64, 64, 3)).astype(np.float32)
    acts = rng.standard_normal((2, 4, 10)).astype(np.float32)
    pred, summ = tr2.test_sequence(frames, frames, acts)
    assert pred.shape == (2, 3, 64, 64, 3) and np.isfinite(pred).all() and 'g_psnr' in summ


def test_saver_keeps_five_checkpoints_and_writes_in_the_background(tmp_path):
    """tf.train.Saver() keeps the 5 most recent checkpoints (its default max_to_keep; train.py:215 passes nothing) - 600 saves of
    ~90 MB at the reference's cadence otherwise.  Saver.save(background=True), what train() uses: the state is copied when save is
    called, a writer thread writes the files in order (atomically: .tmp then rename), wait() returns when they are on disk, a
    restore waits by itself, and a write that fails is raised by the next save / wait."""
    import os
    from action_conditioned_gans_amd.saver import Saver, latest_checkpoint
    x, y, a, s = TC.MG.inputs(2)
    sess, tr = TC.build_trainer(cpu_session, 'c4_dna_wass_rmsprop')
    saver = Saver()
    states = []
    for k in range(8):
        tr.train_d(x, y, a)
        states.append({n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()})
        saver.save(sess, str(tmp_path / ('model%d' % (100 * k))), background=True)
    saver.wait()
    files = sorted(f for f in os.listdir(str(tmp_path)))
    assert files == sorted('model%d.npz' % (100 * k) for k in range(3, 8)), files          # the five newest, no .tmp left
    assert latest_checkpoint(str(tmp_path)) == str(tmp_path / 'model700')
    # each file holds the state AT ITS save call, not a later one
    for k in (3, 7):
        sess2, tr2 = TC.build_trainer(cpu_session, 'c4_dna_wass_rmsprop')
        Saver().restore(sess2, str(tmp_path / ('model%d' % (100 * k))))
        for n, v in G.get_default_graph().variables.items():
            assert torch.equal(sess2.get_value(v), states[k][n]), (k, n)
    # max_to_keep=None keeps everything; a synchronous save after background ones stays in order
    keep_all = Saver(max_to_keep=None)
    for k in range(7):
        keep_all.save(sess2, str(tmp_path / 'all' / ('m%d' % k)), background=k % 2 == 0)
    keep_all.wait()
    assert len(os.listdir(str(tmp_path / 'all'))) == 7
    # a failing write surfaces
    blocked = tmp_path / 'blocked'
    blocked.write_text('a file where a directory is needed')
    bad = Saver()
    with pytest.raises(OSError):
        bad.save(sess2, str(blocked / 'model0'), background=True)
        bad.wait()


def test_feeds_missing_unused_and_aliased():
    """TF feed semantics of the session: a placeholder the program reads must be fed (ValueError, as TF's "You must feed a value
    for placeholder tensor"); a fed placeholder the program does not read is validated and ignored (the D step is fed next_state
    like the reference does); a concatenation of fed placeholders is written by the feed itself (Graph.add_feed_alias)."""
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam', batch=2)
    x, y, a, s = TC.MG.inputs(2)
    with pytest.raises(ValueError, match='must feed'):
        sess.run(tr.g_next_frame, {tr.img_ph: x, tr._img_pad: x})                 # actions missing
    with pytest.raises(ValueError, match='shape'):
        bad = tr._feed(x, y, a)
        bad[tr.next_state] = np.zeros((2, 4), np.float32)
        sess.run([tr.d_opt_op, tr.clip_d], bad)
    tr.train_d(x, y, a)                                                           # next_state fed, unused: fine
    g = G.get_default_graph()
    real = [o for o in g.ops if o.name == 'd_in_real'][0]
    gen = [o for o in g.ops if o.name == 'd_in_gen'][0]
    assert len(real.fed_inputs) == 2 and len(gen.fed_inputs) == 1
    got = sess._materialize(real.outputs[0])
    assert torch.equal(got[..., 0:3], torch.from_numpy(x)) and torch.equal(got[..., 3:6], torch.from_numpy(y)) and bool((got[..., 6:] == 0).all())
    assert torch.equal(sess._materialize(gen.outputs[0])[..., 0:3], torch.from_numpy(x))
    # the generated frame reaches D's input out of the DNA kernel (acg_dna_fwd out2), and its gradient goes back into the DNA
    # backward as a window of d(D input) (dout2): no concat, slice or add op is launched for it
    assert gen.by_producer and gen.bind(sess.rt) is None
    tr.train_g(x, y, a, s)
    frame = sess._materialize(tr.g_next_frame)
    assert torch.equal(sess._materialize(gen.outputs[0])[..., 3:6], frame)
    bwd = [o for o in g.ops if o.name == 'g/dna/bwd'][0]
    assert bwd.dout2 is not None and bwd.dout2[1] == 3
    names = {o.name for o in g.ops if id(o) in sess.rt.program_ops}
    assert 'g/dna/bwd' in names and not any(n.startswith('d_in_gen/bwd') for n in names), sorted(n for n in names if 'd_in' in n)
    # the tiled action channels of the concatenated feature maps come from the feed as well (g: [B,4,4,256+10]; d on the joined
    # batch: [2B,16,16,128+10], the same actions for the fake and the real half)
    cats = [o for o in g.ops if isinstance(o, O.ConcatActionsOp)]
    assert len(cats) >= 2 and all(o.fed_inputs for o in cats), [(o.name, bool(o.fed_inputs)) for o in cats]
    for o in cats:
        t = sess._materialize(o.outputs[0])
        if t.shape[0] == 0 or id(o) not in sess.rt.program_ops:
            continue
        want = torch.from_numpy(a)[torch.arange(t.shape[0]) % 2]          # (the joined batch repeats the two samples)
        got_a = t[..., o.c:o.c + a.shape[1]]
        assert torch.equal(got_a, want[:, None, None, :].expand_as(got_a)), o.name


def test_side_chain_flags_and_hoisting():
    """Graph.side_branch: the DNA state head, its loss and all their gradient ops are flagged for the side stream, nothing
    else is; Session._hoist_side_chains moves them to where their producers allow without breaking any dependency."""
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam', batch=2)
    g = G.get_default_graph()
    side = [o for o in g.ops if o.side_stream]
    assert side and all(('sconv' in o.name or 'g_state_loss' in o.name) for o in side), [o.name for o in side][:8]
    assert all(o.side_stream for o in g.ops if 'sconv' in o.name and not o.name.startswith('g_opt')), \
        [o.name for o in g.ops if 'sconv' in o.name and not o.side_stream]
    needed = [o for o in g.ops if not isinstance(o, G.InitOp)]
    ops = sorted(needed, key=lambda o: (o.run_last, o.index))
    out = sess._hoist_side_chains(list(ops))
    assert sorted(map(id, out)) == sorted(map(id, ops))
    where = {id(o): k for k, o in enumerate(out)}
    for o in out:
        for d in [t.op for t in o.inputs if t.op is not None] + list(o.control_inputs):
            if id(d) in where:
                assert where[id(d)] < where[id(o)], (d.name, o.name)
    names = [o.name for o in out]
    first_bwd_side = min(k for k, o in enumerate(out) if o.side_stream and o.name.endswith('g_state_loss/grad'))
    dna_bwd = names.index('g/dna/bwd')
    assert first_bwd_side < dna_bwd, (first_bwd_side, dna_bwd)      # the backward side chain starts before the frame decoder's
    assert [o for o in out if o.side_stream] == [o for o in ops if o.side_stream]      # its own order is kept


def test_c_abi_exports_every_declared_symbol():
    """include/acgan_hip.h <-> both libraries: every declared entry point is exported (no compute here)."""
    import re, os, ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'acgan_hip.h')).read()
    tuning = re.findall(r'#ifdef ACG_TUNING(.*?)#endif', header, flags=re.S)
    header = re.sub(r'#ifdef ACG_TUNING.*?#endif', '', header, flags=re.S)     # tuning builds only: not part of the shipped ABI
    declared = set(re.findall(r'\b(acg_[a-z0-9_]+)\s*\(', header))
    declared -= {'acg_conv_desc', 'acg_stream_t', 'acg_edge_t'}
    assert declared == set(_lib.SIGNATURES), sorted(declared ^ set(_lib.SIGNATURES))
    hip = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(hip, name), 'libacgan_hip.so lacks ' + name
    for name in set(re.findall(r'\b(acg_[a-z0-9_]+)\s*\(', ' '.join(tuning))):
        assert not hasattr(hip, name), 'the shipped library exports the tuning hook ' + name
    assert _lib.get().version() == _lib.ABI_VERSION
    assert cbind.load().version() == _lib.ABI_VERSION


def test_cdna_transformation_layer_and_gradients():
    """ops.cdna_transformation (reference ops.py:52-98) - linear layer `cdna_params`, kernel
    normalisation, per-sample depthwise transform, M outputs - and its gradients into the layer's weights, against
    autograd on the torch restatement.  Only pieces 0 and 2 enter the loss: piece 1's gradient is the zero window."""
    from oracle import tf_ops as OT
    B, H, W, C, F, M, K = 2, 10, 9, 3, 6, 3, 5
    rng = np.random.default_rng(4)
    img = rng.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    feat = rng.standard_normal((B, F)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = cpu_session()
    img_ph, feat_ph, tgt_ph = G.placeholder((B, H, W, C), 'img'), G.placeholder((B, F), 'feat'), G.placeholder((B, H, W, C), 'tgt')
    with O.variable_scope('g'):
        pieces = O.cdna_transformation(img_ph, feat_ph, M, C, ksize=K)
    assert len(pieces) == M and all(p.shape == (B, H, W, C) for p in pieces)
    loss = O.l2_norm(pieces[0], tgt_ph) + O.l1_norm(pieces[2], tgt_ph) * 0.5
    step = optim.RMSPropOptimizer(1e-3).minimize(loss, G.get_default_graph().trainable_variables('g'))
    sess.run(G.global_variables_initializer())
    g = G.get_default_graph()
    wv, bv = g.variables['g/cdna_params/weights'], g.variables['g/cdna_params/biases']
    assert wv.shape == (1, 1, F, K * K * M) and bv.shape == (K * K * M,)
    w0 = (torch.from_numpy(rng.standard_normal((F, K * K * M)).astype(np.float32)) * 0.5)
    b0 = torch.from_numpy(rng.standard_normal(K * K * M).astype(np.float32) * 0.5 + 0.5)
    sess.set_value(wv, w0.reshape(wv.shape))
    sess.set_value(bv, b0)
    fd = {img_ph: img, feat_ph: feat, tgt_ph: tgt}
    got = sess.run(pieces, fd)
    wd, bd = w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    ref = OT.cdna_transform(torch.from_numpy(feat).double() @ wd + bd, torch.from_numpy(img).double(), M, K)
    for j in range(M):
        assert TC.rel(got[j], ref[j].detach().numpy()) <= 1e-5, j
    t = torch.from_numpy(tgt).double()
    ((ref[0] - t).pow(2).sum().sqrt() + 0.5 * (ref[2] - t).abs().sum()).backward()
    sess.run([step], fd)
    norms = TC.flat_grad_norms(sess, step)
    assert abs(norms['g/cdna_params/weights'] - float(wd.grad.norm())) <= 1e-4 * float(wd.grad.norm())
    assert abs(norms['g/cdna_params/biases'] - float(bd.grad.norm())) <= 1e-4 * float(bd.grad.norm())


def test_capture_fallback_only_for_capture_unsupported_errors():
    """the eager fallback of Session._execute is for "this stack cannot capture that launch" only; a failing
    kernel argument or an unbalanced fork / join inside the capture body is raised, not swallowed."""
    from action_conditioned_gans_amd import _lib, graph as G
    ok = G.Session._capture_unsupported
    assert ok(RuntimeError('HIP error: operation not permitted when stream is capturing'))
    assert ok(RuntimeError('hipErrorStreamCaptureUnsupported'))
    assert ok(RuntimeError('HIP error: operation failed due to a previous error during capture'))
    assert not ok(RuntimeError('hipErrorStreamCaptureUnjoined: capture was not joined'))
    assert not ok(RuntimeError('hipErrorStreamCaptureUnmatched'))
    assert not ok(_lib.AcgError('acg_conv2d_fwd failed (code 1): null pointer while capturing'))
    assert not ok(ValueError('operation not permitted when stream is capturing'))
    # round 5: a collective refused on a capturing stream comes back through RCCL's own error text; other RCCL failures are bugs
    from action_conditioned_gans_amd.comm import CommError
    assert ok(CommError('ncclAllReduce failed: unhandled cuda error (ncclResult 1)'))
    assert not ok(CommError('ncclCommInitRank failed: internal error (ncclResult 3)'))
    try:
        try:
            raise RuntimeError('HIP error: operation not permitted when stream is capturing')
        except RuntimeError:
            raise RuntimeError('capture_end failed')          # the error on top of the first one
    except RuntimeError as chained:
        assert ok(chained)
    try:
        try:
            raise _lib.AcgError('acg_conv2d_fwd failed (code 1): bad argument')
        except _lib.AcgError:
            raise RuntimeError('capture_end failed')
    except RuntimeError as chained:
        assert not ok(chained)
    sess, tr = TC.build_trainer(cpu_session, 'c1_plain_l1')
    x, y, a, s = TC.MG.inputs(2)
    tr.pretrain_g(x, y, a, s)
    sess.use_hip_graphs = True

    def broken(seg):
        raise _lib.AcgError('acg_conv2d_fwd failed (code 1): bad argument')
    sess.capture_segment = broken
    with pytest.raises(_lib.AcgError):
        tr.pretrain_g(x, y, a, s)


def test_every_placeholder_a_program_reads_must_be_fed():
    """TF: "You must feed a value for placeholder tensor".  Includes placeholders that reach an op only through a feed
    alias - the action vector tiled into the concatenated maps by the feed copy itself (repeat_batch(placeholder)
    -> ConcatActionsOp launches nothing, so an un-fed action placeholder used to reuse old data silently)."""
    sess, tr = TC.build_trainer(cpu_session, 'c2_dna_bce_adam')
    fd = tr._feed(*TC.MG.inputs(2))
    programs = {'d': [tr.d_opt_op, tr.clip_d], 'g': [tr.g_opt_op, tr.g_next_frame], 'test': [tr.g_next_frame]}
    reads = {'d': {'current_frame', 'current_frame_conv', 'next_frame', 'action'},
             'g': {'current_frame', 'current_frame_conv', 'next_frame', 'action', 'next_state'},
             'test': {'current_frame', 'current_frame_conv', 'action'}}
    for ph in list(fd):
        rest = {k: v for k, v in fd.items() if k is not ph}
        for key, fetch in programs.items():
            if ph.name in reads[key]:
                with pytest.raises(ValueError, match='You must feed a value for placeholder'):
                    sess.run(fetch, rest)
            else:
                sess.run(fetch, rest)      # a placeholder the pruned program never reads may stay un-fed, as in TF


def test_train_cli_reaches_the_data_parallel_options(tmp_path, monkeypatch):
    """--sync_bn / --exact_global_batch / --dp_collectives / --buckets on the reference's own entry point
    (train.py:311-319 has the four model flags; the DP ones are new) arrive in train() and from there in the graph's
    DataParallel configuration."""
    from action_conditioned_gans_amd import optim
    seen = {}
    monkeypatch.setattr(T, 'train', lambda *a, **kw: seen.update(kw, positional=a))
    T.main(['synthetic', str(tmp_path / 'o1'), '--adv', 'True', '--dna', '--sync_bn', '--dp_collectives', 'stream', '--buckets', '3'])
    assert seen['sync_bn'] is True and seen['exact_global_batch'] is False and seen['dp_collectives'] == 'stream' and seen['buckets'] == 3
    T.main(['synthetic', str(tmp_path / 'o2'), '--exact_global_batch', 'True'])
    assert seen['exact_global_batch'] is True and seen['dp_collectives'] is None and seen['buckets'] == 0
    with pytest.raises(SystemExit):
        T.main(['synthetic', str(tmp_path / 'o3'), '--dp_collectives', 'ring'])
    # and what train() makes of them (the defaults: side-stream collectives in two buckets once there is more than one rank)
    G.reset_default_graph()
    optim.set_data_parallel(4, n_buckets=None, sync_bn=False, exact_global_batch=True, collectives='side')
    dp = G.get_default_graph().collections['data_parallel']
    assert dp.sync_bn and dp.exact_global_batch and dp.collectives == 'side' and dp.n_buckets == 2 and dp.active


def test_slab_handoff_argument_is_compared_by_identity():
    """1 == True, so Session(slab_handoff=1) used to select every split layer; N = 1 hands off nothing."""
    G.reset_default_graph()
    assert cpu_session(slab_handoff=1).rt.slab_handoff == 1
    assert cpu_session(slab_handoff=True).rt.slab_handoff == 1 << 30
    assert cpu_session(slab_handoff='quads').rt.slab_handoff == 1 << 30 and not cpu_session(slab_handoff='quads').rt.slab_rows
    assert cpu_session(slab_handoff=False).rt.slab_handoff == 0
    with pytest.raises(ValueError):
        cpu_session(slab_handoff='rows')


def test_side_stream_join_sees_ops_absorbed_by_their_producer():
    """an op whose launch its producer absorbed (BiasActOp behind a deconv with a fused epilogue) never appears
    in a segment; a main-stream consumer names only that op in its deps.  The producer's launch must report it, or no join
    is issued when the producer ran on the side stream."""
    calls = []

    class FakeLib:
        def stream_edge(self, e, a, b):
            calls.append(('edge', a.value, b.value))

        def stream_edge_create(self, ref):
            pass

    class Stream:
        def __init__(self, v):
            self.cuda_stream = v
    g = G.reset_default_graph()
    t_in = G.Tensor(g, (1,), 'in')
    prod = G.Op(g, 'deconv', [t_in], [G.Tensor(g, (1,), 'conv_out')])
    absorbed = G.Op(g, 'bias_act', [prod.outputs[0]], [G.Tensor(g, (1,), 'y')])
    cons = G.Op(g, 'consumer', [absorbed.outputs[0]], [G.Tensor(g, (1,), 'z')])
    prod.side_stream, prod.absorbed = True, (absorbed,)
    sess = cpu_session()
    sess.side_branches = True
    sess.rt.is_cuda, sess.rt.side_stream, sess.rt.lib = True, Stream(22), FakeLib()
    sess.rt.edge_pool = lambda n: [ctypes.c_void_p(i) for i in range(n)]
    import unittest.mock as mock
    with mock.patch.object(torch.cuda, 'current_stream', lambda dev=None: Stream(11)):
        sess._launch_segment([(prod, lambda s: calls.append(('prod', s.value))), (cons, lambda s: calls.append(('cons', s.value)))])
    assert calls == [('edge', 11, 22), ('prod', 22), ('edge', 22, 11), ('cons', 11)], calls


@pytest.mark.parametrize('dna', [True, False], ids=['dna', 'plain'])
def test_lookahead_generator_pass_matches_the_plain_call_path(dna):
    """Trainer(lookahead=True): train_d(..., next_g=(x_g, a_g)) runs the generator ONCE on the pair batch (G step's samples ;
    D step's samples, BatchNorm statistics per half) and the train_g that follows with those inputs starts behind its forward
    pass.  Same arithmetic as the two separate passes of train.py:241-263: frames, both gradient buffers and the weights after
    two iterations agree with the plain call path to rounding; the programs really differ (the D program runs the pair
    instance, the G program lost its generator forward); anything but the announced inputs takes the plain program."""
    x, y, a, s = TC.MG.inputs(2, img=32)          # (32 x 32 frames: the brute-force C stand-in is slow; the graph is the same)
    x2, y2, a2, s2 = [np.ascontiguousarray(t[::-1]) for t in TC.MG.inputs(2, img=32)]

    def run(use):
        G.reset_default_graph()
        sess = cpu_session()
        tr = T.Trainer(sess, True, 'bce', 'rmsprop', dna, batch_size=2, img_size=32, ksize=5)
        sess.run(G.global_variables_initializer())
        for _ in range(2):
            tr.train_d(x, y, a, next_g=(x2, a2) if use else None)
            frames = tr.train_g(x2, y2, a2, s2)
        g = G.get_default_graph()
        grads = {t.name: sess._materialize(t).clone() for t in g.state if t.name.endswith('flat_grad') and 'pretrain' not in t.name}
        return tr, sess, frames, grads, {n: sess.get_value(v) for n, v in g.variables.items()}
    _, s0, f0, g0, w0 = run(False)
    tr, s1, f1, g1, w1 = run(True)
    assert np.abs(f1 - f0).max() <= 1e-5 * np.abs(f0).max()
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-5 * float(g0[k].abs().max()), k
    for n in w0:
        assert float((w1[n] - w0[n]).abs().max()) <= 1e-6 * max(float(w0[n].abs().max()), 1e-3), n
    ops = lambda sess: sorted(sum(len(seg) for kind, seg in p.segments if kind == 'dev') for p in sess._programs.values())     # noqa: E731
    plain, ahead = ops(s0), ops(s1)
    assert len(plain) == 2 and len(ahead) == 2
    assert max(ahead) < max(plain), (plain, ahead)           # the G program (the longer one) lost the generator's forward pass
    # no feed alias may write into the result of a skipped op: the batch-B concatenation (first half of the pair's) would get the
    # D step's action rows while the pair's own feed writes the G step's there - in ONE fused copy launch on a GPU, a race
    # (seen in round 5: correct on the CPU's sequential copies, 10 % off on the GPU)
    for prog in s1._programs.values():
        skipped_out = {id(t) for o in (getattr(prog, 'skip_ref', None) or ()) for t in o.outputs}
        assert skipped_out
        for lst in prog.alias_copies.values():
            assert not [dst for dst, _, _ in lst if id(dst) in skipped_out]
    # the pair instance's tensors are the storage of the batch-B instance (first half): no second copy of the activations
    half = tr.g_out
    assert half.view_of is not None and half.view_of[0] is tr._g_pair_out and half.view_of[1] == 0
    # not the announced inputs -> the plain program (three programs now), same result as a plain run from the same state
    tr.train_d(x, y, a, next_g=(x2, a2))
    n_before = len(s1._programs)
    tr.train_g(x2.copy(), y2, a2, s2)
    assert len(s1._programs) == n_before + 1
    # a logging D step (summaries read the D step's OWN generated frames) keeps the plain path even when asked
    summ = tr.train_d(x, y, a, summarize=True, next_g=(x2, a2))
    assert summ is not None and tr._announced is None


def test_lookahead_pairs_successive_discriminator_steps():
    """n_critic > 1 (train.py:217-220: five D steps per G step under --loss wass): a D step can announce the NEXT D step's inputs
    (train_d next_d=) - it then runs the generator once for both, and the announced step runs no generator at all (its frames
    wait in the spare rows of the discriminator-input buffer).  Three D steps + one G step, paired (D1 -> D2), (D3 -> G), against
    the plain call path: frames and weights equal to rounding after two iterations; three programs (pair-pass D, generator-free
    D, generator-free G) instead of two."""
    rng = np.random.default_rng(0)
    B = 2
    mk = lambda: (rng.uniform(-1, 1, (B, 32, 32, 3)).astype(np.float32), rng.uniform(-1, 1, (B, 32, 32, 3)).astype(np.float32),     # noqa: E731
                  rng.standard_normal((B, 10)).astype(np.float32))
    ds, g = [mk() for _ in range(3)], mk() + (rng.standard_normal((B, 5)).astype(np.float32),)

    def run(use):
        G.reset_default_graph()
        sess = cpu_session()
        tr = T.Trainer(sess, True, 'wass', 'rmsprop', True, batch_size=B, img_size=32, ksize=5)
        sess.run(G.global_variables_initializer())
        for _ in range(2):
            carried = False
            for j, (x, y, a) in enumerate(ds):
                if use and not carried:
                    tr.train_d(x, y, a, next_d=(g[0], g[2]) if j == len(ds) - 1 else (ds[j + 1][0], ds[j + 1][2]))
                    carried = True
                else:
                    tr.train_d(x, y, a)
                    carried = False
            frames = tr.train_g(*g)
        return frames, {n: sess.get_value(v) for n, v in G.get_default_graph().variables.items()}, len(sess._programs)
    f0, w0, n0 = run(False)
    f1, w1, n1 = run(True)
    assert (n0, n1) == (2, 3)
    assert np.abs(f1 - f0).max() <= 1e-5 * np.abs(f0).max()
    for n in w0:
        assert float((w1[n] - w0[n]).abs().max()) <= 1e-6 * max(float(w0[n].abs().max()), 1e-3), n
