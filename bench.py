#!/usr/bin/env python
"""Headline benchmark: GAN train steps/sec (G+D) on synthetic 64x64x3 T=8 push-style sequences.

One step = n_critic discriminator steps + 1 generator step exactly as train.py:241-263 sequences them
(n_critic = 1 for bce, 5 for wass), at BASELINE config 2 per GPU: batch 32, --adv --loss bce --dna (k=5),
Adam, fp32.  N > 1 (launched by torch.distributed.run, one rank per GPU) is weak scaling: every rank
runs the same per-GPU batch and gradients are all-reduced over RCCL; `value` counts the batch-32 steps
completed by ALL ranks per second.  Inputs are resident in HBM before the timed region.

Besides the contract line it reports
  roofline      every conv/deconv fwd, dgrad, wgrad launch of the step (+ their split-K reductions): algorithmic
                FLOPs (2*B*OH*OW*KH*KW*Cin*Cout per contraction) / their duration IN SITU, against the
                fp32 matrix-core peak (157.3 TFLOP/s; 2.5 PFLOP/s with --dtype bf16);
  roofline_dna  the DNA stencil forward: algorithmic bytes (k*k+6)*sizeof per pixel / its duration in situ,
                against 8 TB/s;
                "in situ" = the kernels' own durations inside the replayed step graphs, operands as the step leaves
                them, reduced from a rocprofv3 kernel trace of `bench.py --trace-run` with the same flags
                (tools/insitu_times.py -> profiles/r4/insitu_<workload>.json, next to the kernel_stats.csv it was
                reduced from; tools/evidence_r4.sh regenerates both).  The live measurement of this process - each
                op relaunched 10x back to back inside a small HIP graph between two events - is cache-hot and reads
                2-20 % faster; it is kept as the labelled second field `hot_relaunch` and is the fallback (said so in
                `timing`) when no in-situ profile of the workload is committed;
  cpu_baseline  the CPU restatement of the reference step (oracle/, torch-CPU fp32; TF-1.0 itself cannot
                run here) on this host's cores, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense
PEAK_HBM_GBS = 8000.0


def slab_mode(v):
    """--slab-handoff: 'off' | 'quads' | 'all' | N (digits) -> the Session's slab_handoff argument."""
    table = {'off': False, 'quads': 'quads', 'all': True}
    if v in table:
        return table[v]
    if v.isdigit():
        return int(v)
    raise argparse.ArgumentTypeError("expected 'off', 'quads', 'all' or a number of slabs, got %r" % v)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--img', type=int, default=64)
    ap.add_argument('--seq_len', type=int, default=8)
    ap.add_argument('--ksize', type=int, default=5)
    ap.add_argument('--loss', default='bce')
    ap.add_argument('--opt', default='adam')
    ap.add_argument('--plain', action='store_true', help='plain generator instead of DNA')
    ap.add_argument('--no-adv', action='store_true')
    ap.add_argument('--no-graphs', action='store_true', help='eager launches instead of HIP-graph replay')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='conv arithmetic: exact fp32 MFMA (BASELINE config 2) or bf16 MFMA operands with fp32 storage/accumulation (configs 3, 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dgrad-limit', action='store_true', help='experiment: input gradients also compute the columns of the tiled action channels (acg_conv_desc dgrad_c unset)')
    ap.add_argument('--profile-repeats', type=int, default=3)
    ap.add_argument('--cpu-steps', type=int, default=20)
    ap.add_argument('--lib', default=None, help='experiment: an alternative build of the library (e.g. the tuning build, whose knobs read the environment)')
    ap.add_argument('--no-epilogue-stats', action='store_true', help='experiment: BatchNorm computes its statistics in a pass of its own instead of taking them from the conv epilogue')
    ap.add_argument('--side-branch', action='store_true', help='experiment: the state head as a parallel branch of the HIP graph (second stream) instead of in line on the main stream; measured slower')
    ap.add_argument('--slab-handoff', default=None, metavar='MODE', type=slab_mode, help="split-K hand-off to the consuming BatchNorm: 'off', 'quads' (layers whose BatchNorm reads the quad slab layout), 'all', or N = either layout for layers split into at most N slabs; default: the Session's")
    ap.add_argument('--no-pair', action='store_true', help='experiment: a layer\'s dgrad and wgrad as two launches instead of one')
    ap.add_argument('--dp-collectives', default=None, choices=['stream', 'side'], help='gradient all-reduces (ncclAllReduce captured into the step\'s HIP graph) on the compute stream in program order, or on a side HIP stream overlapping the rest of backward; default: side with more than one rank')
    ap.add_argument('--buckets', type=int, default=0, help='gradient all-reduce buckets per optimizer (data parallel); 0 = 1 for in-order collectives, 2 for side-stream ones')
    ap.add_argument('--min-seconds', type=float, default=1.0, help='the timed K-step block is repeated until this much time has been measured; the median block is reported')
    ap.add_argument('--exact-global-batch', action='store_true', help='data parallel that reproduces one device at the global batch: SyncBN + GDL scaling + global state-loss norm')
    ap.add_argument('--sync-bn', action='store_true', help='data parallel with BatchNorm statistics of the global batch (one small all-reduce per BatchNorm layer and direction)')
    ap.add_argument('--trace-run', action='store_true', help='nothing but training steps (no rollout, no instrumented pass, no CPU baseline): the run tools/insitu_times.py reduces a rocprofv3 kernel trace of')
    ap.add_argument('--force-dp', action='store_true', help='run the data-parallel machinery (RCCL all-reduce buckets, graph segments) even on one rank')
    ap.add_argument('--bn-grid-exchange', default='auto', choices=['auto', 'on', 'off', 'not-beside-collectives'],
                    help="the one-launch BatchNorm kernels (all blocks of a grid resident at once): 'auto' = with more than one rank under side-stream collectives "
                         "the launches between a bucket's all-reduce and the join take the two-launch kernels (an RCCL ring kernel holds CUs beside them), on "
                         "otherwise; 'not-beside-collectives' = that rule whatever the rank count (with --force-dp: its cost on one rank); 'on' / 'off' force it")
    ap.add_argument('--no-lookahead', action='store_true', help='the two generator forward passes of an iteration as separate batch-B launches (Trainer.train_d without next_g), as before round 5')
    ap.add_argument('--no-api-rates', action='store_true', help='skip the two labelled side numbers (plain call path, numpy-in / numpy-out call path)')
    return ap.parse_args()


_FINGERPRINT = []


def lib_fingerprint():
    """(ABI version, first 16 hex digits of the SHA-256 of the loaded libacgan_hip.so): what a committed in-situ profile must
    have been taken with to stand in for this process's kernels."""
    if not _FINGERPRINT:
        import hashlib
        from action_conditioned_gans_amd import _lib
        lib = _lib.get()
        with open(lib.path, 'rb') as f:
            _FINGERPRINT.append((int(lib.version()), hashlib.sha256(f.read()).hexdigest()[:16]))
    return _FINGERPRINT[0]


def conv_flops(op):
    """FLOPs the contraction is asked for: an input gradient limited to the feature channels of an action-concatenated map
    (acg_conv_desc dgrad_c / adj_dgrad_c) counts those channels only - skipped columns are not work done."""
    d = op.desc
    cin, cout = d.in_c, d.out_c
    if type(op).__name__ == 'ConvDgradOp':
        if op.transposed and d.adj_dgrad_c > 0:
            cout = d.adj_dgrad_c
        elif not op.transposed and d.dgrad_c > 0:
            cin = d.dgrad_c
    return 2.0 * d.batch * d.out_h * d.out_w * d.kh * d.kw * cin * cout


def conv_bytes(op):
    """Algorithmic HBM bytes of one contraction: each operand and the result once, at their storage sizes."""
    size = lambda t: t.numel * (2 if t.dtype == torch.bfloat16 else 4)      # noqa: E731
    return float(sum(size(getattr(op, 'wop', t) if i == 1 and hasattr(op, 'wop') else t) for i, t in enumerate(op.inputs)) + sum(size(t) for t in op.outputs))


def cpu_baseline(args, n_critic):
    """The CPU restatement of the reference step on this host (the checker, timed - never shipped)."""
    from oracle import models as OM
    from oracle.trainer import OracleTrainer
    B, S = args.batch, args.img
    # the GPU box gives one GPU's job a share of 16 host cores; torch's default (every core it can see) oversubscribes it
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(cores)
    params = OM.init_params(not args.plain, batch=2, img=S, ksize=args.ksize, seed=0, dtype=torch.float32)
    tr = OracleTrainer(params, not args.no_adv, args.loss, args.opt, not args.plain, args.ksize)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(B, S, S, 3, generator=g) * 2 - 1
    y = torch.rand(B, S, S, 3, generator=g) * 2 - 1
    a = torch.randn(B, 10, generator=g)
    s = torch.randn(B, 5, generator=g)

    def step():
        for _ in range(n_critic):
            tr.train_d(x, y, a)
        tr.train_g(x, y, a, s)
    step()
    t0 = time.time()
    n = 0
    while n < args.cpu_steps and (n == 0 or time.time() - t0 < 30.0):
        step()
        n += 1
    dt = time.time() - t0
    return {'value': n / dt, 'unit': 'steps/s', 'cores': cores, 'nproc': os.cpu_count(), 'kind': 'port',
            'sample': '%d (D+G) steps at the same shapes (batch %d), torch-CPU fp32 restatement of the TF-1.0 step, '
                      '%d threads (the GPU box gives one GPU\'s job a 16-core share of its %d logical CPUs)' % (n, B, cores, os.cpu_count() or 0)}


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Libraries write there too (librccl: a five-line version banner from ncclCommInitRank,
    # NCCL_DEBUG output, whatever a teardown says): file descriptor 1 points at stderr for the whole run, and the result line
    # goes to the real stdout kept here.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs torch.distributed.run with --nproc-per-node %d' % (args.gpus, args.gpus))
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        # control plane only (rendezvous, barriers, the max over ranks): a CPU gloo group.  Gradients travel over this
        # process's own RCCL communicator (action_conditioned_gans_amd/comm.py), bootstrapped through this group.
        import torch.distributed as dist
        dist.init_process_group('gloo')
    if args.dp_collectives is None:
        args.dp_collectives = 'side' if world > 1 else 'stream'

    from action_conditioned_gans_amd import graph as G, ops as O, optim, train as T
    if args.lib:
        from action_conditioned_gans_amd import _lib
        _lib._LIB = _lib.Library(args.lib)

    O.DGRAD_CHANNEL_LIMIT = not args.no_dgrad_limit
    B, S, dna, adv = args.batch, args.img, not args.plain, not args.no_adv
    n_critic = 5 if args.loss == 'wass' else 1
    G.reset_default_graph()
    optim.set_data_parallel(world, n_buckets=args.buckets, force=args.force_dp, sync_bn=args.sync_bn, exact_global_batch=args.exact_global_batch,
                            collectives=args.dp_collectives)
    sess = G.Session(device=device, side_branches=args.side_branch, **({} if args.slab_handoff is None else {'slab_handoff': args.slab_handoff}), epilogue_stats=not args.no_epilogue_stats, pair_bwd=not args.no_pair, use_hip_graphs=not args.no_graphs, world_size=world, rank=rank, dtype=args.dtype,
                     bn_grid_exchange={'auto': None, 'on': True, 'off': False, 'not-beside-collectives': 'not_beside_collectives'}[args.bn_grid_exchange])
    tr = T.Trainer(sess, adv, args.loss, args.opt, dna, batch_size=B, img_size=S, ksize=args.ksize, seed=0, lookahead=not args.no_lookahead)
    sess.run(G.global_variables_initializer())
    lookahead = tr.lookahead

    # synthetic push-style sequences, resident in HBM; (t, t+1) pairs selected as train.py:231-237,249-250,258-259 does: every D
    # step takes a fresh batch and a fresh selection, the G step a NEW selection on the last D step's batch
    rng = np.random.default_rng(7 + 1000 * rank)
    np.random.seed(7 + rank)
    mask = T.build_all_mask(args.seq_len)
    dev = lambda *ts: tuple(torch.from_numpy(np.ascontiguousarray(t)).to(device) for t in ts)      # noqa: E731
    host_seqs, pool = [], []
    for _ in range(4):
        img = rng.uniform(-1, 1, (B, args.seq_len, S, S, 3)).astype(np.float32)
        acts = rng.standard_normal((B, args.seq_len, 10)).astype(np.float32)
        host_seqs.append((img, acts))
        sm, em = T.select_pairs(np.random.randint, mask, B)
        smg, emg = T.select_pairs(np.random.randint, mask, B)
        d_in, g_in = dev(img[sm], img[em], acts[sm]), dev(img[smg], img[emg], acts[smg], acts[:, :, 5:][emg])
        # the joined inputs of the look-ahead generator pass (G step's samples first: Trainer.train_d `pair`), built once -
        # inputs are resident in HBM before the timed region starts
        pair = (torch.cat([g_in[0], d_in[0]]), torch.cat([g_in[2], d_in[2]])) if lookahead else None
        pool.append([d_in, g_in, pair, None])
    if lookahead and n_critic > 1:      # D step k followed by D step k + 1 (pool order): [successor's samples ; own samples]
        for k, e in enumerate(pool):
            nxt = pool[(k + 1) % len(pool)][0]
            e[3] = (torch.cat([nxt[0], e[0][0]]), torch.cat([nxt[2], e[0][2]]))
    torch.cuda.synchronize()

    def step(i, use_lookahead=lookahead):
        # The sub-steps pair up for the look-ahead generator pass: (D1 -> D2), (D3 -> D4), ..., and the last D step with the G
        # step when it is not already the second of a pair: the first of a pair runs the generator once for both (batch 2 B),
        # the second runs none (Trainer.train_d next_d / next_g).
        carried = False
        for j in range(n_critic):
            k = (i * n_critic + j) % len(pool)
            d_in, g_in, pair_g, pair_d = pool[k]
            last = j == n_critic - 1
            if use_lookahead and not carried:
                if last:
                    tr.train_d(*d_in, next_g=(g_in[0], g_in[2]), pair=pair_g)
                else:
                    nxt = pool[(k + 1) % len(pool)][0]
                    tr.train_d(*d_in, next_d=(nxt[0], nxt[2]), pair=pair_d)
                carried = True
            else:
                tr.train_d(*d_in)
                carried = False
        return tr.train_g(*pool[(i * n_critic + n_critic - 1) % len(pool)][1], device_fetch=True)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(3):            # run 1 eager, run 2 captures the HIP graphs, run 3 replays
        step(i)
    for i in range(args.warmup):
        step(i)

    def timed_block():
        """EXACTLY args.steps steps between two barrier + synchronize brackets; the max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    # a K-step block can be a few tens of milliseconds: repeat it until --min-seconds have been measured (every rank
    # derives the same count from the max-reduced first block) and report the median block
    first, frames = timed_block()
    blocks = [first]
    repeats = int(min(max(np.ceil(args.min_seconds / max(first, 1e-6)), 1), 200))
    for _ in range(repeats - 1):
        dt, frames = timed_block()
        blocks.append(dt)
    elapsed = float(np.median(blocks))
    assert torch.isfinite(frames).all(), 'generated frames are not finite'

    # ---- eval rollout: Trainer.test_sequence, T-1 recursive G-only steps through the
    # reference's numpy-in / numpy-out API (train.py:157-176), after the timed region ---------------------------
    rollout = None
    if rank == 0 and dna and world == 1 and not args.trace_run:      # a one-GPU side metric; multi-GPU runs go straight to the result line
        r_img = rng.uniform(-1, 1, (B, args.seq_len, S, S, 3)).astype(np.float32)
        r_act = rng.standard_normal((B, args.seq_len, 10)).astype(np.float32)
        for _ in range(3):
            tr.test_sequence(r_img, r_img, r_act)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            pred, _ = tr.test_sequence(r_img, r_img, r_act)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        rollout = {'frames_per_s': round(B * (args.seq_len - 1) / dt, 1), 'ms_per_rollout': round(dt * 1e3, 3),
                   'steps': args.seq_len - 1, 'batch': B, 'api': 'Trainer.test_sequence: numpy sequences in, numpy predictions out; between the steps the prediction and the predicted state stay on the device (round 5; per-step numpy round trips as the reference: 9.4 ms per rollout at config 2)'}

    # ---- two labelled side numbers, after the timed region (review r4, missing item 5): the same step through the plain call path
    # (no look-ahead: what `value` was up to round 4), and through the reference's own API exactly as train() drives it
    # (train.py:241-263: host numpy sequences, pair selection, numpy feeds, generated frames fetched to the host every step)
    api_rates = None
    if not args.trace_run and not args.no_api_rates and world == 1:      # (time-bounded loops: with several ranks the step counts - and with them the collectives - would differ per rank)
        def rate(fn, seconds=0.5):
            for i in range(4):
                fn(i)
            barrier()
            t1, n = time.perf_counter(), 0
            while n < 3 or time.perf_counter() - t1 < seconds:
                fn(n)
                n += 1
            barrier()
            return round(world * n / (time.perf_counter() - t1), 2)

        def numpy_step(i):
            img, acts = host_seqs[i % len(host_seqs)]
            for j in range(n_critic):
                sm, em = T.select_pairs(np.random.randint, mask, B)
                last = j == n_critic - 1
                if last:
                    smg, emg = T.select_pairs(np.random.randint, mask, B)
                    g_img, g_act = img[smg], acts[smg]
                tr.train_d(img[sm], img[em], acts[sm], next_g=(g_img, g_act) if (last and lookahead) else None)
            return tr.train_g(g_img, img[emg], g_act, acts[:, :, 5:][emg])
        api_rates = {'unit': 'steps/s (all ranks), ~0.5 s each, after the timed region',
                     'plain_call_path_device_resident': rate(lambda i: step(i, False)) if lookahead else None,
                     'numpy_in_numpy_out_as_train_py': rate(numpy_step),
                     'note': 'plain = Trainer.train_d / train_g without next_g (two batch-B generator passes per iteration); numpy = host sequences, '
                             'pair selection, numpy feeds and the generated frames fetched to the host every step, as train() does (train.py:241-263)'}

    # ---- per-op timing of this process (instrumented pass, after the timed region): the HOT numbers -----------------
    roof, roof_dna, kernel_ms, insitu = None, None, {}, None
    # EVERY rank runs the instrumented pass: with world > 1 the programs contain gradient all-reduces, and a
    # collective issued by rank 0 alone would never complete.  Only rank 0 reports.
    if not args.trace_run:
        d_in0, g_in0, pair0, _ = pool[0]
        zero_state = torch.zeros(B, 5, device=device)
        recs = []
        # conv and DNA launches are idempotent: timed as launches inside a small captured HIP graph (graph.profile_ops) - each
        # op 10x back to back, i.e. with every operand cache-hot from the previous identical launch
        relaunch = lambda op: isinstance(op, (O._ConvBase, O.DnaOp, O.DnaBwdOp))   # noqa: E731
        fd_d, fd_g = tr._feed(*d_in0, zero_state), tr._feed(*g_in0)
        if lookahead:
            # the programs the timed step replays: plain D steps, then the D step that carries the look-ahead generator pass
            # (batch 2 B), then the G step that starts behind it
            fd_la = dict(fd_d)
            fd_la.update({tr.pair_img_ph: pair0[0], tr._pair_img_pad: pair0[0], tr.pair_action_ph: pair0[1]})
            heavy = sess.profile_ops([tr.d_opt_op, tr.clip_d, tr._pair_concat], fd_la, repeats=args.profile_repeats, relaunch=relaunch, skip=tr._skip_d)
            n_pairs_dd = n_critic // 2                       # (D1 -> D2), (D3 -> D4), ...: a pair-pass D step and a D step without generator
            recs += heavy * n_pairs_dd
            if n_pairs_dd:
                recs += sess.profile_ops([tr.d_opt_op, tr.clip_d] + tr._g_extra, fd_d, repeats=args.profile_repeats, relaunch=relaunch, skip=tr._skip_g) * n_pairs_dd
            if n_critic % 2:                                  # the last D step pairs with the G step
                recs += heavy
                recs += sess.profile_ops([tr.g_opt_op, tr.g_next_frame] + tr._g_extra, fd_g, repeats=args.profile_repeats, relaunch=relaunch, skip=tr._skip_g)
            else:
                recs += sess.profile_ops([tr.g_opt_op, tr.g_next_frame], fd_g, repeats=args.profile_repeats, relaunch=relaunch)
        else:
            recs += sess.profile_ops([tr.d_opt_op, tr.clip_d], fd_d, repeats=args.profile_repeats, relaunch=relaunch) * n_critic
            recs += sess.profile_ops([tr.g_opt_op, tr.g_next_frame], fd_g, repeats=args.profile_repeats, relaunch=relaunch)
        conv_ms = conv_fl = conv_by = 0.0
        n_conv = n_launch = n_dna = 0
        dna_ms = dna_bytes = dna_bytes_survey = 0.0
        for op, ms in recs:
            kind = type(op).__name__
            kernel_ms[kind] = kernel_ms.get(kind, 0.0) + ms
            if isinstance(op, O._ConvBase):
                conv_ms += ms
                conv_fl += conv_flops(op)
                conv_by += conv_bytes(op)
                n_conv += 1
                n_launch += 1
                if getattr(op, 'pair_active', False):      # this launch also ran the layer's weight gradient
                    conv_fl += conv_flops(op.pair_w)
                    conv_by += conv_bytes(op.pair_w)
                    n_conv += 1
            elif isinstance(op, O.WgradReduceOp):      # the deferred slab reductions of the weight gradients: conv time
                conv_ms += ms
            elif kind == 'DnaOp':      # k*k logits at their storage size + C image values in + C frame values out
                b, h, w, c = op.inputs[1].shape
                dna_ms += ms
                n_dna += 1
                dna_bytes += b * h * w * (op.ksize * op.ksize * (2.0 if args.dtype == 'bf16' else 4.0) + 2 * c * 4.0)
                dna_bytes_survey += b * h * w * (op.ksize * op.ksize + 2 * c) * (2.0 if args.dtype == 'bf16' else 4.0)      # (k*k + 6) * sizeof
                if op.second is not None:      # (both training programs read that tensor)
                    # this launch also writes the discriminator's input pixel (train.py:63-66: concat(frame, generated frame),
                    # 8 channels of the conv storage type) - the bytes of the concat launch it replaces
                    dna_bytes += b * h * w * op.second[1].shape[-1] * (2.0 if args.dtype == 'bf16' else 4.0)
        # ---- the committed evidence of this same command line: in-situ kernel durations (tools/insitu_times.py on a rocprofv3
        # kernel trace of `bench.py --trace-run`) and HBM-side bytes (tools/pmc_traffic.sh: --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md); neither can be taken from inside this
        # process, both are null / replaced by the hot numbers (and say so) when no file matches the workload
        tag = '%s_b%d_s%d_k%d%s' % (args.dtype, B, S, args.ksize, '' if dna else '_plain')
        std_step = adv and args.loss == 'bce' and args.opt == 'adam' and world == 1

        abi, sha = lib_fingerprint()
        stale = []

        def committed(kind):
            """The committed profile of THIS command line taken with THIS library (review r4 item 9): a profile whose recorded
            ABI version / library hash / look-ahead setting differs describes other kernels and is refused - the live numbers
            of this process are reported instead, and `timing` says why."""
            path = os.path.join(ROOT, 'profiles', 'r5', '%s_%s.json' % (kind, tag))
            if not (std_step and os.path.exists(path)):
                return None, None
            if sess.rt.bn_flags & 1 or getattr(sess, 'bn_two_launch_ops', 0):
                # the committed profiles were taken with the one-launch BatchNorm kernels; this run uses the two-launch ones (more
                # than one rank under side-stream collectives, or --bn-grid-exchange off): other kernels between the convs
                stale.append('profiles/r5/%s_%s.json describes a step with the one-launch BatchNorm kernels; this run takes the two-launch path' % (kind, tag))
                return None, None
            with open(path) as f:
                prof = json.load(f)
            if prof.get('abi_version') != abi or prof.get('lib_sha16') != sha or bool(prof.get('lookahead')) != bool(lookahead):
                stale.append('profiles/r5/%s_%s.json was taken with ABI %s / library %s / lookahead %s; loaded: ABI %d / %s / %s' % (
                    kind, tag, prof.get('abi_version'), prof.get('lib_sha16'), prof.get('lookahead'), abi, sha, bool(lookahead)))
                return None, None
            return prof, 'profiles/r5/%s_%s' % (kind, tag)
        insitu, insitu_src = committed('insitu')
        pj, pmc_src = committed('pmc_traffic')
        if conv_ms > 0:
            peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == 'bf16' else PEAK_F32_MFMA_TFLOPS
            hot = conv_fl / (conv_ms * 1e-3) / 1e12
            traffic, traffic_note = None, None
            c = (pj or {}).get('conv')
            if c:
                traffic = round(c['fetch_bytes_per_launch'] + c['write_bytes_per_launch'])     # bytes per launch
                traffic_note = ('HBM-side bytes per conv launch (FETCH_SIZE x2 + WRITE_SIZE) from %s.txt, separate --pmc passes of this '
                                'command; algorithmic (operands + result once) %d bytes per launch' % (pmc_src, round(conv_by / max(n_launch, 1))))
            t_ms, timing = conv_ms, 'hot relaunch, measured live in this process (%s)' % ('; '.join(stale) if stale else 'no in-situ profile of this workload is committed under profiles/r5')
            if insitu is not None:
                t_ms = insitu['family_us_per_step']['conv'] / 1e3
                timing = ('in situ: conv-family kernel time per real step from %s.json = TotalDurationNs / %d step executions of '
                          'the rocprofv3 kernel trace beside it' % (insitu_src, insitu['step_executions']))
            ach = conv_fl / (t_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'achieved': round(ach, 3), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_note': traffic_note, 'timing': timing,
                    'kernel': '%s (+splitk_reduce*, direct_*): %d conv/deconv fwd+dgrad+wgrad contractions per step' % (
                        'conv_mfma_bf16<*> / conv_pair_bf16<*> / conv_glds_bf16<*>' if args.dtype == 'bf16' else 'conv_mfma_f32<*> / conv_pair_f32<*>', n_conv),
                    'algorithmic_gflop_per_step': round(conv_fl / 1e9, 2), 'ms_per_step_in_kernel': round(t_ms, 4),
                    'avg_launch_us': round(t_ms * 1e3 / max(n_launch, 1), 2), 'launches_per_step': n_launch,
                    'algorithmic_bytes_per_launch': round(conv_by / max(n_launch, 1)),
                    'hot_relaunch': {'achieved': round(hot, 3), 'frac': round(hot / peak, 4), 'ms_per_step_in_kernel': round(conv_ms, 4),
                                     'note': 'this process: each op 10x back to back in a small HIP graph between two events (cache-hot operands)'}}
        if dna_ms > 0:
            pmc_dna = (pj or {}).get('dna_fwd')
            t_ms, timing = dna_ms, 'hot relaunch'
            if insitu is not None and 'dna_fwd' in insitu['kernel_us_per_launch']:
                t_ms = insitu['kernel_us_per_launch']['dna_fwd'] * n_dna / 1e3
                timing = 'in situ (%s.json: dna_fwd %.2f us per launch, %d launches per step)' % (insitu_src, insitu['kernel_us_per_launch']['dna_fwd'], n_dna)
            gbs = dna_bytes_survey / (t_ms * 1e-3) / 1e9
            ext = dna_bytes / (t_ms * 1e-3) / 1e9
            roof_dna = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(gbs / PEAK_HBM_GBS, 4),
                        'traffic': round(pmc_dna['fetch_bytes_per_launch'] + pmc_dna['write_bytes_per_launch']) if pmc_dna else None,
                        'kernel': 'dna_rows_kernel<K,fwd>' if args.ksize >= 6 else 'dna_kernel<K,4,fwd>', 'timing': timing,
                        # SURVEY 8(d)'s byte count: (k*k + 6) * sizeof per pixel (16.25 MB per launch at config 2)
                        'algorithmic_mb_per_step': round(dna_bytes_survey / 1e6, 2), 'ms_per_step_in_kernel': round(t_ms, 4),
                        # the same launches counting what the kernel really moves: float32 image / frame whatever the logits are,
                        # plus the discriminator-input pixel it also writes
                        'extended_definition': {'algorithmic_mb_per_step': round(dna_bytes / 1e6, 2), 'achieved': round(ext, 1),
                                                'frac': round(ext / PEAK_HBM_GBS, 4)},
                        'hot_relaunch': {'achieved': round(dna_bytes_survey / (dna_ms * 1e-3) / 1e9, 1),
                                         'frac': round(dna_bytes_survey / (dna_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                         'ms_per_step_in_kernel': round(dna_ms, 4)}}

    # The one-launch BatchNorm kernels flag a grid exchange that timed out (a block's peers were not resident: the launch finished
    # with wrong statistics - acgan_hip.h).  Looked at HERE, on every rank, before anything is reported: a set flag on any rank
    # means no result line and a non-zero exit code on all of them.
    _require_clean_exchange(sess, world)
    # Data parallel: every rank must hold the SAME weights after the timed steps (same start, same all-reduced gradients, same
    # update) - bit for bit, and finite.  Ranks that drifted apart mean the gradient exchange did not do what the step assumes
    # (a collective that silently did not run inside the captured graph, a bucket left out): no result line then either.
    in_sync = _require_weights_in_sync(sess, tr, world)
    if rank != 0:
        _leave_distributed(sess)
        return
    cpu = None
    if world == 1 and not args.no_cpu_baseline and not args.trace_run:
        cpu = cpu_baseline(args, n_critic)
    # which BASELINE.json configuration this run is: config 2 itself (the headline), or the per-GPU shard of configs 3 / 4
    # (global batch 256 / 128 over 8 / 4 GPUs = 32 per GPU) or config 5's geometry at 32 per GPU (its per-GPU batch is not stated)
    std = B == 32 and adv and dna
    if std and S == 64 and args.seq_len == 8 and args.ksize == 5 and args.loss == 'bce' and args.opt == 'adam' and args.dtype == 'f32':
        which = 'BASELINE config 2'
    elif std and S == 64 and args.seq_len == 8 and args.ksize == 5 and args.loss == 'bce' and args.opt == 'adam' and args.dtype == 'bf16':
        which = 'BASELINE config 3, per-GPU shard (global batch 256 over 8 GPUs)'
    elif std and S == 64 and args.seq_len == 8 and args.ksize == 5 and args.loss == 'wass' and args.opt == 'rmsprop':
        which = 'BASELINE config 4, per-GPU shard (global batch 128 over 4 GPUs; n_critic = 5, D weight clip)'
    elif std and S == 128 and args.seq_len == 16 and args.ksize == 11 and args.dtype == 'bf16' and args.loss == 'bce':
        which = 'BASELINE config 5 geometry (128x128x3 T=16, 11x11 DNA kernel, bf16), 32 per GPU'
    else:
        which = 'variant of BASELINE config 2'
    line = {
        'metric': 'GAN train steps/sec (G+D) on 64x64x3xT=8 push seq', 'value': round(world * args.steps / elapsed, 3),
        'unit': 'steps/s (batch-%d G+D steps, all ranks)' % B, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': '%s per GPU: batch=%d %dx%dx3 T=%d %s--loss %s %s(k=%d) --opt %s %s'
                               % (which, B, S, S, args.seq_len, '--adv ' if adv else '', args.loss, '--dna ' if dna else 'plain-G ',
                                  args.ksize, args.opt, 'fp32' if args.dtype == 'f32' else 'bf16-MFMA/fp32-accumulate'),
                   'global_batch': B * world, 'n_critic': n_critic, 'parallelism': 'dp%d' % world,
                   'hip_graphs': not args.no_graphs, 'timed_blocks': len(blocks), 'timed_seconds': round(float(np.sum(blocks)), 3),
                   'block_ms_per_step_min_max': [round(min(blocks) / args.steps * 1e3, 4), round(max(blocks) / args.steps * 1e3, 4)],
                   'dp_collectives': args.dp_collectives if (world > 1 or args.force_dp) else None, 'sequences_per_s': round(world * B * args.steps / elapsed, 1),
                   # what the first real N > 1 run can be checked against (review r4 item 7): the communicator's size as RCCL reports it,
                   # the all-reduce buckets per optimizer in bytes, and how often THIS rank called ncclAllReduce (eager first run +
                   # capture only: replays of a captured step call nothing from the host)
                   'data_parallel': dict(dp_report(sess, G.get_default_graph(), optim), weights_in_sync_on_all_ranks=in_sync) if (world > 1 or args.force_dp) else None,
                   'bn_grid_exchange': ('off' if sess.rt.bn_flags & 1 else ('two-launch kernels for the %d BatchNorm launches beside side-stream collectives, one-launch elsewhere' % sess.bn_two_launch_ops if getattr(sess, 'bn_two_launch_ops', 0) else 'on')), 'lookahead': bool(lookahead), 'abi_version': lib_fingerprint()[0], 'lib_sha16': lib_fingerprint()[1],
                   'opt': args.opt, 'trace_run': bool(args.trace_run),
                   # every training step this process executed (eager + capture + first replay, warm-up, all timed blocks)
                   'step_executions': 3 + args.warmup + args.steps * len(blocks)},
        'roofline': roof, 'roofline_dna': roof_dna, 'cpu_baseline': cpu, 'eval_rollout': rollout, 'api_rates': api_rates,
        # per kernel family, in situ (same source as `roofline.timing`); null without a committed profile of this workload
        'kernel_ms_per_step_insitu': ({k: round(v / 1e3, 4) for k, v in insitu['family_us_per_step'].items()} if insitu else None),
        # per op kind, this process: an event pair around every eager launch (includes the ~4 us the event records open per op)
        'op_ms_per_step': {k: round(v, 4) for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])},
    }
    result_out.write(json.dumps(line) + '\n')
    result_out.flush()
    _leave_distributed(sess)


def dp_report(sess, graph, optim):
    comm = sess.rt._comm
    buckets = {}
    for op in graph.ops:
        if isinstance(op, optim.AllReduceOp):
            buckets.setdefault(op.name.rsplit('/', 1)[0], []).append((op.end - op.start) * 4)
    return {'transport': type(comm).__name__ if comm is not None else None,
            'comm_world_size': getattr(comm, 'world_size', None), 'comm_rank': getattr(comm, 'rank', None),
            'rccl_reports_count_rank': (list(comm.reported()) if getattr(comm, 'reported', None) and comm.reported() else None),
            'nccl_allreduce_calls_this_rank': getattr(comm, 'calls', None),
            'bucket_bytes_per_optimizer': buckets,
            'sync_bn': bool(graph.collections['data_parallel'].sync_bn)}


def _require_clean_exchange(sess, world):
    from action_conditioned_gans_amd import _lib
    err = None
    try:
        sess.rt.check_exchange_flags()
    except _lib.AcgError as e:
        err = e
    bad = 0 if err is None else 1
    if world > 1:
        t = torch.tensor([bad], dtype=torch.int32)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        bad = int(t.item())
    if bad:
        sess.close(check=False)
        raise err if err is not None else _lib.AcgError('another rank reported a BatchNorm grid-exchange timeout: no result')


def _require_weights_in_sync(sess, tr, world):
    """-> True when all ranks hold bit-identical, finite trainable weights (checksums compared over the gloo control group);
    None on one rank; raises on every rank otherwise."""
    if world <= 1:
        return None
    from action_conditioned_gans_amd import _lib
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    mine = []
    for op in (tr.d_opt_op, tr.g_opt_op):
        flat = op.inputs[0].buf                 # the optimizer's flat float32 parameter buffer
        mine.append((int(flat.view(torch.int32).to(torch.int64).sum().item()), bool(torch.isfinite(flat).all().item())))
    seen = [None] * world
    torch.distributed.all_gather_object(seen, mine)
    if any(s != seen[0] for s in seen) or not all(ok for _, ok in seen[0]):
        sess.close(check=False)
        raise _lib.AcgError('data parallel: the ranks do not hold the same finite weights after the timed steps '
                            '(checksum, finite per optimizer and rank: %s): no result' % (seen,))
    return True


def _leave_distributed(sess):
    """Ordinary teardown: ncclCommDestroy of this process's communicator, then the gloo control group, then a normal
    interpreter exit."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    sess.close()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
