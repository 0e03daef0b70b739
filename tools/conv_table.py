#!/usr/bin/env python
"""Per-conv-op table of one G+D step on the GPU: time (20 launches captured in a HIP graph and replayed: the in-graph cost of a launch), algorithmic
FLOPs, TFLOP/s and the time above a 90 TFLOP/s line - where the conv time of a step goes.
  python tools/conv_table.py [--batch 32] > gpurun_out/conv_table.txt"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from action_conditioned_gans_amd import graph as G, ops as O, optim, train as T   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--other', action='store_true', help='table of the NON-conv ops instead (time, tensor bytes moved, GB/s)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--img', type=int, default=64)
    ap.add_argument('--ksize', type=int, default=5)
    ap.add_argument('--lib', default=None, help='an alternative build of the library (kernel A/B experiments)')
    ap.add_argument('--no-lookahead', action='store_true', help='the plain call path (two batch-B generator passes per iteration) instead of the programs bench.py replays')
    args = ap.parse_args()
    if args.lib:
        from action_conditioned_gans_amd import _lib
        _lib._LIB = _lib.Library(args.lib)
    B, S = args.batch, args.img
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = G.Session(device='cuda:0', dtype=args.dtype)
    tr = T.Trainer(sess, True, 'bce', 'adam', True, batch_size=B, img_size=S, ksize=args.ksize)
    sess.run(G.global_variables_initializer())
    rng = np.random.default_rng(0)
    x, y = (rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32) for _ in range(2))
    a, s = rng.standard_normal((B, 10)).astype(np.float32), rng.standard_normal((B, 5)).astype(np.float32)
    import ctypes
    rows = []
    fd_d, fd_g = tr._feed(x, y, a, np.zeros((B, 5), np.float32)), tr._feed(x, y, a, s)
    progs = (('D', [tr.d_opt_op, tr.clip_d], fd_d, None), ('G', [tr.g_opt_op, tr.g_next_frame], fd_g, None))
    if tr.lookahead and not args.no_lookahead:
        # the programs of the bench's step: the D step that carries the generator pass for both sub-steps (batch 2 B), the G
        # step that starts behind it
        pair_x, pair_a = np.concatenate([x, x]), np.concatenate([a, a])
        fd_la = dict(fd_d)
        fd_la.update({tr.pair_img_ph: pair_x, tr._pair_img_pad: pair_x, tr.pair_action_ph: pair_a})
        progs = (('D', [tr.d_opt_op, tr.clip_d, tr._pair_concat], fd_la, tr._skip_d), ('G', [tr.g_opt_op, tr.g_next_frame] + tr._g_extra, fd_g, tr._skip_g))
    for tag, fetch, feed, skip in progs:
        sess.run(fetch, feed, skip=skip)
        key = [k for k in sess._programs][-1]
        prog = sess._programs[key]
        for kind, seg in prog.segments:
            if kind == 'host':
                continue
            for op, fn in seg:
                if isinstance(op, O._ConvBase) == args.other:
                    continue
                # 20 launches of the op captured in a HIP graph: the in-graph cost of one launch (kernel + reduce + gaps)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    for _ in range(20):
                        fn(sp)
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                if args.other:    # bytes = every input and output tensor once (the algorithmic traffic of an elementwise/reduction op)
                    by = float(sum(t.numel * (2 if t.dtype == torch.bfloat16 else 4) for t in list(op.inputs) + list(op.outputs)))
                    shape = 'x'.join(str(v) for v in (op.inputs[0].shape if op.inputs else ()))
                    rows.append((tag, op.name, type(op).__name__ + ' ' + shape, by, e0.elapsed_time(e1) * 1e3 / 60))
                    continue
                d = op.desc
                fl = 2.0 * d.batch * d.out_h * d.out_w * d.kh * d.kw * d.in_c * d.out_c
                wfl = fl                       # (the paired weight gradient is not limited)
                if type(op).__name__ == 'ConvDgradOp':      # an input gradient limited to the feature channels (dgrad_c / adj_dgrad_c)
                    if op.transposed and d.adj_dgrad_c > 0:
                        fl *= d.adj_dgrad_c / d.out_c
                    elif not op.transposed and d.dgrad_c > 0:
                        fl *= d.dgrad_c / d.in_c
                paired = getattr(op, 'pair_active', False)     # this launch also ran the layer's weight gradient (same FLOPs again)
                splits = sess.rt.lib.conv2d_splits(ctypes.byref(d), op.which, sess.rt.conv_dtype)
                rows.append((tag, op.name + ('+wgrad' if paired else ''), type(op).__name__ + ('+W' if paired else '') + ' s%d' % splits, fl + (wfl if paired else 0.0),
                             e0.elapsed_time(e1) * 1e3 / 60))
    tot_us = sum(r[4] for r in rows)
    if args.other:
        print('# %d non-conv ops, %.1f us, %.1f MB in+out tensors' % (len(rows), tot_us, sum(r[3] for r in rows) / 1e6))
        print('# step op kind+input-shape MB us GB/s')
        for tag, name, kind, by, us in sorted(rows, key=lambda r: -r[4]):
            print('%s %-40s %-34s %7.2f %7.1f %7.0f' % (tag, name[:40], kind, by / 1e6, us, by / us / 1e3))
        return
    tot_fl = sum(r[3] for r in rows)
    print('# %d conv launches (+W: input gradient and weight gradient of a layer in one launch), %.1f us, %.2f GFLOP, %.1f TFLOP/s average' % (len(rows), tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6))
    line = 90e6 if args.dtype == 'f32' else 750e6      # 90 TFLOP/s (fp32) / 750 TFLOP/s (bf16: 30 % of the dense peak)
    print('# step op kind+splits GFLOP us TFLOP/s us_above_line')
    for tag, name, kind, fl, us in sorted(rows, key=lambda r: -(r[4] - r[3] / line)):
        print('%s %-44s %-18s %6.2f %7.1f %6.1f %7.1f' % (tag, name[:44], kind, fl / 1e9, us, fl / us / 1e6, us - fl / line))


if __name__ == '__main__':
    main()
