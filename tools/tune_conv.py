#!/usr/bin/env python
"""Sweep tile configuration x split-K for every distinct conv contraction of a Trainer graph on the GPU.

Uses the acg_debug_conv_plan tuning hook.  Output: one line per (layer, kind, cfg, splits) with the mean
launch time and the achieved algorithmic TFLOP/s, plus the planner's current choice.  Evidence for the
planner heuristic in csrc/conv_f32.hip (see profiles/).
  python tools/tune_conv.py [--batch 32] [--plain] > gpurun_out/tune_conv.txt
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from action_conditioned_gans_amd import _lib, graph as G, ops as O, optim, train as T   # noqa: E402

CFG = {2: '128x32', 3: '64x64'}   # 0 (128x128) and 1 (128x64) were retired with the v5 kernel


def time_graph(fn, reps=20, replays=3):
    """Mean time of one call with `reps` calls captured in a HIP graph (the host launch rate does not enter)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--plain', action='store_true')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--splits', default='1,2,3,4,6,8,12,16,32')
    ap.add_argument('--only', default='', help='substring filter on the op name (e.g. d/conv2/conv2d/wgrad)')
    ap.add_argument('--cfgs', default='-1,2,3', help='tile configurations to try (-1 = planner)')
    ap.add_argument('--unbatched-d', action='store_true')
    ap.add_argument('--top', type=int, default=6, help='rows printed per layer')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='bf16: tile configurations 1 (128x128) and 3 (64x64)')
    ap.add_argument('--img', type=int, default=64)
    ap.add_argument('--ksize', type=int, default=5)
    args = ap.parse_args()
    half = args.dtype == 'bf16'
    dt = _lib.ACG_BF16 if half else _lib.ACG_F32
    if half:
        CFG.clear()
        CFG.update({1: '128x128', 3: '64x64'})
    lib = _lib.load_tuning()
    dev = torch.device('cuda:0')
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = G.Session(device=dev, dtype=args.dtype)
    T.Trainer(sess, True, 'bce', 'adam', not args.plain, batch_size=args.batch, batched_d=not args.unbatched_d, img_size=args.img, ksize=args.ksize)
    seen = {}
    for op in G.get_default_graph().ops:
        if isinstance(op, O._ConvBase):
            kind = type(op).__name__ + ('T' if op.transposed else '')
            key = (op.which, op.desc.key(), isinstance(op, O.ConvWgradOp))
            seen.setdefault(key, (op, kind, []))[2].append(op.name)
    splits_list = [int(s) for s in args.splits.split(',')]
    print('# batch %d; columns: name kind which M-ish desc | cfg splits us TFLOP/s' % args.batch)
    cfgs = [int(c) for c in args.cfgs.split(',')]
    for (which, dkey, is_w), (op, kind, names) in seen.items():
        if args.only and not any(n == args.only for n in names):
            continue
        d = op.desc
        flops = 2.0 * d.batch * d.out_h * d.out_w * d.kh * d.kw * d.in_c * d.out_c
        r8 = lambda c: -(-c // 8) * 8     # noqa: E731
        nx = d.batch * d.in_h * d.in_w * (r8(d.in_c) if half else max(d.in_c, d.in_pitch))
        ny = d.batch * d.out_h * d.out_w * (r8(d.out_c) if half else max(d.out_c, d.out_pitch))
        nw = d.kh * d.kw * (max(d.in_c * r8(d.out_c), d.out_c * r8(d.in_c)) if half else d.in_c * d.out_c)
        tdt = torch.bfloat16 if half else torch.float32
        x = torch.randn(nx, device=dev).to(tdt)
        y = torch.randn(ny, device=dev).to(tdt)
        w = (torch.randn(nw, device=dev) * 0.05).to(tdt)       # bf16: stands in for either prepared filter copy
        if which == _lib.CONV_WGRAD:
            w = torch.zeros(d.kh * d.kw * d.in_c * d.out_c, device=dev)
        if which == _lib.CONV_FWD:
            a, b, out, fn = x, w, y, lib.conv2d_fwd
        elif which == _lib.CONV_DGRAD:
            a, b, out, fn = y, w, x, lib.conv2d_dgrad
        else:
            a, b, out, fn = x, y, w, lib.conv2d_wgrad
        results = []
        for cfg in cfgs:
            for sp in ([-1] if cfg == -1 else splits_list):
                lib.debug_conv_plan(cfg, sp)
                nbytes = lib.conv2d_workspace_bytes(ctypes.byref(d), which, dt)
                if nbytes > (2 << 30):
                    continue
                ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
                pa, pb, po, pw = (ctypes.c_void_p(t.data_ptr()) for t in (a, b, out, ws))

                def call():
                    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    if which == _lib.CONV_WGRAD:
                        fn(pa, pb, po, 0.0, ctypes.byref(d), dt, pw, nbytes, stream)
                    else:
                        fn(pa, pb, po, ctypes.byref(d), dt, pw, nbytes, stream)
                us = time_graph(call, reps=args.reps)
                results.append((us, cfg, sp))
        lib.debug_conv_plan(-1, -1)
        # the planner's own choice again, LAST (the first measurement of a layer also pays its first touches), and its split count
        nbytes = lib.conv2d_workspace_bytes(ctypes.byref(d), which, dt)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        pa, pb, po, pw = (ctypes.c_void_p(t.data_ptr()) for t in (a, b, out, ws))
        if -1 in cfgs:
            results.append((time_graph(call, reps=args.reps), -1, -lib.conv2d_splits(ctypes.byref(d), which, dt)))
        auto = min([r for r in results if r[1] == -1] or [min(results)])
        best = min(results)
        print('%-28s %-14s x%d  flops %.2fG  auto %.1fus (%.1f TF)  best %s s=%d %.1fus (%.1f TF)' % (
            names[0], kind, len(names), flops / 1e9, auto[0], flops / auto[0] / 1e6, CFG.get(best[1], 'auto'), best[2], best[0],
            flops / best[0] / 1e6))
        for us, cfg, sp in sorted(results)[:args.top]:
            print('      %-8s s=%-3d %8.1f us  %6.1f TF' % (CFG.get(cfg, 'auto'), sp, us, flops / us / 1e6))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
