#!/usr/bin/env python
"""Separate the fixed cost of a conv launch from its per-K-step cost (evidence for kernel work, profiles/).

Synthetic 5x5/s2 FWD convs with M, N fixed and Cin swept, splits forced to 1: the grid stays the same while the
number of 32-deep K-steps grows, so a line fit gives  time = fixed + nk * per_step.  Launches are captured in a HIP
graph (100 per replay) so the host launch rate does not enter.  One MFMA-bound K-step of a 64x64 tile costs a wave
16 x v_mfma_f32_32x32x2 = 1024 cycles = 0.43 us at 2.4 GHz; that is the floor `per_step` is compared with.
  python tools/conv_kloop.py > gpurun_out/conv_kloop.txt
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from action_conditioned_gans_amd import _lib   # noqa: E402


def time_graph(fn, reps=100, replays=5):
    for _ in range(3):
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
    ap.add_argument('--dtype', type=int, default=0)
    ap.add_argument('--cfg', type=int, default=3)
    ap.add_argument('--modes', default='fwd,dgrad,wgrad')
    ap.add_argument('--data', default='randn', help='randn | zeros | ones | small (randn * 1e-3): operand values (the MFMA rate turned out to depend on them)')
    args = ap.parse_args()
    lib = _lib.load_tuning()
    dev = torch.device('cuda:0')
    print('# cfg %d dtype %d; rows: blocks  [Cin -> nk: us]  fit fixed us + per K-step us (floor 0.43 us fp32 64x64)' % (args.cfg, args.dtype))
    for which, wname in ((_lib.CONV_FWD, 'fwd'), (_lib.CONV_DGRAD, 'dgrad'), (_lib.CONV_WGRAD, 'wgrad')):
        if wname not in args.modes.split(','):
            continue
        for batch, hw, cout in ((8, 8, 256), (32, 8, 256), (64, 8, 256), (128, 8, 256), (32, 32, 64)):
            pts = []
            for cin in (32, 64, 128, 256):
                d = _lib.ConvDesc()
                lib.conv_desc_init(ctypes.byref(d), batch, hw, hw, cin, 5, 5, cout, 2, 1)
                nx, ny, nw = batch * hw * hw * cin, batch * d.out_h * d.out_w * cout, 25 * cin * cout
                x, y, w = (torch.randn(n, device=dev) for n in (nx, ny, nw))
                if args.data != 'randn':
                    for t_ in (x, y, w):
                        if args.data == 'zeros':
                            t_.zero_()
                        elif args.data == 'ones':
                            t_.fill_(1.0)
                        else:
                            t_.mul_(1e-3)
                lib.debug_conv_plan(args.cfg, 1)
                stream_of = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)   # noqa: E731
                px, py, pw = (ctypes.c_void_p(t.data_ptr()) for t in (x, y, w))
                if which == _lib.CONV_FWD:
                    fn = lambda: lib.conv2d_fwd(px, pw, py, ctypes.byref(d), args.dtype, None, 0, stream_of())   # noqa: E731
                    nk = 25 * cin // 32
                    blocks = (batch * d.out_h * d.out_w + 63) // 64 * ((cout + 63) // 64)
                elif which == _lib.CONV_DGRAD:
                    fn = lambda: lib.conv2d_dgrad(py, pw, px, ctypes.byref(d), args.dtype, None, 0, stream_of())   # noqa: E731
                    nk = 9 * cout // 32      # class 0 (3x3 taps); here cout is fixed, cin is the N extent
                    blocks = 4 * ((batch * (hw // 2) ** 2 + 63) // 64) * ((cin + 63) // 64)
                else:
                    fn = lambda: lib.conv2d_wgrad(px, py, pw, 0.0, ctypes.byref(d), args.dtype, None, 0, stream_of())   # noqa: E731
                    nk = batch * d.out_h * d.out_w // 32
                    blocks = (25 * cin + 63) // 64 * ((cout + 63) // 64)
                us = time_graph(fn)
                pts.append((cin, nk, blocks, us))
            lib.debug_conv_plan(-1, -1)
            desc = '  '.join('%d->nk %d, %d blk: %.1f' % p for p in pts)
            if which == _lib.CONV_FWD:
                a, b = np.polyfit([p[1] for p in pts], [p[3] for p in pts], 1)
                fit = 'fixed %.1f us + %.3f us/K-step' % (b, a)
            else:
                fit = ''
            print('%-5s B=%-3d %dx%d Cout=%d | %s | %s' % (wname, batch, hw, hw, cout, desc, fit))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
