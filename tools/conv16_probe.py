#!/usr/bin/env python
"""bf16 conv kernel probe: per-K-step cost and fixed (prologue + epilogue + launch) cost of one contraction.
Times conv2d fwd / dgrad / wgrad at fixed M, N and growing K (input channels), 20 launches in a HIP graph.
  python tools/conv16_probe.py [--batch 32] [--hw 64] [--cout 128]"""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from action_conditioned_gans_amd import _lib as L
from abi_call import Abi, _p


def time_graph(fn, reps=20, replays=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--hw', type=int, default=64)
    ap.add_argument('--cout', type=int, default=128)
    ap.add_argument('--cins', default='64,128,256')
    ap.add_argument('--which', default='fwd,dgrad,wgrad')
    ap.add_argument('--lib', default=None, help='an alternative build of the library (kernel A/B experiments)')
    args = ap.parse_args()
    lib = L.Library(args.lib) if args.lib else L.get()
    abi = Abi(lib, 'cuda:0', conv_dtype=L.ACG_BF16)
    B, S, N = args.batch, args.hw, args.cout
    for which in args.which.split(','):
        pts = []
        for cin in [int(c) for c in args.cins.split(',')]:
            d = abi.desc(B, S, S, cin, 5, 5, N, 2, 'SAME')
            x = torch.randn(B, S, S, cin, device='cuda').bfloat16()
            w = torch.randn(5, 5, cin, N, device='cuda') * 0.05
            rm, tr = abi.prep_weights(w)
            y = torch.zeros(B, d.out_h, d.out_w, N, dtype=torch.bfloat16, device='cuda')
            dy = torch.randn(B, d.out_h, d.out_w, N, device='cuda').bfloat16()
            dx = torch.zeros_like(x)
            dw = torch.zeros(5, 5, cin, N, device='cuda')
            code = {'fwd': L.CONV_FWD, 'dgrad': L.CONV_DGRAD, 'wgrad': L.CONV_WGRAD}[which]
            ws, n = abi.ws(lib.conv2d_workspace_bytes(ctypes.byref(d), code, L.ACG_BF16))
            splits = lib.conv2d_splits(ctypes.byref(d), code, L.ACG_BF16)
            dref = ctypes.byref(d)
            if which == 'fwd':
                fn = lambda: lib.conv2d_fwd(_p(x), _p(tr), _p(y), dref, L.ACG_BF16, _p(ws), n, abi.stream())
            elif which == 'dgrad':
                fn = lambda: lib.conv2d_dgrad(_p(dy), _p(rm), _p(dx), dref, L.ACG_BF16, _p(ws), n, abi.stream())
            else:
                fn = lambda: lib.conv2d_wgrad(_p(x), _p(dy), _p(dw), 0.0, dref, L.ACG_BF16, _p(ws), n, abi.stream())
            us = time_graph(fn)
            fl = 2.0 * B * d.out_h * d.out_w * 25 * cin * N
            pts.append((cin, us, fl / us / 1e6, splits))
            print('%-6s B=%d %dx%d Cin=%4d Cout=%d splits=%d  %8.1f us  %7.1f TFLOP/s' % (which, B, S, S, cin, N, splits, us, fl / us / 1e6), flush=True)
        if len(pts) >= 2:
            (c0, u0, _, _), (c1, u1, _, _) = pts[0], pts[-1]
            print('   slope %.3f us per 64 input channels (= 25 K-steps of the FWD contraction), intercept %.1f us' % ((u1 - u0) / ((c1 - c0) / 64.0), u0 - (u1 - u0) / (c1 - c0) * c0))


if __name__ == '__main__':
    main()
