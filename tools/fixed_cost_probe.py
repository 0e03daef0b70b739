#!/usr/bin/env python
"""Fixed cost of a conv launch: forward / input gradient / weight gradient of 1x1, 3x3 and 5x5 stride-1 layers with few K-steps
on a 32768 x 128 output (32 x 32 x 32 pixels, 128 channels), 20 launches in a HIP graph; the intercept at one K-step is what a
launch costs before it computes anything.   python tools/fixed_cost_probe.py [f32|bf16] [alternative library]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from action_conditioned_gans_amd import _lib as L
from abi_call import Abi, _p
from conv16_probe import time_graph

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
lib = L.Library(sys.argv[2]) if len(sys.argv) > 2 else L.get()
half = dtype == 'bf16'
code = L.ACG_BF16 if half else L.ACG_F32
abi = Abi(lib, 'cuda:0', conv_dtype=code)
B, S, N = 32, int(os.environ.get("PROBE_S", "32")), 128
for which in ('fwd', 'dgrad', 'wgrad'):
    for k, cin in ((1, 8), (1, 64), (1, 256), (3, 64), (5, 64)):
        d = abi.desc(B, S, S, cin, k, k, N, 1, 'SAME')
        x = torch.randn(B, S, S, cin, device='cuda'); w = torch.randn(k, k, cin, N, device='cuda') * 0.05
        dy = torch.randn(B, S, S, N, device='cuda')
        y = torch.zeros(B, S, S, N, device='cuda'); dx = torch.zeros_like(x); dw = torch.zeros_like(w)
        if half:
            x, dy, y, dx = abi.to16(x), abi.to16(dy), abi.to16(y), abi.to16(dx)
            rm, tr = abi.prep_weights(w)
        wf, wd = (tr, rm) if half else (w, w)
        cw = {'fwd': L.CONV_FWD, 'dgrad': L.CONV_DGRAD, 'wgrad': L.CONV_WGRAD}[which]
        ws, n = abi.ws(lib.conv2d_workspace_bytes(ctypes.byref(d), cw, code))
        sp = lib.conv2d_splits(ctypes.byref(d), cw, code)
        if which == 'fwd':
            fn = lambda: lib.conv2d_fwd(_p(x), _p(wf), _p(y), ctypes.byref(d), code, _p(ws), n, abi.stream())
        elif which == 'dgrad':
            fn = lambda: lib.conv2d_dgrad(_p(dy), _p(wd), _p(dx), ctypes.byref(d), code, _p(ws), n, abi.stream())
        else:
            fn = lambda: lib.conv2d_wgrad(_p(x), _p(dy), _p(dw), 0.0, ctypes.byref(d), code, _p(ws), n, abi.stream())
        us = time_graph(fn); fl = 2.0 * B * S * S * k * k * cin * N
        print('%s %-5s %dx%d Cin=%3d <-> 128: splits %3d  %6.1f us  %6.1f TF/s' % (dtype, which, k, k, cin, sp, us, fl / us / 1e6))
