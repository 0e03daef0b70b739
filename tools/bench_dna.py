#!/usr/bin/env python
"""DNA stencil micro-benchmark: back-to-back launches of acg_dna_fwd / acg_dna_bwd through the C ABI,
timed with events on the launch stream; reports algorithmic GB/s ((k*k+2C)*4 B/pixel
forward, (2*k*k+2C)*4 backward) against the 8 TB/s HBM peak.  The batch sweep separates the launch floor
(config-2 size is 16 MB per launch) from the streaming rate."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from action_conditioned_gans_amd import _lib   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='32,128,512,2048')
    ap.add_argument('--img', type=int, default=64)
    ap.add_argument('--ksize', type=int, default=5)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--cdna', action='store_true', help='time the CDNA transformation (10 masks, k=5) instead')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help='storage of logits / dlogits')
    ap.add_argument('--lib', default=None, help='an alternative build of the library (kernel experiments)')
    args = ap.parse_args()
    lib, dev = (_lib.Library(args.lib) if args.lib else _lib.get()), torch.device('cuda:0')
    k, S, C = args.ksize, args.img, 3
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    if args.cdna:
        M = 10
        for B in [int(b) for b in args.batches.split(',')]:
            par = torch.randn(B, k * k * M, device=dev)
            img = torch.rand(B, S, S, C, device=dev) * 2 - 1
            out = torch.empty(M, B, S, S, C, device=dev)
            kn = torch.empty(B, k * k * M, device=dev)
            dout = torch.randn_like(out)
            dpar, dimg = torch.empty_like(par), torch.empty_like(img)
            nb = lib.cdna_workspace_bytes(B, S, S, C, M, k)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            img_b = B * S * S * C * 4
            for name, fn, nbytes in (
                    ('fwd', lambda: lib.cdna_fwd(p(par), p(img), p(out), p(kn), B, S, S, C, M, k, 1e-12, 0, stream), (1 + M) * img_b),
                    ('bwd', lambda: lib.cdna_bwd(p(par), p(kn), p(img), p(dout), p(dpar), p(dimg), B, S, S, C, M, k, 1e-12, 0, p(ws), nb, stream), (2 + M) * img_b)):
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.reps
                print('cdna_%s k=%d masks=%d B=%-5d %7.2f MB  %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)' % (
                    name, k, M, B, nbytes / 1e6, us, nbytes / us / 1e3, 100 * nbytes / us / 1e3 / 8000))
        return
    half = args.dtype == 'bf16'
    dt, es, lp = (1, 2, (k * k + 7) // 8 * 8) if half else (0, 4, k * k)
    for B in [int(b) for b in args.batches.split(',')]:
        logits = torch.randn(B, S, S, lp, device=dev).to(torch.bfloat16 if half else torch.float32)
        bias = torch.randn(k * k, device=dev)
        dbias = torch.zeros(k * k, device=dev)
        img = torch.rand(B, S, S, C, device=dev) * 2 - 1
        out = torch.empty_like(img)
        dout = torch.randn_like(img)
        dl = torch.zeros_like(logits)
        nws = lib.dna_workspace_bytes(B, S, S, k)
        ws = torch.zeros(max(nws, 16), dtype=torch.uint8, device=dev)
        # algorithmic bytes: k*k logits (+ k*k dlogits) at their storage size, 2C float32 image / frame values
        for name, fn, nbytes in (
                ('fwd', lambda: lib.dna_fwd(p(logits), p(bias), p(img), p(out), None, 0, 0, 0, B, S, S, C, k, dt, stream), B * S * S * (k * k * es + 2 * C * 4)),
                ('bwd', lambda: lib.dna_bwd(p(logits), p(bias), p(img), p(dout), None, 0, 0, 0, p(dl), p(dbias), 0.0, B, S, S, C, k, dt, p(ws), nws, stream),
                 B * S * S * (2 * k * k * es + 2 * C * 4))):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.reps
            print('dna_%s k=%d B=%-5d %7.2f MB  %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)' % (
                name, k, B, nbytes / 1e6, us, nbytes / us / 1e3, 100 * nbytes / us / 1e3 / 8000))
        del logits, dl


if __name__ == '__main__':
    main()
