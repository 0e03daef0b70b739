#!/usr/bin/env python
"""BatchNorm micro-benchmark: acg_bn_act_fwd / acg_bn_act_fwd_partials / acg_bn_act_bwd through the C ABI at the tensor sizes of
BASELINE configs 2, 3 and 5, timed the way the step runs them (N launches captured into a HIP graph, replayed between two
events).  `--rot K` rotates over K sets of tensors, so that with K sets larger than the 256 MB memory-side cache every launch
finds its operands in HBM (the step itself finds them where the producing conv left them: mostly cache-warm).
  python tools/bench_bn.py [--dtype bf16] [--set c2|c5] [--rot 1] [--lib alt.so] [--check]
GB/s = algorithmic bytes (fwd: x in + y out; fwd without epilogue statistics: 2 x in + y out; bwd: x, dy in + dx out)."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from action_conditioned_gans_amd import _lib   # noqa: E402

SETS = {
    # (rows per group, channels, groups, activation, label)
    'c2': [(32768, 128, 1, 'relu', 'g/tconv3'), (32768, 64, 2, 'lrelu', 'd/conv1 D'), (32768, 64, 1, 'lrelu', 'd/conv1 G'),
           (8192, 128, 2, 'lrelu', 'd/conv2 D'), (32768, 32, 1, 'relu', 'g/conv1'), (8192, 128, 1, 'relu', 'g/tconv2'),
           (8192, 64, 1, 'relu', 'g/conv2'), (4096, 128, 1, 'lrelu', 'd/conv3 D/2'), (2048, 128, 1, 'relu', 'g/conv3')],
    'small': [(2048, 128, 1, 'relu', 'g/conv3'), (2048, 128, 2, 'lrelu', 'd/conv3 D'), (2048, 32, 1, 'relu', 'g/sconv3'), (1024, 128, 1, 'relu', '1024 x 128'),
              (512, 256, 1, 'relu', 'g/conv4'), (512, 256, 2, 'lrelu', 'd/conv4 D'), (128, 512, 2, 'lrelu', 'd/conv5 D')],
    'c5': [(131072, 128, 1, 'relu', 'g/tconv3'), (131072, 64, 2, 'lrelu', 'd/conv1 D'), (131072, 64, 1, 'lrelu', 'd/conv1 G'),
           (32768, 128, 2, 'lrelu', 'd/conv2 D'), (131072, 32, 1, 'relu', 'g/conv1'), (32768, 128, 1, 'relu', 'g/tconv2'),
           (32768, 64, 1, 'relu', 'g/conv2'), (8192, 128, 2, 'lrelu', 'd/conv3 D')],
}
ACT = {'relu': _lib.ACT_RELU, 'lrelu': _lib.ACT_LRELU, None: _lib.ACT_NONE}


def timed(fns, n_graph=20, reps=5):
    """fns: list of callables taking the stream pointer, one per rotation set; returns microseconds per launch."""
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(n_graph):
            fns[i % len(fns)](sp)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_graph)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--set', default='c2', choices=sorted(SETS))
    ap.add_argument('--rot', type=int, default=1)
    ap.add_argument('--lib', default=None)
    ap.add_argument('--tile', type=int, default=64, help='rows per conv tile of the synthetic epilogue statistics')
    ap.add_argument('--check', action='store_true', help='compare every result with a float64 torch restatement')
    args = ap.parse_args()
    lib, dev = (_lib.Library(args.lib) if args.lib else _lib.get()), torch.device('cuda:0')
    half = args.dtype == 'bf16'
    tdt, es, code = (torch.bfloat16, 2, _lib.ACG_BF16) if half else (torch.float32, 4, _lib.ACG_F32)
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    print('# %s, set %s, rot %d, tile %d; us per launch and GB/s by algorithmic bytes' % (args.dtype, args.set, args.rot, args.tile))
    print('# %-14s %8s %5s %2s | %-18s | %-18s | %-18s' % ('layer', 'rows', 'C', 'g', 'fwd (own stats)', 'fwd (epilogue stats)', 'bwd'))
    tot = [0.0, 0.0, 0.0]
    for R, C, G, act, label in SETS[args.set]:
        rows = R * G
        nb = lib.bn_workspace_bytes(rows, C, G)
        sets = []
        for k in range(args.rot):
            g = torch.Generator(device=dev).manual_seed(k)
            x = (torch.randn(rows, C, device=dev, generator=g) * 1.5 + 0.3).to(tdt)
            dy = torch.randn(rows, C, device=dev, generator=g).to(tdt)
            beta = torch.randn(C, device=dev, generator=g) * 0.1
            y, dx = torch.empty_like(x), torch.empty_like(x)
            mean, rstd, dbeta = (torch.empty(G * C, device=dev) for _ in range(3))
            ws = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)       # forward's; backward has its own (wsb): the one-launch
            wsb = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)      # kernels keep state in a workspace that must be theirs alone
            nblk = R // args.tile
            xt = x.float().view(G, nblk, args.tile, C)
            s = xt.sum(2)
            part = torch.stack([s, ((xt - (s / args.tile).unsqueeze(2)) ** 2).sum(2)], dim=2).contiguous()     # [G][nblk][2][C]
            sets.append((x, dy, beta, y, dx, mean, rstd, dbeta, ws, part, nblk, wsb))
        a = ACT[act]

        def f_fwd(t):
            x, dy, beta, y, dx, mean, rstd, dbeta, ws, part, nblk, wsb = t
            return lambda s: lib.bn_act_fwd(p(x), p(beta), p(y), p(mean), p(rstd), rows, C, 0, 0, G, 1e-3, a, 0.2, code, 0, p(ws), nb, s)

        def f_fwdp(t):
            x, dy, beta, y, dx, mean, rstd, dbeta, ws, part, nblk, wsb = t
            return lambda s: lib.bn_act_fwd_partials(p(x), p(beta), p(part), nblk, args.tile, R, p(y), p(mean), p(rstd), rows, C, 0, 0, G, 1e-3, a, 0.2, code, s)

        def f_bwd(t):
            x, dy, beta, y, dx, mean, rstd, dbeta, ws, part, nblk, wsb = t
            return lambda s: lib.bn_act_bwd(p(x), p(dy), p(beta), p(mean), p(rstd), p(dx), p(dbeta), 0.0, rows, C, 0, 0, G, a, 0.2, code, 0, p(wsb), nb, s)
        us = [timed([f(t) for t in sets]) for f in (f_fwd, f_fwdp, f_bwd)]
        by = [3.0 * rows * C * es, 2.0 * rows * C * es, 3.0 * rows * C * es]
        print('%-16s %8d %5d %2d | %7.2f us %7.0f | %7.2f us %7.0f | %7.2f us %7.0f' % (
            label, R, C, G, us[0], by[0] / us[0] / 1e3, us[1], by[1] / us[1] / 1e3, us[2], by[2] / us[2] / 1e3))
        for i in range(3):
            tot[i] += us[i]
        if args.check:
            x, dy, beta, y, dx, mean, rstd, dbeta, ws, part, nblk, wsb = sets[0]
            sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            x64 = x.double().view(G, R, C)
            mu, var = x64.mean(1, keepdim=True), x64.var(1, unbiased=False, keepdim=True)
            xh = (x64 - mu) / torch.sqrt(var + 1e-3)
            u = xh + beta.double()
            want = torch.relu(u) if act == 'relu' else 0.6 * u + 0.4 * u.abs()
            for name, f in (('fwd', f_fwd), ('fwd_partials', f_fwdp)):
                y.zero_()
                f(sets[0])(sp)
                torch.cuda.synchronize()
                err = (y.double().view(G, R, C) - want).abs().max().item()
                print('    check %-12s max |err| %.3e (tolerance %s)' % (name, err, '2e-2 bf16 rounding' if half else '2e-5'))
            f_bwd(sets[0])(sp)
            torch.cuda.synchronize()
            d = (torch.where(u > 0, 1.0, 0.0) if act == 'relu' else 0.6 + 0.4 * torch.sign(u)) * dy.double().view(G, R, C)
            want_dx = (d - d.mean(1, keepdim=True) - xh * (d * xh).mean(1, keepdim=True)) / torch.sqrt(var + 1e-3)
            print('    check bwd          max |err| %.3e, dbeta %.3e' % ((dx.double().view(G, R, C) - want_dx).abs().max().item(),
                                                                       (dbeta.double()[:C] - d.sum((0, 1))).abs().max().item()))
    print('# total %8.1f us | %8.1f us | %8.1f us' % tuple(tot))


if __name__ == '__main__':
    main()
