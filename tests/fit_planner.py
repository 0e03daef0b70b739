#!/usr/bin/env python
"""Score split-K planner rules against a tools/tune_conv.py sweep (the CPU side of the planner tuning).

Rebuilds the Trainer graph on the CPU (C oracle as the stand-in library) to get each conv op's GEMM extents, reads the
per-(layer, config, splits) times of a sweep file and prints, per rule, the summed time of all distinct contractions
(weighted by how many ops share them) next to the per-layer optimum.
  python tests/fit_planner.py gpurun_out/tune.txt
"""
import re
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from action_conditioned_gans_amd import _lib, graph as G, ops as O, optim, train as T   # noqa: E402
from oracle import cbind   # noqa: E402  (test-side tool: graph geometry only; lives under tests/ because it uses the oracle)


def geometry(batch=32):
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = G.Session(device='cpu', lib=cbind.load())
    T.Trainer(sess, True, 'bce', 'adam', True, batch_size=batch)
    geo = {}
    for op in G.get_default_graph().ops:
        if not isinstance(op, O._ConvBase):
            continue
        d, which = op.desc, op.which
        cin_p, cout_p = (d.in_c + 3) & ~3, (d.out_c + 3) & ~3
        if which == _lib.CONV_FWD:
            M, N, K, classes = d.batch * d.out_h * d.out_w, d.out_c, d.kh * d.kw * cin_p, 1
        elif which == _lib.CONV_DGRAD:
            hc, wc = -(-d.in_h // d.stride_h), -(-d.in_w // d.stride_w)
            M, N = d.batch * hc * wc, d.in_c
            K = -(-d.kh // d.stride_h) * -(-d.kw // d.stride_w) * cout_p
            classes = d.stride_h * d.stride_w
        else:
            M, N, K, classes = d.kh * d.kw * cin_p, d.out_c, d.batch * d.out_h * d.out_w, 1
        geo.setdefault(op.name, dict(M=M, N=N, nk=-(-K // 32), classes=classes, which=which))
    return geo


def read_sweep(path):
    layers, cur = [], None
    for line in open(path):
        m = re.match(r'(\S+)\s+(\S+)\s+x(\d+)\s+flops ([\d.]+)G\s+auto ([\d.]+)us', line)
        if m:
            cur = dict(name=m.group(1), mult=int(m.group(3)), auto=float(m.group(5)), res={})
            layers.append(cur)
            continue
        m = re.match(r'\s+(\S+)\s+s=(-?\d+)\s+([\d.]+) us', line)
        if m and cur is not None:
            cur['res'].setdefault(m.group(1), {})[int(m.group(2))] = float(m.group(3))
    return layers


def lookup(res, s):
    """time at the sampled split count nearest to s (log scale)"""
    import math
    ks = sorted(res)
    k = min(ks, key=lambda v: abs(math.log(v) - math.log(max(s, 1))))
    return res[k], k


def main():
    layers = read_sweep(sys.argv[1])
    geo = geometry()
    rows = []
    for L in layers:
        g = geo.get(L['name'])
        if g is None:
            print('no geometry for', L['name'])
            continue
        rows.append((L, g))
    best = sum(min(min(r.values()) for r in L['res'].values()) * L['mult'] for L, g in rows)
    auto = sum(L['auto'] * L['mult'] for L, g in rows)
    print('layers %d  sum(auto) %.1f us  sum(per-layer best) %.1f us' % (len(rows), auto, best))

    def tiles_of(g, bm, bn):
        return -(-g['M'] // bm) * -(-g['N'] // bn) * g['classes']

    results = []
    for target in (192, 256, 320, 384, 448, 512, 640, 768, 1024):
        for min_steps in (2, 3, 4, 6, 8):
            for smax in (64, 128):
                tot = 0.0
                for L, g in rows:
                    cfg = '128x32' if g['N'] <= 32 else '64x64'
                    bm, bn = (128, 32) if cfg == '128x32' else (64, 64)
                    res = L['res'].get(cfg) or L['res'].get('64x64')
                    tl = tiles_of(g, bm, bn)
                    s = max(1, min((target + tl // 2) // tl, max(1, g['nk'] // min_steps), smax))
                    tot += lookup(res, s)[0] * L['mult']
                results.append((tot, target, min_steps, smax))
    results.sort()
    for tot, target, ms, smax in results[:12]:
        print('target %4d  min K-steps/block %d  max splits %3d : %.1f us' % (target, ms, smax, tot))
    tot, target, ms, smax = results[0]
    print('--- per layer under the best rule (target %d, min steps %d, smax %d)' % (target, ms, smax))
    for L, g in rows:
        cfg = '128x32' if g['N'] <= 32 else '64x64'
        bm, bn = (128, 32) if cfg == '128x32' else (64, 64)
        res = L['res'].get(cfg) or L['res'].get('64x64')
        tl = tiles_of(g, bm, bn)
        s = max(1, min((target + tl // 2) // tl, max(1, g['nk'] // ms), smax))
        t, k = lookup(res, s)
        bt = min((v, c, kk) for c, r in L['res'].items() for kk, v in r.items())
        print('%-34s M %6d N %4d nk %5d tiles %5d -> %s s=%-3d %6.1f us | best %s s=%d %.1f | auto %.1f' % (
            L['name'][:34], g['M'], g['N'], g['nk'], tl, cfg, k, t, bt[1], bt[2], bt[0], L['auto']))


if __name__ == '__main__':
    main()
