#!/usr/bin/env python
"""Upper bound on what running a layer's dgrad and wgrad CONCURRENTLY could save (evidence for / against a merged
dgrad+wgrad launch).  For every conv layer of the G+D step: N repetitions of (dgrad, wgrad) on one stream, against
dgrad on one stream and wgrad on another with no dependency between them at all (both free-running: more overlap than
a paired launch could ever get).  Eager launches; the kernels are 6-80 us, the host keeps ahead.
  python tools/micro/pair_probe.py > gpurun_out/pair_probe.txt"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from action_conditioned_gans_amd import _lib, graph as G, ops as O, optim, train as T   # noqa: E402


def wgrad_batch(lib, dev):
    """All weight gradients of the G step (and of the D step) back to back on one stream, against the same launches
    spread over 2 / 3 free-running streams (longest first, greedy): what batching them at the end of backward could save."""
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = G.Session(device='cuda:0')
    T.Trainer(sess, True, 'bce', 'adam', True, batch_size=32)
    groups = {'g': [], 'd': []}
    for op in G.get_default_graph().ops:
        if isinstance(op, O.ConvWgradOp) and op.name[0] in groups and len(groups[op.name[0]]) < 40:
            if not any(o.name == op.name for o in groups[op.name[0]]):
                groups[op.name[0]].append(op)
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    ptrs = [ctypes.c_void_p(s.cuda_stream) for s in streams]
    for tag, ops_ in groups.items():
        calls = []
        for op in ops_:
            d = op.desc
            nx = d.batch * d.in_h * d.in_w * max(d.in_c, d.in_pitch)
            ny = d.batch * d.out_h * d.out_w * max(d.out_c, d.out_pitch)
            nw = d.kh * d.kw * d.in_c * d.out_c
            x, y, dw = (torch.randn(n, device=dev) for n in (nx, ny, nw))
            nbytes = lib.conv2d_workspace_bytes(ctypes.byref(d), _lib.CONV_WGRAD, 0)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            fl = 2.0 * d.batch * d.out_h * d.out_w * d.kh * d.kw * d.in_c * d.out_c
            calls.append((fl, lambda s, x=x, y=y, dw=dw, d=d, ws=ws, nb=nbytes: lib.conv2d_wgrad(
                ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(dw.data_ptr()), 0.0, ctypes.byref(d), 0,
                ctypes.c_void_p(ws.data_ptr()), nb, s), (x, y, dw, ws)))
        calls.sort(key=lambda c: -c[0])

        def run(nstreams, reps=50):
            load = [0.0] * nstreams
            plan = []
            for fl, fn, _ in calls:
                k = load.index(min(load))
                load[k] += fl + 2e8        # a launch is worth ~0.2 GFLOP of time
                plan.append((k, fn))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                for k, fn in plan:
                    fn(ptrs[k])
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / reps
        run(1, 5)
        print('%s step: %d weight gradients: 1 stream %.1f us, 2 streams %.1f us, 3 streams %.1f us (incl. one host sync per batch)' % (
            tag.upper(), len(calls), min(run(1) for _ in range(3)), min(run(2) for _ in range(3)), min(run(3) for _ in range(3))))
        sys.stdout.flush()


def main():
    dev = torch.device('cuda:0')
    if '--wgrads' in sys.argv:
        return wgrad_batch(_lib.get(), dev)
    lib = _lib.get()
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = G.Session(device='cuda:0')
    T.Trainer(sess, True, 'bce', 'adam', True, batch_size=32)
    layers = {}
    for op in G.get_default_graph().ops:
        if isinstance(op, (O.ConvDgradOp, O.ConvWgradOp)):
            base = op.name.rsplit('/', 1)[0]
            layers.setdefault((base, op.desc.key()), {})['w' if isinstance(op, O.ConvWgradOp) else 'd'] = op
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    tot_seq = tot_con = 0.0
    print('# layer                                 seq us/pair   concurrent us/pair   saved')
    for (base, _), pair in layers.items():
        if 'd' not in pair or 'w' not in pair:
            continue
        d = pair['d'].desc
        nx = d.batch * d.in_h * d.in_w * max(d.in_c, d.in_pitch)
        ny = d.batch * d.out_h * d.out_w * max(d.out_c, d.out_pitch)
        nw = d.kh * d.kw * d.in_c * d.out_c
        x, y, w, dx, dw = (torch.randn(n, device=dev) for n in (nx, ny, nw, nx, nw))
        calls = []
        for key, op in (('d', pair['d']), ('w', pair['w'])):
            which = op.which
            nbytes = lib.conv2d_workspace_bytes(ctypes.byref(d), which, 0)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            pw = ctypes.c_void_p(ws.data_ptr())
            P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
            if which == _lib.CONV_DGRAD:
                calls.append((lambda s, pw=pw, nb=nbytes: lib.conv2d_dgrad(P(y), P(w), P(dx), ctypes.byref(d), 0, pw, nb, s), ws))
            elif which == _lib.CONV_FWD:       # a transposed layer's dgrad
                calls.append((lambda s, pw=pw, nb=nbytes: lib.conv2d_fwd(P(x), P(w), P(y), ctypes.byref(d), 0, pw, nb, s), ws))
            else:
                calls.append((lambda s, pw=pw, nb=nbytes: lib.conv2d_wgrad(P(x), P(y), P(dw), 0.0, ctypes.byref(d), 0, pw, nb, s), ws))
        p1, p2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
        reps = 200

        def run(sa, sb):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                calls[0][0](sa)
                calls[1][0](sb)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e6 / reps
        run(p1, p1)
        seq = min(run(p1, p1) for _ in range(3))
        con = min(run(p1, p2) for _ in range(3))
        tot_seq += seq
        tot_con += con
        print('%-40s %8.1f %16.1f %12.1f' % (base[:40], seq, con, seq - con))
        sys.stdout.flush()
    print('# total: sequential %.1f us, concurrent %.1f us, saved %.1f us per G+D step (upper bound)' % (tot_seq, tot_con, tot_seq - tot_con))


if __name__ == '__main__':
    main()
