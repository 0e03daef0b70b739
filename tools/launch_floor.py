#!/usr/bin/env python
"""Cost of a dependent kernel launch inside a captured HIP graph on this part: N trivial acg_step_inc kernels
(and N small acg_add kernels) captured back to back, replayed, wall time / N.  Calibrates how much of the
training step is launch floor rather than kernel work."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from action_conditioned_gans_amd import _lib   # noqa: E402

lib, dev = _lib.get(), torch.device('cuda:0')
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
a = torch.zeros(4096, device=dev)
b = torch.ones(4096, device=dev)
c = torch.zeros(4096, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for name, call in (('step_inc (1 thread)', lambda s: lib.step_inc(p(cnt), s)),
                   ('add 4096 floats (16 blocks)', lambda s: lib.add(p(a), p(b), p(c), 4096, 0, s))):
    for n in (100, 400):
        s0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        call(s0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sp = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for _ in range(n):
                call(sp)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print('%-30s graph of %4d kernels: %8.1f us per replay = %.2f us per kernel' % (name, n, dt * 1e6, dt * 1e6 / n))
