#!/usr/bin/env python
"""Which forward / input-gradient contractions of the training programs are split over K, and which of them hand their slabs to the
consuming BatchNorm (Session(slab_handoff=...)) instead of running a reduction launch.   python tools/list_splits.py [f32|bf16]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from action_conditioned_gans_amd import _lib, graph as G, ops as O, optim, train as T
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
G.reset_default_graph(); optim.set_data_parallel(1)
sess = G.Session(device='cuda:0', dtype=dtype)
tr = T.Trainer(sess, True, "bce", "adam", True, batch_size=32)
sess.run(G.global_variables_initializer())
import numpy as np
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (32, 64, 64, 3)).astype(np.float32); a = rng.standard_normal((32, 10)).astype(np.float32); s = rng.standard_normal((32, 5)).astype(np.float32)
tr.train_d(x, x, a); tr.train_g(x, x, a, s); torch.cuda.synchronize()
dt = _lib.ACG_BF16 if dtype == 'bf16' else _lib.ACG_F32
for op in G.get_default_graph().ops:
    if isinstance(op, (O.Conv2dOp, O.ConvDgradOp)):
        sp = sess.rt.lib.conv2d_splits(ctypes.byref(op.desc), op.which, dt)
        if sp > 1:
            cons = [c.name for c in getattr(op, 'consumers', [])] if hasattr(op, 'consumers') else ''
            print('%-44s splits %3d  slab hand-off %s' % (op.name, sp, 'yes layout %s' % (op._slab[2],) if getattr(op, '_slab', None) is not None else 'NO'))
