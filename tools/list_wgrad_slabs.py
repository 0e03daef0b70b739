#!/usr/bin/env python
"""The weight-gradient contractions of a (look-ahead) training step: splits, slab bytes written (and read back by the ONE deferred
reduction launch per optimizer), launch kind.   python tools/list_wgrad_slabs.py [f32|bf16]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from action_conditioned_gans_amd import _lib, graph as G, ops as O, optim, train as T
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
G.reset_default_graph(); optim.set_data_parallel(1)
sess = G.Session(device='cuda:0', dtype=dtype)
tr = T.Trainer(sess, True, "bce", "adam", True, batch_size=32)
sess.run(G.global_variables_initializer())
dt = _lib.ACG_BF16 if dtype == 'bf16' else _lib.ACG_F32
tot = {}
for op in G.get_default_graph().ops:
    if isinstance(op, O.ConvWgradOp) and 'pretrain' not in op.outputs[0].name:
        sp = sess.rt.lib.conv2d_splits(ctypes.byref(op.desc), _lib.CONV_WGRAD, dt)
        numel = op.outputs[0].numel
        d = op.desc
        scope = op.name.split('/')[0] + ('@' + str(d.batch))
        mb = numel * 4 * sp / 1e6 if sp > 1 else 0.0
        tot[scope] = tot.get(scope, 0.0) + mb
        print('%-44s batch %3d  K(pixels) %7d  out %8d  splits %3d  slabs %7.2f MB' % (op.name, d.batch, d.batch * d.out_h * d.out_w, numel, sp, mb))
print('slab MB by scope@batch:', {k: round(v, 1) for k, v in tot.items()})
