"""How far do two rounding-level-different runs of the same training drift apart, and is the look-ahead path within that?
Variants from the same weights / inputs, batch 8, float32, RMSProp: plain (reference), plain without HIP graphs, plain with
another summation order (slab hand-off off), look-ahead with and without HIP graphs.  Per iteration: relative distance of the
generated frames and of the G / D flat gradients to the reference run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import train_cases as TC
from action_conditioned_gans_amd import graph as G, optim, train as T

x, y, a, s = TC.MG.inputs(8)
xb, yb, ab, sb = [np.ascontiguousarray(np.roll(t, 3, axis=0)[::-1]) for t in (y, x, a, s)]
def run(dna, use, **kw):
    G.reset_default_graph(); optim.set_data_parallel(1)
    sess = G.Session(device='cuda:0', **kw)
    tr = T.Trainer(sess, True, 'bce', 'rmsprop', dna, batch_size=8, ksize=5)
    sess.run(G.global_variables_initializer())
    out = []
    for it in range(5):
        tr.train_d(x, y, a, next_g=(xb, ab) if use else None)
        dg = [sess._materialize(t).clone() for t in G.get_default_graph().state if t.name == 'd_opt/flat_grad'][0]
        f = tr.train_g(xb, yb, ab, sb)
        gg = [sess._materialize(t).clone() for t in G.get_default_graph().state if t.name == 'g_opt/flat_grad'][0]
        out.append((np.array(f, copy=True), dg.cpu().double(), gg.cpu().double()))
    sess.close()
    return out
nrel = lambda p, q: float(np.linalg.norm(np.asarray(p, np.float64) - np.asarray(q, np.float64)) / np.linalg.norm(np.asarray(q, np.float64)))
for dna in (True, False):
    ref = run(dna, False)
    for label, use, kw in (('plain, no graphs', False, dict(use_hip_graphs=False)), ('plain, slab hand-off off', False, dict(slab_handoff=False)),
                           ('plain, epilogue stats off', False, dict(epilogue_stats=False)),
                           ('look-ahead', True, {}), ('look-ahead, no graphs', True, dict(use_hip_graphs=False))):
        o = run(dna, use, **kw)
        print('%-5s %-26s' % ('dna' if dna else 'plain', label), ' | '.join('it%d f %.1e dD %.1e dG %.1e' % (i, nrel(o[i][0], ref[i][0]), nrel(o[i][1], ref[i][1]), nrel(o[i][2], ref[i][2])) for i in range(5)), flush=True)
