"""bf16 pipeline smoke: one D + G step in f32 and bf16 sessions from the same weights; prints the deviations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from action_conditioned_gans_amd import graph as G, optim, train as T
from oracle import models as OM

B, S, K = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 64, 5
params = OM.init_params(True, batch=B, img=S, ksize=K, seed=9, dtype=torch.float32)
rng = np.random.default_rng(21)
x = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
y = np.clip(np.roll(x, 2, axis=2) + 0.05 * rng.standard_normal(x.shape).astype(np.float32), -1, 1)
a = rng.standard_normal((B, 10)).astype(np.float32)
s = rng.standard_normal((B, 5)).astype(np.float32)
out = {}
for dt in ('f32', 'bf16'):
    G.reset_default_graph()
    optim.set_data_parallel(1)
    sess = G.Session(device='cuda:0', dtype=dt)
    tr = T.Trainer(sess, True, 'bce', 'rmsprop', True, batch_size=B, img_size=S, ksize=K)
    sess.run(G.global_variables_initializer())
    g = G.get_default_graph()
    for n, v in g.variables.items():
        sess.set_value(v, params[n])
    frame, state, summ = tr.test(x, y, a)
    dsumm = tr.train_d(x, y, a, summarize=True)
    res = sess.run([tr.g_opt_op, tr.g_loss], tr._feed(x, y, a, s))
    for _ in range(3):
        tr.train_d(x, y, a); tr.train_g(x, y, a, s)
    torch.cuda.synchronize()
    w = {n: sess.get_value(v).double() for n, v in g.variables.items()}
    out[dt] = (frame, state, dsumm['discriminator_loss'], float(res[1][0]), w)
    print(dt, 'd_loss', dsumm['discriminator_loss'], 'g_loss', float(res[1][0]), 'psnr', summ['g_psnr'], flush=True)
rel = lambda p, q: float(np.abs(np.asarray(p, np.float64) - np.asarray(q, np.float64)).max() / max(np.abs(np.asarray(q)).max(), 1e-30))
f, h = out['f32'], out['bf16']
print('frame rel', rel(h[0], f[0]), 'state rel', rel(h[1], f[1]))
worst = max(((float((h[4][n] - f[4][n]).abs().max() / max(f[4][n].abs().max(), 1e-6)), n) for n in f[4]))
print('worst weight dev after 4 steps', worst)
print('BF16_SMOKE_DONE')
