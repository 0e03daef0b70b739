"""Soak run: 400 iterations of train() on synthetic data on cuda:0 (float32 and bf16); checks the weights stay finite and
reports peak memory (the whole working set is allocated at compile time).  python tools/soak_train.py [f32|bf16 ...]"""
import sys, time, torch
sys.path.insert(0, '.')
from action_conditioned_gans_amd import train as T
for dtype in (sys.argv[1:] or ['f32', 'bf16']):
    t0 = time.time()
    tr = T.train('synthetic', None, None, None, None, True, 'bce', 'adam', True, batch_size=32, seq_len=8, train_iter=int(__import__('os').environ.get('SOAK_ITERS', 400)),
                 pretrain_iter=20, device='cuda:0', quiet=True, eval_every=100, dtype=dtype)
    torch.cuda.synchronize()
    ok = all(torch.isfinite(tr.sess.get_value(v)).all().item() for v in tr.g_vars + tr.d_vars)
    tr.sess.rt.check_exchange_flags()       # raises if a one-launch BatchNorm kernel ever timed out waiting for its peers
    print('SOAK', dtype, 'ok' if ok else 'NONFINITE', '%.1f s for the run' % (time.time() - t0), 'max mem %.2f GB' % (torch.cuda.max_memory_allocated() / 1e9), flush=True)
