#!/usr/bin/env python
"""Soak of train() on push TFRecords as the CLI runs it (process workers, announced frames, frame cache, logging + checkpoints +
evaluation rollouts on): ITER iterations (environment, default 6000), resident memory of this process every 1000 iterations,
finite weights and clean device-side flags at the end (LOG_EVERY: logging / checkpoint interval, default 500).  Looks for what a short test cannot: staging / event / cache growth,
worker shutdown at exit.

  python tools/soak_train_loop.py"""
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def rss_mb():
    with open('/proc/self/status') as f:
        for line in f:
            if line.startswith('VmRSS'):
                return int(line.split()[1]) / 1024.0
    return 0.0


def main():
    import torch
    import bench_train_loop as BL
    from action_conditioned_gans_amd import train as T
    iters = int(os.environ.get('ITER', '6000'))
    log_every = int(os.environ.get('LOG_EVERY', '500'))       # the reference logs and saves every 100 iterations (train.py:269-274)
    tmp = tempfile.mkdtemp(prefix='push_soak_')
    BL.make_shards(tmp, 96)
    out = tempfile.mkdtemp(prefix='push_soak_out_')
    log_dir, model_dir = os.path.join(out, 'logs'), os.path.join(out, 'models')
    os.makedirs(log_dir)
    os.makedirs(model_dir)
    stop = threading.Event()
    t0 = time.time()

    def watch():
        while not stop.wait(5.0):
            n = 0
            try:
                with open(os.path.join(log_dir, 'train.jsonl')) as f:
                    n = sum(1 for _ in f)
            except OSError:
                pass
            print('t = %5.1f s   logged intervals %3d   host RSS %7.1f MB   device memory %7.1f MB' %
                  (time.time() - t0, n, rss_mb(), torch.cuda.memory_allocated() / 2 ** 20), flush=True)
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    tr = T.train(tmp, None, None, log_dir, model_dir, True, 'bce', 'adam', True, batch_size=32, train_iter=iters, pretrain_iter=20,
                 device='cuda:0', quiet=True, eval_every=1000, log_every=log_every, data_workers='process', data_threads=16, data_cache_gb=1.0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    stop.set()
    th.join()
    ok = all(bool(torch.isfinite(tr.sess.get_value(v)).all()) for v in tr.g_vars + tr.d_vars)
    tr.sess.close()                     # raises if a device-side flag is set
    left = [t.name for t in threading.enumerate() if t.name.startswith('push-')]
    print('%d iterations in %.1f s (%.1f iterations/s incl. set-up, first epoch, logging, %d checkpoints, %d rollouts); weights finite: %s; '
          'flags clean; decode threads left: %s; host RSS %.1f MB' % (iters, dt, iters / dt, len(os.listdir(model_dir)), iters // 1000 + (1 if iters % 1000 else 0),
                                                                     ok, left or 'none', rss_mb()))
    if not ok or left:
        sys.exit(1)


if __name__ == '__main__':
    main()
