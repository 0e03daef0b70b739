#!/usr/bin/env python
"""train() as the reference drives it (train.py:222-263: get_batch -> pair selection -> train_d / train_g with numpy in, numpy
out), on synthetic sequences and on a directory of push TFRecords (512x640 JPEGs, decoded by PushDataset's worker threads), on
this box's host cores: iterations per second of both loops and the decode rate alone.

  python tools/bench_train_loop.py [--batch 32] [--iters 150] [--records 96] [--threads N] [--dtype f32]

The TFRecord shards are written to a temporary directory first (synthetic smooth images, JPEG quality 90; a record holds the 7
frames the reader uses).  The stream wraps around: the reader re-reads and re-decodes the shards, nothing is cached."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_shards(path, n_records, per_shard=16, seed=0):
    from action_conditioned_gans_amd import push_data as P
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:P.ORIGINAL_HEIGHT, 0:P.ORIGINAL_WIDTH].astype(np.float32)
    k = 0
    while k < n_records:
        seqs = []
        for _ in range(min(per_shard, n_records - k)):
            frames = []
            a, b, c = rng.uniform(0.002, 0.02, 3)
            for t in range(7):
                img = np.stack([127 + 90 * np.sin(a * xx + b * yy + 0.3 * t), 127 + 90 * np.cos(b * xx - 0.2 * t), 127 + 90 * np.sin(c * yy)], -1)
                img += rng.normal(0, 6, img.shape)               # sensor-like noise: a realistic JPEG size / decode cost
                frames.append(np.clip(img, 0, 255).astype(np.uint8))
            seqs.append((np.stack(frames), rng.standard_normal((7, 5)).astype(np.float32), rng.standard_normal((7, 5)).astype(np.float32)))
        P.write_push_tfrecord(os.path.join(path, 'push_%03d.tfrecord' % (k // per_shard)), seqs, quality=90)
        k += len(seqs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--iters', type=int, default=150)
    ap.add_argument('--records', type=int, default=96)
    ap.add_argument('--threads', type=int, default=None)
    ap.add_argument('--dtype', default='f32')
    args = ap.parse_args()
    import torch
    from action_conditioned_gans_amd import push_data as P, train as T
    cpus = len(os.sched_getaffinity(0))
    print('# host: %d CPUs usable by this process (os.cpu_count %d); batch %d, %d iterations per loop' % (cpus, os.cpu_count(), args.batch, args.iters))
    tmp = tempfile.mkdtemp(prefix='push_bench_')
    t0 = time.time()
    make_shards(tmp, args.records)
    size = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp))
    print('# wrote %d records (%.1f MB, %.0f KB per record) in %.1f s' % (args.records, size / 1e6, size / 1e3 / args.records, time.time() - t0))

    # ---- decode rate alone: get_batch in a loop, nothing else running
    settings = [(args.threads, 'thread')] if args.threads is not None else [(0, 'thread'), (4, 'thread'), (8, 'thread'), (16, 'thread'), (8, 'process'), (16, 'process')]
    for threads, kind in settings:
        with P.PushDataset(tmp, args.batch, train_val_split=1.0, num_threads=threads, workers=kind) as ds:
            ds.get_batch()
            ds.get_batch()
            t0, n = time.time(), 0
            while time.time() - t0 < 4.0:
                ds.get_batch()
                n += 1
            dt = time.time() - t0
        print('decode only: %2d %-9s %6.1f batches/s  %7.0f records/s  %8.0f JPEG frames/s' % (threads, kind + ('es' if kind == 'process' else 's'), n / dt, n * args.batch / dt, n * args.batch * 7 / dt))

    # ---- the training loop, synthetic vs TFRecords (same iteration count; pretraining and evaluation off; logging off)
    def loop(input_path, label, **extra):
        torch.cuda.synchronize()
        kw = dict(batch_size=args.batch, train_iter=args.iters + 20, pretrain_iter=0, device='cuda:0', quiet=True, eval_every=0, log_every=10 ** 9,
                  dtype=args.dtype, **extra)
        # warm: the first 20 iterations (kernel loading, graph capture) are timed separately by running a short loop first
        tr = T.train(input_path, None, None, None, None, True, 'bce', 'adam', True, **dict(kw, train_iter=20))
        tr.sess.close()
        t0 = time.time()
        tr = T.train(input_path, None, None, None, None, True, 'bce', 'adam', True, **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
        tr.sess.close()
        print('train(%-22s): %6.1f iterations/s  (%.2f ms per D + G iteration, %d iterations incl. session set-up and 20 warm-up iterations)'
              % (label, (args.iters + 20) / dt, dt / (args.iters + 20) * 1e3, args.iters + 20))
        return (args.iters + 20) / dt
    r_syn = loop('synthetic', 'synthetic')
    r_thr = loop(tmp, 'tfrecords, 16 threads', data_workers='thread', data_threads=16)
    r_prc = loop(tmp, 'tfrecords, 16 processes', data_workers='process', data_threads=16)
    print('tfrecords / synthetic = %.2f (threads), %.2f (processes)' % (r_thr / r_syn, r_prc / r_syn))


if __name__ == '__main__':
    main()
