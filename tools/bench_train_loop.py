#!/usr/bin/env python
"""train() as the reference drives it (train.py:222-263: get_batch -> pair selection -> train_d / train_g with numpy in, numpy
out), on synthetic sequences and on a directory of push TFRecords (512x640 JPEGs, decoded by PushDataset's worker threads), on
this box's host cores: iterations per second of both loops and the decode rate alone.

  python tools/bench_train_loop.py [--batch 32] [--iters 200] [--records 96] [--threads N] [--dtype f32]

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
    ap.add_argument('--iters', type=int, default=200)
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

    # ---- decode rate alone: get_batch in a loop, nothing else running.  'selected': the frames of a D + G iteration announced
    # ahead (train._PairSelections does this in the loop: the union of two frame pairs per record, 3.4 of 7 frames on average)
    def needs(rng):
        need = np.zeros((args.batch, 7), bool)
        for _ in range(2):
            t = rng.integers(0, 6, args.batch)
            need[np.arange(args.batch), t] = True
            need[np.arange(args.batch), t + 1] = True
        return need
    settings = [(args.threads, 'thread', 'all', 'exact')] if args.threads is not None else \
        [(0, 'thread', 'all', 'exact'), (8, 'thread', 'all', 'exact'), (16, 'thread', 'all', 'exact'), (8, 'process', 'all', 'exact'),
         (16, 'process', 'all', 'exact'), (16, 'process', 'selected', 'exact'), (16, 'process', 'all', 'dct'), (16, 'process', 'selected', 'dct'),
         (16, 'thread', 'selected', 'dct')]
    for threads, kind, frames, decode in settings:
        rng = np.random.default_rng(5)
        with P.PushDataset(tmp, args.batch, train_val_split=1.0, num_threads=threads, workers=kind, decode=decode) as ds:
            if frames == 'selected':
                for _ in range(8):
                    ds.announce(needs(rng))
            ds.get_batch()
            ds.get_batch()
            t0, n, got = time.time(), 0, 0
            while time.time() - t0 < 4.0:
                if frames == 'selected':
                    ds.announce(needs(rng))
                got += int(np.isfinite(ds.get_batch()[0][:, :, 0, 0, 0]).sum())
                n += 1
            dt = time.time() - t0
        print('decode only: %2d %-9s %-8s %-5s %6.1f batches/s  %7.0f records/s  %8.0f JPEG frames/s' %
              (threads, kind + ('es' if kind == 'process' else 's'), frames, decode, n / dt, n * args.batch / dt, got / dt))

    # ---- the training loop, synthetic vs TFRecords: iterations per second of the RUNNING loop = (iters) / (time of a run of
    # 40 + iters iterations - time of a run of 40): session set-up, kernel loading, graph capture and worker start-up cancel
    def loop(input_path, label, scale=1, **extra):
        n_long = 40 + args.iters * scale
        kw = dict(batch_size=args.batch, pretrain_iter=0, device='cuda:0', quiet=True, eval_every=0, log_every=10 ** 9, dtype=args.dtype, **extra)
        times = []
        for iters in (40, 40, n_long):                   # (the first short run also warms the process: its time is dropped)
            torch.cuda.synchronize()
            t0 = time.time()
            tr = T.train(input_path, None, None, None, None, True, 'bce', 'adam', True, **dict(kw, train_iter=iters))
            torch.cuda.synchronize()
            times.append(time.time() - t0)
            tr.sess.close()
        rate = (n_long - 40) / (times[2] - times[1])
        print('train(%-40s): %6.1f iterations/s  (%.2f ms per D + G iteration; runs of 40 / %d iterations took %.2f / %.2f s)'
              % (label, rate, 1e3 / rate, n_long, times[1], times[2]))
        return rate
    r_syn = loop('synthetic', 'synthetic, drawn per batch', scale=2)
    r_pool = loop('synthetic', 'synthetic, pool of 8 batches', scale=10, synthetic_pool=8)        # (fast loops: longer runs)
    rows = [('tfrecords, 16 threads, all frames, exact', dict(data_workers='thread', data_threads=16, data_frames='all')),
            ('tfrecords, 16 processes, all, exact', dict(data_workers='process', data_threads=16, data_frames='all')),
            ('tfrecords, 16 processes, selected, exact', dict(data_workers='process', data_threads=16)),
            ('tfrecords, 16 processes, selected, dct', dict(data_workers='process', data_threads=16, data_decode='dct')),
            # the frame cache: the 96 records come round every 3 iterations, so past the first few iterations every frame is
            # served from memory - the rate of the second and later epochs of a run whose decoded set fits the cache
            ('tfrecords, 16 proc., selected, exact, CACHED', dict(data_workers='process', data_threads=16, data_cache_gb=4.0))]
    for label, extra in rows:
        r = loop(tmp, label, scale=10 if 'CACHED' in label else 1, **extra)
        print('   = %.2f of the drawn-per-batch synthetic loop, %.2f of the pooled one' % (r / r_syn, r / r_pool))


if __name__ == '__main__':
    main()
