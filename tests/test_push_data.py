"""Host-side input format of the path: push-dataset TFRecords read without TensorFlow."""
import io
import os
import struct

import numpy as np
import pytest

from action_conditioned_gans_amd import push_data as P

HERE = os.path.dirname(os.path.abspath(__file__))


def smooth_frames(rng, n=7, h=P.ORIGINAL_HEIGHT, w=P.ORIGINAL_WIDTH):
    """low-frequency images: JPEG keeps them almost exactly"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = []
    for _ in range(n):
        a, b, c = rng.uniform(0.002, 0.01, 3)
        img = np.stack([127 + 100 * np.sin(a * xx + b * yy), 127 + 100 * np.cos(b * xx), 127 + 100 * np.sin(c * yy)], -1)
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return np.stack(out)


def make_shard(path, rng, n_records):
    seqs = [(smooth_frames(rng), rng.standard_normal((7, 5)).astype(np.float32), rng.standard_normal((7, 5)).astype(np.float32))
            for _ in range(n_records)]
    P.write_push_tfrecord(path, seqs, quality=95)
    return seqs


def test_crc32c_known_answers():
    assert P.crc32c(b'123456789') == 0xE3069283          # the CRC-32C check value
    assert P.crc32c(b'') == 0
    assert P.crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4: 32 bytes of zeros
    assert P.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43    # RFC 3720 B.4: 32 bytes of ones


def test_record_framing_and_corruption(tmp_path):
    path = str(tmp_path / 'a.tfrecord')
    payloads = [b'', b'x', os.urandom(1000)]
    P.write_records(path, payloads)
    assert list(P.read_records(path, verify_crc=True)) == payloads
    raw0 = open(path, 'rb').read()
    assert [raw0[o:o + n] for o, n in P.record_spans(path)] == payloads          # the spans worker processes fetch their records by
    raw = bytearray(open(path, 'rb').read())
    raw[-10] ^= 0x40                                     # flip a bit inside the last payload
    open(path, 'wb').write(raw)
    assert len(list(P.read_records(path))) == 3          # unchecked read still frames correctly
    with pytest.raises(IOError):
        list(P.read_records(path, verify_crc=True))
    open(path, 'wb').write(raw[:-3])
    with pytest.raises(IOError):
        list(P.read_records(path))


def test_example_round_trip_and_unpacked_floats():
    feats = {'a/b': b'\x00\x01jpeg', 'v': np.array([1.5, -2.0, 3.25], np.float32)}
    got = P.parse_example(P.serialize_example(feats))
    assert got['a/b'] == [b'\x00\x01jpeg'] and np.array_equal(got['v'], feats['v'])
    assert P.parse_example(P.serialize_example(feats), keys={'v'}).keys() == {'v'}
    # FloatList written UNPACKED (wire type 5 per element) and an Int64List, as other writers may emit them
    fl = b''.join(P._enc_varint((1 << 3) | 5) + struct.pack('<f', x) for x in (0.5, 7.0))
    il = P._ld(1, P._enc_varint(3) + P._enc_varint((1 << 64) - 2))        # packed [3, -2]
    ex = P._ld(1, P._ld(1, P._ld(1, b'f') + P._ld(2, P._ld(2, fl))) + P._ld(1, P._ld(1, b'i') + P._ld(2, P._ld(3, il))))
    got = P.parse_example(ex)
    assert np.array_equal(got['f'], np.array([0.5, 7.0], np.float32)) and np.array_equal(got['i'], np.array([3, -2]))


def test_resize_area_matches_box_integration():
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (10, 14, 3)).astype(np.float32)
    got = P.resize_area(img, 4, 6)                       # non-integer ratios 2.5 and 2.333
    want = np.zeros((4, 6, 3))
    for o in range(4):
        for p in range(6):
            y0, y1, x0, x1 = o * 2.5, (o + 1) * 2.5, p * 14 / 6.0, (p + 1) * 14 / 6.0
            acc = 0.0
            for y in range(10):
                for x in range(14):
                    wy = max(0.0, min(y1, y + 1) - max(y0, y))
                    wx = max(0.0, min(x1, x + 1) - max(x0, x))
                    acc = acc + wy * wx * img[y, x].astype(np.float64)
            want[o, p] = acc / ((y1 - y0) * (x1 - x0))
    assert np.abs(got - want).max() < 1e-3
    box = P.resize_area(img[:8, :12], 4, 6)              # integer ratio: plain 2x2 means
    assert np.allclose(box, img[:8, :12].reshape(4, 2, 6, 2, 3).mean((1, 3)), atol=1e-4)


def test_resize_area_known_answers_by_hand():
    """tf.image.resize_area as TF 1.0 defines it (core/kernels/resize_area_op.cc: output pixel o integrates the input over
    [o * scale, (o + 1) * scale), scale = in / out, every input pixel weighted by the length of its overlap, the sum divided by
    scale).  Values worked out by hand, not by the code under test; odd sizes and non-integer ratios, down and up."""
    col = lambda v: np.asarray(v, np.float32).reshape(-1, 1, 1)               # a [n, 1, 1] image: the row axis alone
    # 3 -> 2, scale 1.5:  [0, 1.5) = p0 + p1/2,  [1.5, 3) = p1/2 + p2
    assert np.allclose(P.resize_area(col([3, 6, 9]), 2, 1).ravel(), [(3 + 3) / 1.5, (3 + 9) / 1.5], atol=1e-5)             # 4, 8
    # 5 -> 2, scale 2.5:  p0 + p1 + p2/2,  p2/2 + p3 + p4
    assert np.allclose(P.resize_area(col([10, 20, 30, 40, 50]), 2, 1).ravel(), [18.0, 42.0], atol=1e-5)
    # 5 -> 3, scale 5/3:  p0 + 2/3 p1,  1/3 p1 + p2 + 1/3 p3,  2/3 p3 + p4
    assert np.allclose(P.resize_area(col([3, 6, 9, 12, 15]), 3, 1).ravel(), [4.2, 9.0, 13.8], atol=1e-5)
    # 7 -> 4, scale 1.75:  p0 + .75 p1,  .25 p1 + p2 + .5 p3,  .5 p3 + p4 + .25 p5,  .75 p5 + p6   (p = 1..7)
    assert np.allclose(P.resize_area(col([1, 2, 3, 4, 5, 6, 7]), 4, 1).ravel(), [2.5 / 1.75, 5.5 / 1.75, 8.5 / 1.75, 11.5 / 1.75], atol=1e-5)
    # enlarging, 2 -> 3, scale 2/3:  [0, 2/3) lies in p0,  [2/3, 4/3) is half p0 half p1,  [4/3, 2) lies in p1
    assert np.allclose(P.resize_area(col([10, 40]), 3, 1).ravel(), [10.0, 25.0, 40.0], atol=1e-5)
    # two axes at once, 3 x 5 -> 2 x 3: the weights are separable, so with img[y, x] = r[y] * c[x] the answer is the outer
    # product of the two one-axis answers above
    r, c = np.array([3, 6, 9], np.float32), np.array([3, 6, 9, 12, 15], np.float32)
    got = P.resize_area((r[:, None] * c[None, :])[..., None], 2, 3)[..., 0]
    assert np.allclose(got, np.outer([4.0, 8.0], [4.2, 9.0, 13.8]), atol=1e-4)
    # channels are independent, and a constant image stays constant at any ratio
    assert np.allclose(P.resize_area(np.full((9, 11, 3), 7.0, np.float32), 4, 5), 7.0, atol=1e-5)
    # the push pipeline's own case (ops.py:190-194: 512 -> 64) is the integer-ratio branch: 8 x 8 block means
    ramp = np.arange(16 * 16, dtype=np.float32).reshape(16, 16, 1)
    assert np.allclose(P.resize_area(ramp, 2, 2)[..., 0], [[59.5, 67.5], [187.5, 195.5]], atol=1e-4)


def test_crop_or_pad_center_known_answers_by_hand():
    """tf.image.resize_image_with_crop_or_pad (TF 1.0 image_ops_impl.py): with diff = target - size, the crop offset is
    (-diff) // 2 and the pad offset diff // 2 - for an odd difference the extra row / column is cropped or padded AFTER."""
    row = np.arange(10).reshape(1, 10, 1)
    assert P.crop_or_pad_center(row, 1, 7)[0, :, 0].tolist() == [1, 2, 3, 4, 5, 6, 7]               # diff -3: skip 1, drop 2 behind
    assert P.crop_or_pad_center(row, 1, 6)[0, :, 0].tolist() == [2, 3, 4, 5, 6, 7]                  # diff -4: 2 and 2
    five = np.arange(1, 6).reshape(1, 5, 1)
    assert P.crop_or_pad_center(five, 1, 8)[0, :, 0].tolist() == [0, 1, 2, 3, 4, 5, 0, 0]           # diff +3: pad 1 before, 2 after
    assert P.crop_or_pad_center(five, 3, 5)[:, :, 0].tolist() == [[0] * 5, [1, 2, 3, 4, 5], [0] * 5]  # rows: diff +2: 1 and 1
    # crop one axis, pad the other (the order does not matter: they are independent)
    img = np.arange(12).reshape(3, 4, 1)
    assert P.crop_or_pad_center(img, 4, 2)[:, :, 0].tolist() == [[1, 2], [5, 6], [9, 10], [0, 0]]
    # the push pipeline's own case (ops.py:186-188): 512 x 640 -> 512 x 512 drops 64 columns on each side
    wide = np.zeros((512, 640, 1), np.uint8)
    wide[:, 64] = 1
    wide[:, 575] = 2
    out = P.crop_or_pad_center(wide, 512, 512)
    assert out.shape == (512, 512, 1) and (out[:, 0] == 1).all() and (out[:, 511] == 2).all()


def test_crop_or_pad_center():
    img = np.arange(6 * 10 * 1).reshape(6, 10, 1)
    assert np.array_equal(P.crop_or_pad_center(img, 6, 6), img[:, 2:8])
    out = P.crop_or_pad_center(img, 8, 6)
    assert out.shape == (8, 6, 1) and np.array_equal(out[1:7], img[:, 2:8]) and not out[0].any() and not out[7].any()


def test_decode_frame_is_crop_then_8x8_means():
    from PIL import Image
    rng = np.random.default_rng(1)
    frame = smooth_frames(rng, 1)[0]
    b = io.BytesIO()
    Image.fromarray(frame).save(b, format='JPEG', quality=95)
    dec = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert('RGB')).astype(np.float64)
    got = P.decode_frame(b.getvalue())
    assert got.shape == (64, 64, 3) and got.dtype == np.float32 and -1.0 <= got.min() and got.max() <= 1.0
    for (p, q) in ((0, 0), (63, 63), (10, 50), (31, 32)):
        want = dec[8 * p:8 * p + 8, 64 + 8 * q:64 + 8 * q + 8].mean((0, 1)) / 127.5 - 1.0    # 640 -> centre 512: offset 64
        assert np.abs(got[p, q] - want).max() < 1e-5
    assert np.abs(got - (P.resize_area(frame[:, 64:576], 64, 64) / 127.5 - 1)).max() < 0.03      # JPEG is near-lossless here


def test_push_dataset_batches_split_and_ranks(tmp_path):
    rng = np.random.default_rng(2)
    seqs = []
    for k in range(4):
        seqs.append(make_shard(str(tmp_path / ('push_%02d.tfrecord' % k)), rng, 2))
    ds = P.PushDataset(str(tmp_path), batch_size=3, train_val_split=0.75, training=True, verify_crc=True)
    assert [os.path.basename(f) for f in ds.files] == ['push_00.tfrecord', 'push_01.tfrecord', 'push_02.tfrecord']
    val = P.PushDataset(str(tmp_path), batch_size=1, train_val_split=0.75, training=False)
    assert [os.path.basename(f) for f in val.files] == ['push_03.tfrecord']
    img, img2, acts, states = ds.get_batch()
    assert img.shape == (3, 7, 64, 64, 3) and img is img2 and acts.shape == (3, 7, 10) and states.shape == (3, 7, 5)
    assert np.array_equal(acts[:, :, 5:], states) and img.dtype == np.float32
    # every record of the batch is one of the written sequences, vectors bit-exact
    written = {tuple(np.round(a[0], 5)): (a, s) for shard in seqs for (_, a, s) in shard}
    for b in range(3):
        a, s = written[tuple(np.round(acts[b, 0, :5], 5))]
        assert np.array_equal(acts[b, :, :5], a) and np.array_equal(acts[b, :, 5:], s)
    # two ranks see disjoint halves of the same stream
    r0 = P.PushDataset(str(tmp_path), 3, train_val_split=1.0, rank=0, world_size=2)
    r1 = P.PushDataset(str(tmp_path), 3, train_val_split=1.0, rank=1, world_size=2)
    k0 = {tuple(np.round(v, 5)) for v in r0.get_batch()[2][:, 0, :5]}
    k1 = {tuple(np.round(v, 5)) for v in r1.get_batch()[2][:, 0, :5]}
    assert len(k0) == 3 and len(k1) == 3 and not (k0 & k1)
    no_state = P.PushDataset(str(tmp_path), 2, use_state=False, train_val_split=1.0)
    assert not no_state.get_batch()[2].any()
    with pytest.raises(RuntimeError, match='No data files found'):
        P.PushDataset(str(tmp_path / 'missing'), 2)


def test_prefetch_workers_keep_stream_order_and_shut_down(tmp_path):
    """PushDataset decodes on worker threads into a bounded queue (ops.py:209-213: num_threads = batch_size,
    a capacity of records) - and the batches must not depend on that: same records, same order, bit-identical arrays for 0
    (synchronous), 1, 3 and 8 workers and for a queue smaller than a batch; per-rank streams stay disjoint; close() ends every
    thread it started; a damaged shard surfaces from get_batch as the reader's own exception instead of a hang."""
    import threading
    rng = np.random.default_rng(4)
    for k in range(3):
        make_shard(str(tmp_path / ('push_%02d.tfrecord' % k)), rng, 3)
    before = {t.name for t in threading.enumerate()}
    ref = P.PushDataset(str(tmp_path), batch_size=4, train_val_split=1.0, num_threads=0)
    want = [ref.get_batch() for _ in range(4)]                  # 16 records: the 9 on disk wrap around, reshuffled
    assert ref._prefetch is None
    for threads, capacity, kind in ((1, None, 'thread'), (3, 2, 'thread'), (8, 64, 'thread'), (2, None, 'process')):
        with P.PushDataset(str(tmp_path), batch_size=4, train_val_split=1.0, num_threads=threads, capacity=capacity, workers=kind) as ds:
            assert ds.num_threads == threads and ds._prefetch is None      # the workers start with the first get_batch / announce
            for w in want:
                got = ds.get_batch()
                assert ds._prefetch is not None
                assert all(np.array_equal(g, e) for g, e in zip(got, w)), (threads, capacity, kind)
        assert ds._prefetch is None
    # default thread count: the batch size, capped by the CPUs of this process (ops.py:212 num_threads=batch_size)
    ds = P.PushDataset(str(tmp_path), batch_size=2, train_val_split=1.0)
    assert 1 <= ds.num_threads <= 2 and ds.capacity == 8
    ds.get_batch()
    ds.close()
    ds.close()                                                  # idempotent
    pf = P._Prefetcher(iter([]), lambda r: r, 1, 1)
    pf.close()
    with pytest.raises(RuntimeError, match='closed'):
        pf.get()
    import time
    time.sleep(0.2)
    left = {t.name for t in threading.enumerate()} - before
    assert not [n for n in left if n.startswith('push-')], left
    # two ranks through the workers: disjoint halves of one stream
    with P.PushDataset(str(tmp_path), 3, train_val_split=1.0, rank=0, world_size=2, num_threads=2) as r0, \
            P.PushDataset(str(tmp_path), 3, train_val_split=1.0, rank=1, world_size=2, num_threads=2) as r1:
        k0 = {tuple(np.round(v, 5)) for v in r0.get_batch()[2][:, 0, :5]}
        k1 = {tuple(np.round(v, 5)) for v in r1.get_batch()[2][:, 0, :5]}
        assert len(k0) == 3 and len(k1) == 3 and not (k0 & k1)
    # a truncated shard: the reader's IOError comes out of get_batch (and again on the next call), nothing hangs
    bad = tmp_path / 'bad'
    bad.mkdir()
    make_shard(str(bad / 'push_00.tfrecord'), rng, 2)
    raw = open(str(bad / 'push_00.tfrecord'), 'rb').read()
    open(str(bad / 'push_00.tfrecord'), 'wb').write(raw[:len(raw) - 7])
    with P.PushDataset(str(bad), 2, train_val_split=1.0, num_threads=2) as ds:
        for _ in range(2):
            with pytest.raises(IOError):
                ds.get_batch()


def test_announced_frames_are_the_only_ones_decoded(tmp_path):
    """Round 5: a training step reads 2 of a record's 7 frames (train.py:231-232), the reference decodes all 7 (ops.py:171-196).
    PushDataset.announce(need) tells the workers, batches ahead, which frames of a coming batch will be read: those come back
    bit-identical to the full decode, every other frame is NaN (a read of a frame nobody asked for cannot go unnoticed), the
    pose vectors are complete, the record order is untouched; a batch without an announcement, or one the workers reached
    before it was announced, is decoded in full.  Synchronous, threads and processes."""
    rng = np.random.default_rng(8)
    for k in range(2):
        make_shard(str(tmp_path / ('push_%02d.tfrecord' % k)), rng, 3)
    with P.PushDataset(str(tmp_path), batch_size=3, train_val_split=1.0, num_threads=0) as ref:
        want = [ref.get_batch() for _ in range(5)]
    needs = []
    for k in range(5):
        need = np.zeros((3, 7), bool)
        t = rng.integers(0, 6, 3)
        need[np.arange(3), t] = True
        need[np.arange(3), t + 1] = True
        if k == 1:
            need[0] = False                                     # a record nobody reads at all
        needs.append(need)
    for threads, kind in ((0, 'thread'), (2, 'thread'), (2, 'process')):
        with P.PushDataset(str(tmp_path), batch_size=3, train_val_split=1.0, num_threads=threads, workers=kind) as ds:
            for k in (0, 1, 2):
                ds.announce(needs[k])                           # three batches ahead of the first read
            for k in range(5):
                if k == 3:
                    got = ds.get_batch()                        # batch 3: never announced -> complete
                    assert all(np.array_equal(g, w) for g, w in zip(got, want[3])), (threads, kind)
                    ds.announce(needs[3])                       # too late for batch 3: counted, not applied
                    ds.announce(needs[4])
                    continue
                img, img2, act, st = ds.get_batch()
                assert img is img2 and np.array_equal(act, want[k][2]) and np.array_equal(st, want[k][3])
                if k == 4 and threads:                          # the workers ran ahead of this announcement: full decode is fine
                    sel = np.isfinite(img).all(axis=(2, 3, 4))
                    assert (sel | ~needs[4]).all()
                else:
                    assert np.array_equal(np.isfinite(img).all(axis=(2, 3, 4)), needs[k]), (threads, kind, k)
                    assert np.isnan(img[~needs[k]]).all()
                assert np.array_equal(img[needs[k]], want[k][0][needs[k]]), (threads, kind, k)
            with pytest.raises(ValueError, match='boolean array'):
                ds.announce(np.zeros((2, 7), bool))


def test_frame_cache_serves_later_epochs_with_the_same_bits(tmp_path):
    """cache_bytes > 0: decoded frames are kept and a record that comes round again is not decoded again.  6 records on disk,
    batches of 3: from the third batch on every record has been seen.  With and without announcements, synchronous and through
    workers: the batches equal those of an uncached reader bit for bit (frames that were asked for; pose vectors always), the
    hit / miss counters add up, later epochs decode only frames never asked for before, and a budget smaller than one frame
    caches nothing."""
    rng = np.random.default_rng(12)
    for k in range(2):
        make_shard(str(tmp_path / ('push_%02d.tfrecord' % k)), rng, 3)
    with P.PushDataset(str(tmp_path), batch_size=3, train_val_split=1.0, num_threads=0) as ref:
        want = [ref.get_batch() for _ in range(8)]
    needs = []
    for k in range(8):
        need = np.zeros((3, 7), bool)
        t = rng.integers(0, 6, 3)
        need[np.arange(3), t] = True
        need[np.arange(3), t + 1] = True
        needs.append(need)
    for threads, kind in ((0, 'thread'), (2, 'thread'), (2, 'process')):
        # (a) everything decoded: epoch 1 misses, later epochs hit
        with P.PushDataset(str(tmp_path), batch_size=3, train_val_split=1.0, num_threads=threads, workers=kind, capacity=3, cache_bytes=1 << 30) as ds:
            for k in range(8):
                got = ds.get_batch()
                assert isinstance(got[0], np.ndarray) and all(np.array_equal(g, w) for g, w in zip(got, want[k])), (threads, kind, k)
            assert ds.cache_hits + ds.cache_misses == 8 * 3 * 7
            assert ds.cache_misses >= 6 * 7 and ds.cache_hits >= 3 * 3 * 7      # (workers run ahead: a record may be drawn again before its first decode was kept)
            assert ds._cached_bytes == 6 * (7 * 64 * 64 * 3 * 4 + 2 * 7 * 5 * 4)
        # (b) announced frames: sparse batches, the cache fills frame by frame
        with P.PushDataset(str(tmp_path), batch_size=3, train_val_split=1.0, num_threads=threads, workers=kind, capacity=3, cache_bytes=1 << 30) as ds:
            for k in range(3):
                ds.announce(needs[k])
            for k in range(8):
                if k + 3 < 8:
                    ds.announce(needs[k + 3])
                img, _, act, st = ds.get_batch()
                assert isinstance(img, P.SparseFrames) and img.shape == (3, 7, 64, 64, 3) and len(img) == 3
                assert np.array_equal(act, want[k][2]) and np.array_equal(st, want[k][3])
                assert np.array_equal(img[needs[k]], want[k][0][needs[k]]), (threads, kind, k)
                first = np.zeros((3, 7), bool)
                first[np.arange(3), needs[k].argmax(axis=1)] = True              # the loop's own access: one frame per record
                assert np.array_equal(img[first], want[k][0][first])
                dense = np.asarray(img)
                assert np.array_equal(np.isfinite(dense).all(axis=(2, 3, 4)), needs[k]) and np.array_equal(dense[needs[k]], want[k][0][needs[k]])
                assert np.isnan(img[~needs[k]]).all()                            # a frame nobody asked for reads as NaN
            assert ds.cache_hits + ds.cache_misses == int(sum(n.sum() for n in needs))
            assert ds.cache_hits > 0 and ds._cached_bytes <= 6 * (7 * 64 * 64 * 3 * 4 + 2 * 7 * 5 * 4)
    with P.PushDataset(str(tmp_path), batch_size=3, train_val_split=1.0, num_threads=0, cache_bytes=1000) as ds:
        for k in range(4):
            assert np.array_equal(ds.get_batch()[0], want[k][0])
        assert ds.cache_hits == 0 and ds._cached_bytes == 0 and not ds._cache


def test_pair_selections_are_the_loops_own_draws_made_early():
    """train._PairSelections draws the frame-pair selections of coming iterations ahead of time so that the dataset can be told
    which frames to decode.  They must be the numbers the loop would have drawn call by call from numpy's global generator
    (train.py:14 seeds it; 231-232 one selection per pretraining iteration, 249-250 one per D step, 258-259 one more for the G
    step), and the announcements must be one per get_batch, in order: the union of the pairs read from that batch."""
    from action_conditioned_gans_amd import train as T
    from action_conditioned_gans_amd.util import build_all_mask
    mask, B, D, pre, iters = build_all_mask(7), 5, 3, 2, 9
    np.random.seed(7)
    want = []
    for i in range(iters):
        want.append([T.select_pairs(np.random.randint, mask, B) for _ in range(1 if i < pre else D + 1)])

    class Sink:
        def __init__(self):
            self.needs = []

        def announce(self, need):
            self.needs.append(need.copy())
    np.random.seed(7)
    sink = Sink()
    sel = T._PairSelections(mask, B, D, pre, iters, sink, ahead=4)
    state = np.random.get_state()[1].copy()
    for i in range(iters):
        got = sel.next()
        assert len(sink.needs) >= min(pre, i + 5) + max(0, min(iters, i + 5) - pre) * D      # announced 4 iterations ahead
        assert len(got) == len(want[i])
        for (gs, ge), (ws, we) in zip(got, want[i]):
            assert np.array_equal(gs, ws) and np.array_equal(ge, we)
    assert np.array_equal(np.random.get_state()[1], state)        # the global generator itself is left alone
    k = 0
    for i in range(iters):                                        # one announcement per batch the loop fetches
        if i < pre:
            assert np.array_equal(sink.needs[k], want[i][0][0] | want[i][0][1])
            k += 1
            continue
        for j in range(D):
            need = want[i][j][0] | want[i][j][1]
            if j == D - 1:
                need = need | want[i][D][0] | want[i][D][1]
            assert np.array_equal(sink.needs[k], need), (i, j)
            k += 1
    assert k == len(sink.needs) and all(2 <= n.sum(axis=1).min() and n.sum(axis=1).max() <= 4 for n in sink.needs)
    # a source without announce (SyntheticPush) is simply not told
    assert T._PairSelections(mask, B, D, pre, iters, object()).announce is None


def test_dct_decode_is_close_to_the_exact_frames():
    """decode='dct' (opt-in): the reduction inside libjpeg's inverse DCT instead of decode -> crop -> 8x8 box mean.  Not the
    reference's arithmetic; the distance is measured here on 512x640 frames with texture and noise: at most 3 levels of 255
    anywhere, below 0.7 level on average, at both 64x64 (1x1 IDCT) and 128x128 (2x2 IDCT); a ratio libjpeg cannot serve
    (below 2) and a non-JPEG file take the exact path bit for bit."""
    import io
    from PIL import Image
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:512, 0:640].astype(np.float32)
    img = np.stack([127 + 90 * np.sin(0.011 * xx + 0.006 * yy), 127 + 90 * np.cos(0.008 * xx), 127 + 90 * np.sin(0.014 * yy)], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(img).save(b, format='JPEG', quality=90)
    for size in (64, 128):
        exact, fast = P.decode_frame(b.getvalue(), size), P.decode_frame(b.getvalue(), size, dct=True)
        assert fast.shape == exact.shape == (size, size, 3) and fast.dtype == np.float32
        d = np.abs(fast - exact) * 127.5
        assert d.max() <= 3.0 and d.mean() <= 0.7, (size, d.max(), d.mean())
        assert d.max() > 0                                  # (it IS another arithmetic: the test would notice a silent fallback)
    small = io.BytesIO()
    Image.fromarray(img[:48, :60]).save(small, format='JPEG', quality=90)
    assert np.array_equal(P.decode_frame(small.getvalue(), 32, dct=True), P.decode_frame(small.getvalue(), 32))     # 48 < 2 * 32
    png = io.BytesIO()
    Image.fromarray(img).save(png, format='PNG')
    assert np.array_equal(P.decode_frame(png.getvalue(), 64, dct=True), P.decode_frame(png.getvalue(), 64))
    with pytest.raises(ValueError, match='decode'):
        P.PushDataset(os.path.join(HERE, 'golden'), 1, train_val_split=1.0, decode='fast')


def test_integer_box_sums_equal_the_float_mean():
    """The round-5 fast path of resize_area for decoded JPEGs (uint8, integer ratio): two-stage integer sums and one division
    - bit-identical to the float32 mean it replaces, for power-of-two and other box sizes, wide boxes (uint32 sums) included."""
    rng = np.random.default_rng(6)
    for (h, w, oh, ow) in ((512, 512, 64, 64), (60, 90, 10, 30), (128, 128, 2, 2), (24, 40, 24, 8), (7, 5, 7, 5)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = a.astype(np.float32).reshape(oh, h // oh, ow, w // ow, 3).mean(axis=(1, 3), dtype=np.float32)
        got = P.resize_area(a, oh, ow)
        assert got.dtype == np.float32 and np.array_equal(got, want), (h, w, oh, ow)
        assert np.array_equal(P.resize_area(a.astype(np.float32), oh, ow), want)      # the float path, unchanged


def test_committed_fixture_decodes_to_pinned_values():
    """tests/golden/push_tiny.tfrecord (+ make_push_fixture.py) pins the on-disk format and the decode arithmetic."""
    exp = np.load(os.path.join(HERE, 'golden', 'push_tiny_expected.npz'))
    recs = list(P.read_records(os.path.join(HERE, 'golden', 'push_tiny.tfrecord'), verify_crc=True))
    assert len(recs) == 1
    img, act, state = P.decode_example(recs[0])
    assert np.abs(img - exp['images']).max() <= 2e-2      # libjpeg builds may differ by a level or two
    assert np.array_equal(act, exp['action']) and np.array_equal(state, exp['state'])
    with pytest.raises(KeyError):
        P.decode_example(P.serialize_example({'move/6/image/encoded': b'x'}))


@pytest.mark.gpu
@pytest.mark.parametrize('workers', ['thread', 'process'])
def test_training_loop_reads_tfrecords(tmp_path, workers):
    """train() end to end from a directory of push TFRecords (a few iterations, a logging one among them) on the GPU, the records
    decoded by worker threads / spawned worker processes; the returned Trainer's session is OPEN (round 5) and closes clean."""
    import torch
    from action_conditioned_gans_amd import train as T
    rng = np.random.default_rng(3)
    for k in range(2):
        make_shard(str(tmp_path / ('push_%02d.tfrecord' % k)), rng, 2)
    tr = T.train(str(tmp_path), None, None, None, None, True, 'bce', 'adam', True, batch_size=2, train_iter=4,
                 pretrain_iter=1, device='cuda:0', quiet=True, eval_every=2, log_every=2, data_workers=workers, data_threads=2)
    for v in tr.g_vars + tr.d_vars:
        assert torch.isfinite(tr.sess.get_value(v)).all(), v.name
    frames, _, _ = tr.test(np.zeros((2, 64, 64, 3), np.float32), np.zeros((2, 64, 64, 3), np.float32), np.zeros((2, 10), np.float32))
    assert np.isfinite(frames).all()                  # the session still runs
    tr.sess.close()                                   # device-side flags clean, transport torn down
    if workers == 'thread':
        # the loop announced its frames (data_frames='selected', the default): decoding every frame instead changes no weight
        want = {v.name: tr.sess.get_value(v).cpu().numpy() for v in tr.g_vars + tr.d_vars}
        tr2 = T.train(str(tmp_path), None, None, None, None, True, 'bce', 'adam', True, batch_size=2, train_iter=4, pretrain_iter=1,
                      device='cuda:0', quiet=True, eval_every=2, log_every=2, data_workers=workers, data_threads=2, data_frames='all')
        for v in tr2.g_vars + tr2.d_vars:
            assert np.array_equal(tr2.sess.get_value(v).cpu().numpy(), want[v.name]), v.name
        tr2.sess.close()
        # ... and neither does serving the records' later visits from the frame cache (4 records on disk, 5 batches of 2)
        tr4 = T.train(str(tmp_path), None, None, None, None, True, 'bce', 'adam', True, batch_size=2, train_iter=4, pretrain_iter=1,
                      device='cuda:0', quiet=True, eval_every=2, log_every=2, data_workers=workers, data_threads=2, data_cache_gb=0.25)
        for v in tr4.g_vars + tr4.d_vars:
            assert np.array_equal(tr4.sess.get_value(v).cpu().numpy(), want[v.name]), v.name
        tr4.sess.close()
    else:
        tr3 = T.train(str(tmp_path), None, None, None, None, True, 'bce', 'adam', True, batch_size=2, train_iter=3, pretrain_iter=1,
                      device='cuda:0', quiet=True, eval_every=0, log_every=2, data_workers=workers, data_threads=2, data_decode='dct')
        for v in tr3.g_vars + tr3.d_vars:
            assert torch.isfinite(tr3.sess.get_value(v)).all(), v.name
        tr3.sess.close()
