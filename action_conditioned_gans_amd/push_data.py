"""The on-disk format on the input side of the hot path: the BAIR/Google "push" TFRecords, read without TensorFlow.

Restates ``build_tfrecord_input`` (reference ops.py:140-223): every record is a ``tf.train.Example`` holding, for the
frames 6, 8, ..., 18 of a push (7 frames), a 512x640 JPEG (``move/<i>/image/encoded``), the commanded pose
(``move/<i>/commanded_pose/vec_pitch_yaw``, 5 floats) and the end-effector pose
(``move/<i>/endeffector/vec_pitch_yaw``, 5 floats).  A frame is decoded, centre-cropped to 512x512
(``resize_image_with_crop_or_pad``), area-resized to 64x64 (``tf.image.resize_area``; exactly the mean of 8x8 boxes at
this ratio), and mapped to [-1, 1] by ``x / 127.5 - 1``.  A batch is ``images [B, 7, 64, 64, 3]`` and
``action_state [B, 7, 10]`` (action 0:5, state 5:10), the two arrays ``get_batch`` (ops.py:15-17) hands the Trainer.

Pieces (all host code; numpy + PIL for the JPEG):
  * TFRecord framing: ``uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)``;
  * a minimal protobuf reader / writer for ``Example { Features { map<string, Feature> } }`` with ``BytesList`` (1),
    ``FloatList`` (2, packed or not) and ``Int64List`` (3);
  * ``resize_area`` for arbitrary ratios (box filter with fractional overlaps, as TF defines it);
  * ``PushDataset`` - file split by ``train_val_split`` as the reference does, shuffled record stream, batches, decoded by a
    pool of worker threads into a bounded prefetch queue (the reference: ``tf.train.batch(num_threads=batch_size,
    capacity=500 * batch_size)``, ops.py:209-213) so that decoding overlaps the training step; ``announce`` - the caller names,
    batches ahead, the frames it will read and only those are decoded (same bits); ``decode='dct'`` - opt-in approximate
    reduction inside libjpeg's inverse DCT;
  * ``write_push_tfrecord`` - the inverse, used by the tests and for making small synthetic shards.

What is pinned and what is not.  The framing (RFC 3720 CRC vectors), the protobuf wire format, the crop offsets and the
area filter are pinned by known answers worked out by hand from TensorFlow's published definitions
(tests/test_push_data.py: ``*_known_answers_by_hand``; 3 -> 2, 5 -> 2, 5 -> 3, 7 -> 4, 2 -> 3, two axes, odd crops and pads).
The JPEG decoder is NOT pinned against TensorFlow's: ``tf.image.decode_jpeg(channels=3)`` (ops.py:184) runs libjpeg with
``dct_method=''`` (the library default, the accurate integer IDCT ``JDCT_ISLOW``) and ``fancy_upscaling=True``; PIL runs
libjpeg-turbo with the same two defaults, whose ISLOW IDCT is bit-exact with libjpeg's, while the chroma ("fancy") upsampling
of different libjpeg generations may differ by one level per pixel.  After the 8x8 box mean and ``/ 127.5`` that is at most
0.008 and typically below 1e-3 in the [-1, 1] frames - input noise far below JPEG's own quantisation, but not a bit-exact
restatement, and there is no TensorFlow here to measure it against.
"""
import glob
import io
import os
import queue
import struct
import threading
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

ORIGINAL_WIDTH, ORIGINAL_HEIGHT, COLOR_CHAN = 640, 512, 3      # ops.py:129-131
IMG_WIDTH = IMG_HEIGHT = 64                                    # ops.py:134-135
STATE_DIM = 5                                                  # ops.py:138
FRAME_IDS = tuple(range(6, 20, 2))                             # ops.py:171


# ---- CRC32C (Castagnoli), masked as TFRecord does -------------------------------------------------------------
def _crc_table():
    poly, table = 0x82F63B78, []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_TABLE = _crc_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- TFRecord framing -----------------------------------------------------------------------------------------
def read_records(path, verify_crc=False):
    """Yield the payload of every record of a TFRecord file.  ``verify_crc`` checks both checksums (pure Python: slow
    on multi-megabyte records, meant for tests and for diagnosing a damaged shard)."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError('%s: truncated record header' % path)
            (length,), (lcrc,) = struct.unpack('<Q', head[:8]), struct.unpack('<I', head[8:])
            if verify_crc and masked_crc32c(head[:8]) != lcrc:
                raise IOError('%s: corrupted record length' % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError('%s: truncated record' % path)
            if verify_crc and masked_crc32c(data) != struct.unpack('<I', tail)[0]:
                raise IOError('%s: corrupted record data' % path)
            yield data


def record_spans(path):
    """(offset, length) of every record's payload, from the framing alone (12-byte headers; the payloads are skipped, not read):
    what a worker process needs to fetch its record itself."""
    size = os.path.getsize(path)
    with open(path, 'rb') as f:
        pos = 0
        while pos < size:
            head = f.read(8)
            if len(head) < 8:
                raise IOError('%s: truncated record header' % path)
            (length,) = struct.unpack('<Q', head)
            if pos + 12 + length + 4 > size:
                raise IOError('%s: truncated record' % path)
            yield pos + 12, length
            pos += 12 + length + 4
            f.seek(pos)


def write_records(path, payloads):
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', masked_crc32c(head)) + data + struct.pack('<I', masked_crc32c(data)))


# ---- protobuf (only what tf.train.Example needs) ----------------------------------------------------------------
def _varint(buf, pos):
    shift = value = 0
    while True:
        b = buf[pos]
        pos += 1
        value |= (b & 0x7F) << shift
        if not b & 0x80:
            return value, pos
        shift += 7
        if shift > 63:
            raise ValueError('malformed varint')


def _fields(buf):
    """Yield (field number, wire type, value) of one message; length-delimited values as memoryviews."""
    buf = memoryview(buf)
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        if pos > end:
            raise ValueError('truncated protobuf message')
        yield num, wt, val


def _parse_feature(buf):
    for num, wt, val in _fields(buf):
        if num == 1:                                   # BytesList { repeated bytes value = 1 }
            return [bytes(v) for n, w, v in _fields(val) if n == 1]
        if num == 2:                                   # FloatList { repeated float value = 1 [packed] }
            out = []
            for n, w, v in _fields(val):
                if n == 1:
                    out.append(np.frombuffer(bytes(v), '<f4') if w == 2 else np.frombuffer(v, '<f4'))
            return np.concatenate(out) if out else np.zeros(0, np.float32)
        if num == 3:                                   # Int64List { repeated int64 value = 1 [packed] }
            out = []
            for n, w, v in _fields(val):
                if n != 1:
                    continue
                if w == 2:
                    p, v = 0, bytes(v)
                    while p < len(v):
                        x, p = _varint(v, p)
                        out.append(x - (1 << 64) if x >> 63 else x)
                else:
                    out.append(v - (1 << 64) if v >> 63 else v)
            return np.array(out, np.int64)
    return None


def parse_example(buf, keys=None):
    """tf.train.Example bytes -> {feature name: list of bytes | float32 array | int64 array}; ``keys`` limits the
    features that are decoded (the push records carry 30 frames and more, the reference reads 7)."""
    out = {}
    for num, wt, features in _fields(buf):
        if num != 1:
            continue
        for fnum, fwt, entry in _fields(features):     # map<string, Feature> feature = 1
            if fnum != 1:
                continue
            name = value = None
            for enum_, ewt, ev in _fields(entry):
                if enum_ == 1:
                    name = bytes(ev).decode('utf-8')
                elif enum_ == 2:
                    value = ev
            if name is not None and value is not None and (keys is None or name in keys):
                out[name] = _parse_feature(value)
    return out


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
    """{name: bytes | sequence of floats} -> tf.train.Example bytes (floats packed, as TensorFlow writes them)."""
    entries = b''
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            feat = _ld(1, _ld(1, bytes(v)))
        else:
            feat = _ld(2, _ld(1, np.asarray(v, '<f4').tobytes()))
        entries += _ld(1, _ld(1, name.encode('utf-8')) + _ld(2, feat))
    return _ld(1, entries)


# ---- image ops ----------------------------------------------------------------------------------------------------
def crop_or_pad_center(img, th, tw):
    """tf.image.resize_image_with_crop_or_pad: centred crop and / or zero pad to th x tw."""
    h, w = img.shape[:2]
    if h > th:
        o = (h - th) // 2
        img = img[o:o + th]
    if w > tw:
        o = (w - tw) // 2
        img = img[:, o:o + tw]
    h, w = img.shape[:2]
    if h < th or w < tw:
        out = np.zeros((th, tw) + img.shape[2:], img.dtype)
        oh, ow = (th - h) // 2, (tw - w) // 2
        out[oh:oh + h, ow:ow + w] = img
        img = out
    return img


def _area_weights(n_in, n_out):
    """[n_out, n_in] box-filter weights of tf.image.resize_area along one axis (rows sum to 1)."""
    scale = n_in / float(n_out)
    w = np.zeros((n_out, n_in), np.float64)
    for o in range(n_out):
        lo, hi = o * scale, (o + 1) * scale
        for i in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_in)):
            w[o, i] = max(0.0, min(hi, i + 1) - max(lo, i))
        w[o] /= scale
    return w


def resize_area(img, oh, ow):
    """tf.image.resize_area of an HxWxC image to oh x ow, float32."""
    h, w = img.shape[:2]
    if h % oh == 0 and w % ow == 0:                    # integer ratio: plain box means (the push pipeline: 512 -> 64)
        fh, fw = h // oh, w // ow
        if img.dtype == np.uint8 and img.ndim == 3:
            # decoded JPEGs: integer box sums in two contiguous stages (rows of a box, then its columns), one division at the
            # end - 0.8 ms for 512 x 512 x 3 where the float32 mean over the strided axes took 6 ms, and bit-identical to it
            # (sums of at most fh * fw bytes are exact in float32 either way; tests/test_push_data.py)
            acc = np.uint16 if fh * fw * 255 < 65536 else np.uint32
            s1 = np.add.reduce(img.reshape(oh, fh, w * img.shape[2]), axis=1, dtype=acc)
            s2 = np.add.reduce(s1.reshape(oh, ow, fw, img.shape[2]), axis=2, dtype=acc)
            return s2.astype(np.float32) / np.float32(fh * fw)
        return img.astype(np.float32).reshape(oh, fh, ow, fw, -1).mean(axis=(1, 3), dtype=np.float32)
    x = img.astype(np.float32)
    wh, ww = _area_weights(h, oh).astype(np.float32), _area_weights(w, ow).astype(np.float32)
    rows = (wh @ x.reshape(h, -1)).reshape(oh, w, -1)          # two small matrix products (one unoptimised three-operand einsum
    return np.matmul(ww, rows)                                  # of this took minutes at 512 -> 400)


def decode_frame(jpeg_bytes, img_size=IMG_HEIGHT, dct=False):
    """ops.py:184-196: decode (3 channels), centre-crop to the short side, area-resize, scale to [-1, 1].

    ``dct`` (opt-in, NOT the reference's arithmetic): let libjpeg reduce the frame by 2, 4 or 8 inside the inverse DCT (PIL's
    ``draft``: the largest of those that still leaves at least ``img_size`` on the short side), then crop and box-resize what is
    left.  At the push ratio (512 -> 64) that is the 1x1 IDCT - the mean of every 8x8 luma block straight from its DC term, chroma
    from the 2x2 IDCT of its subsampled block - i.e. the same box mean taken BEFORE instead of after rounding, clamping and
    chroma upsampling: 3x cheaper (the Huffman pass is what remains), and within 2 levels of 255 of the exact frame, 0.5 level on
    average (tests/test_push_data.py measures it; JPEG quantisation at quality 90 moves pixels by more).  A file that is not a
    JPEG, or a ratio libjpeg cannot serve, silently takes the exact path."""
    from PIL import Image
    im = Image.open(io.BytesIO(jpeg_bytes))
    if dct and im.format == 'JPEG':
        w, h = im.size
        short = min(w, h)
        if short >= 2 * img_size:
            # request the size that keeps the aspect: draft() picks the largest scale s in {2, 4, 8} with w/s, h/s >= the request
            im.draft('RGB', (-(-w * img_size // short), -(-h * img_size // short)))
    if im.mode != 'RGB':
        im = im.convert('RGB')                      # decode_jpeg(channels=3); an RGB file needs no second copy
    w, h = im.size
    crop = min(w, h)
    if w > crop or h > crop:                        # the centred crop of crop_or_pad_center (same offsets), taken before the pixels
        left, top = (w - crop) // 2, (h - crop) // 2    # become an array: the array is then contiguous and a fifth smaller
        im = im.crop((left, top, left + crop, top + crop))
    return resize_area(np.asarray(im), img_size, img_size) / np.float32(255.0 / 2.0) - np.float32(1.0)


def decode_selected(buf, use_state=True, img_size=IMG_HEIGHT, need=None, dct=False):
    """One record -> (frame indices, images [n, S, S, 3] of those frames, action [7, 5], state [7, 5]).  ``need``: 7 booleans, the
    frames to decode (None: all) - the JPEGs of the others are skipped, their pose vectors are still read (they are 10 floats)."""
    keys = set()
    n = len(FRAME_IDS)
    if need is None:
        which = list(range(n))
    else:
        if len(need) != n:
            raise ValueError('need: expected %d booleans, one per frame' % n)
        which = [j for j in range(n) if need[j]]
    for j, i in enumerate(FRAME_IDS):
        if j in which:
            keys.add('move/%d/image/encoded' % i)
        if use_state:
            keys.add('move/%d/commanded_pose/vec_pitch_yaw' % i)
            keys.add('move/%d/endeffector/vec_pitch_yaw' % i)
    feats = parse_example(buf, keys)
    missing = sorted(keys - set(feats))
    if missing:
        raise KeyError('record lacks features %s' % missing[:3])
    imgs, acts, states = [], [], []
    for j, i in enumerate(FRAME_IDS):
        if j in which:
            enc = feats['move/%d/image/encoded' % i]
            if len(enc) != 1:
                raise ValueError('move/%d/image/encoded: expected one JPEG, got %d' % (i, len(enc)))
            imgs.append(decode_frame(enc[0], img_size, dct))
        if use_state:
            a, s = feats['move/%d/commanded_pose/vec_pitch_yaw' % i], feats['move/%d/endeffector/vec_pitch_yaw' % i]
            if a.shape != (STATE_DIM,) or s.shape != (STATE_DIM,):
                raise ValueError('move/%d: pose vectors must have %d floats' % (i, STATE_DIM))
            acts.append(a)
            states.append(s)
    acts = np.stack(acts).astype(np.float32) if use_state else np.zeros((n, STATE_DIM), np.float32)
    states = np.stack(states).astype(np.float32) if use_state else np.zeros((n, STATE_DIM), np.float32)
    imgs = np.stack(imgs).astype(np.float32) if imgs else np.zeros((0, img_size, img_size, COLOR_CHAN), np.float32)
    return tuple(which), imgs, acts, states


def decode_example(buf, use_state=True, img_size=IMG_HEIGHT, dct=False):
    """One record -> (images [7, S, S, 3], action [7, 5], state [7, 5]) (zeros for the vectors without use_state)."""
    return decode_selected(buf, use_state, img_size, None, dct)[1:]


def _decode_task(item, use_state=True, img_size=IMG_HEIGHT, dct=False):
    """Worker task: ``item`` = (record, need); the record is its payload bytes or a (path, offset, length) span that the worker
    (a process: the payload never passes through a pipe) reads itself."""
    rec, need = item
    if isinstance(rec, tuple):
        path, offset, length = rec
        with open(path, 'rb') as f:
            f.seek(offset)
            rec = f.read(length)
        if len(rec) < length:
            raise IOError('%s: truncated record' % path)
    return decode_selected(rec, use_state, img_size, need, dct)


class _Prefetcher:
    """Records of one stream decoded by ``num_threads`` workers, handed out IN STREAM ORDER from a bounded queue.

    A feeder thread pulls records from the (sequential, deterministic) stream, submits each to the pool and parks the future
    in a queue of ``capacity`` entries; ``get()`` takes the oldest and waits for it.  Order of delivery = order of the
    stream whatever the workers' timing; at most ``capacity`` decoded records are held; an exception in the stream or in a
    decode surfaces from ``get()``; ``close()`` stops the feeder, cancels what has not started and joins everything."""

    def __init__(self, stream, decode, num_threads, capacity, processes=False):
        self._stream, self._decode = stream, decode
        self._q = queue.Queue(maxsize=max(int(capacity), 1))
        if processes:
            # worker PROCESSES: started by 'spawn' (never a fork of a process that may hold a GPU context), each imports this
            # module only (numpy + PIL; the package's __init__ pulls in nothing else); records go over as bytes, decoded
            # sequences come back as arrays
            import multiprocessing
            self._pool = ProcessPoolExecutor(max_workers=max(int(num_threads), 1), mp_context=multiprocessing.get_context('spawn'))
        else:
            self._pool = ThreadPoolExecutor(max_workers=max(int(num_threads), 1), thread_name_prefix='push-decode')
        self._stop = threading.Event()
        self._feeder = threading.Thread(target=self._feed, name='push-feeder', daemon=True)
        self._feeder.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def _feed(self):
        try:
            for rec in self._stream:
                if self._stop.is_set() or not self._put(self._pool.submit(self._decode, rec)):
                    return
            self._put(StopIteration('the record stream ended'))
        except BaseException as e:          # a damaged shard, a decode that could not be submitted: hand it to the consumer
            self._put(e)

    def get(self):
        if self._stop.is_set():
            raise RuntimeError('the dataset is closed')
        item = self._q.get()
        if isinstance(item, BaseException):
            self._q.put(item)               # every later call reports it too
            raise item
        return item.result()

    def close(self):
        self._stop.set()
        while True:                         # unblock the feeder and drop what is queued
            try:
                item = self._q.get_nowait()
                if not isinstance(item, BaseException):
                    item.cancel()
            except queue.Empty:
                break
        self._feeder.join(timeout=10)
        self._pool.shutdown(wait=True, cancel_futures=True)


class PushDataset:
    """``build_tfrecord_input`` + ``get_batch`` (ops.py:140-223, 15-17) as an iterator of numpy batches.

    Files are split by ``train_val_split`` exactly as the reference does (first ``floor(split * n)`` files train, the
    rest validation; sorted here so the split is reproducible); records are streamed file by file in a shuffled file
    order (``string_input_producer(shuffle=True)``), forever.  ``rank`` / ``world_size`` give every data-parallel
    process its own interleaved share of the record stream.

    ``num_threads`` decode workers (default: ``batch_size`` as ops.py:212, at most 16 and at most the CPUs this process may
    use) fill a queue of ``capacity`` decoded records (default 4 batches; the reference's ``500 * batch_size`` records would be
    5.5 GB of decoded float32 frames at batch 32) ahead of ``get_batch``: JPEG decoding and the box resize run while the
    training step does.  ``workers='thread'`` (default): PIL's decoder and numpy's reductions release the GIL, the protobuf
    walk and the array plumbing do not - measured on the GPU box: 3.6x one thread at 8 threads, nothing beyond;
    ``workers='process'``: spawned worker processes, no shared interpreter lock (the reference's queue runners are C++
    threads).  The order of the batches is the order of the record stream - deterministic per (seed, rank, world_size)
    whatever the worker count or kind; ``num_threads=0`` decodes inside ``get_batch`` as before round 5.  The workers start
    with the first ``get_batch`` or ``announce``.  ``close()`` (or the context manager) stops them; a dataset that is
    garbage-collected closes itself.

    Decoding less (the loop is decode-bound: one 512x640 JPEG is 2-3 ms of a core, a batch of 32 records holds 224 of them, and
    a training step consumes 2 frames of each record):
      * ``announce(need)`` - the caller says, batches ahead, WHICH frames of a coming batch it will read (``need`` [B, 7] bool;
        one call per future ``get_batch``, in order).  Only those are decoded; the others come back as NaN so that a read of a
        frame that was not asked for cannot go unnoticed.  The decoded frames are the same bits as without the announcement.
        A batch the workers reached before its announcement is decoded in full.  ``train()`` draws its frame-pair selections
        ahead of time to do this (train._PairSelections).
      * ``decode='dct'`` - opt-in, approximate: the 8x reduction inside libjpeg's inverse DCT (``decode_frame``).
    """

    def __init__(self, data_dir, batch_size, train_val_split=0.95, use_state=True, training=True, img_size=IMG_HEIGHT,
                 seed=7, rank=0, world_size=1, verify_crc=False, num_threads=None, capacity=None, workers='thread', decode='exact'):
        files = sorted(glob.glob(os.path.join(data_dir, '*')))
        if not files:
            raise RuntimeError('No data files found.')                          # ops.py:159
        index = int(np.floor(train_val_split * len(files)))
        self.files = files[:index] if training else files[index:]
        if not self.files:
            raise RuntimeError('No data files found for the %s split.' % ('training' if training else 'validation'))
        self.batch_size, self.use_state, self.img_size = batch_size, use_state, img_size
        self.rng = np.random.default_rng(seed)
        self.rank, self.world_size, self.verify_crc = rank, world_size, verify_crc
        self.seq_len = len(FRAME_IDS)
        if num_threads is None:
            try:
                cpus = len(os.sched_getaffinity(0))
            except AttributeError:
                cpus = os.cpu_count() or 1
            num_threads = max(1, min(batch_size, cpus, 16))
        if workers not in ('thread', 'process'):
            raise ValueError("workers must be 'thread' or 'process'")
        if decode not in ('exact', 'dct'):
            raise ValueError("decode must be 'exact' or 'dct'")
        self.num_threads, self.workers, self.decode = int(num_threads), workers, decode
        self.capacity = int(capacity) if capacity else 4 * batch_size
        # worker processes fetch their record from the file themselves (the feeder walks the 12-byte frame headers only): the
        # payload bytes - 0.5-2 MB per push record - never pass through this process or a pipe
        self._spans = self.num_threads > 0 and workers == 'process' and not verify_crc
        import functools
        self._task = functools.partial(_decode_task, use_state=self.use_state, img_size=self.img_size, dct=decode == 'dct')    # picklable
        self._plan, self._plan_lock = {}, threading.Lock()      # batch number -> need [B, 7], announced and not yet reached
        self._announced = 0                                     # batches announced so far
        self._served = 0                                        # records handed to a decoder so far
        self._stream = self._items()
        self._prefetch, self._closed = None, False

    def _records(self, spans=False):
        """The record stream of this rank: payload bytes, or (``spans``) (path, offset, length) triples for workers that read
        their record themselves - same files, same order."""
        n = 0
        while True:
            seen = False
            for k in self.rng.permutation(len(self.files)):
                path = self.files[k]
                for rec in (record_spans(path) if spans else read_records(path, self.verify_crc)):
                    seen = True
                    if n % self.world_size == self.rank:
                        yield (path,) + rec if spans else rec
                    n += 1
            if not seen:
                raise RuntimeError('the data files hold no records')

    def _items(self):
        """(record, need) in stream order: the need row is looked up when the record is DRAWN (by the feeder thread, up to
        ``capacity`` records ahead of the consumer) - a batch announced later than that is decoded in full."""
        for rec in self._records(self._spans):
            with self._plan_lock:
                batch, row = divmod(self._served, self.batch_size)
                need = self._plan.get(batch)
                if need is not None and row == self.batch_size - 1:
                    del self._plan[batch]
                self._served += 1
            yield rec, (None if need is None else tuple(bool(v) for v in need[row]))

    def announce(self, need):
        """The frames of the next not-yet-announced batch that will be read: ``need`` [B, 7] bool.  One call per future
        ``get_batch``, in the same order; call it a few batches ahead (the workers run up to ``capacity`` records ahead)."""
        need = np.asarray(need, bool)
        if need.shape != (self.batch_size, self.seq_len):
            raise ValueError('announce: expected a [%d, %d] boolean array, got %s' % (self.batch_size, self.seq_len, need.shape))
        with self._plan_lock:
            if self._announced * self.batch_size >= self._served:      # (a batch the workers already started on is past announcing)
                self._plan[self._announced] = need.copy()
            self._announced += 1
        self._start()

    def _start(self):
        if self._closed:
            raise RuntimeError('the dataset is closed')
        if self._prefetch is None and self.num_threads > 0:
            self._prefetch = _Prefetcher(self._stream, self._task, self.num_threads, self.capacity, processes=self.workers == 'process')

    def get_batch(self):
        """-> (frames, frames, action||state [B,T,10], state [B,T,5]), the tuple the training loop consumes."""
        self._start()
        if self._prefetch is not None:
            decoded = [self._prefetch.get() for _ in range(self.batch_size)]
        else:
            decoded = [self._task(next(self._stream)) for _ in range(self.batch_size)]
        which, imgs, acts, states = zip(*decoded)
        full = tuple(range(self.seq_len))
        if all(w == full for w in which):
            img = np.stack(imgs)
        else:                                     # announced batches: the frames nobody asked for are NaN
            img = np.full((self.batch_size, self.seq_len, self.img_size, self.img_size, COLOR_CHAN), np.nan, np.float32)
            for b, (w, im) in enumerate(zip(which, imgs)):
                if w:
                    img[b, list(w)] = im
        action_state = np.concatenate([np.stack(acts), np.stack(states)], axis=2)
        return img, img, action_state, action_state[:, :, STATE_DIM:].copy()

    def close(self):
        self._closed = True
        if self._prefetch is not None:
            self._prefetch.close()
            self._prefetch = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_push_tfrecord(path, sequences, quality=95):
    """Write records in the push format.  ``sequences``: iterable of (frames uint8 [7, H, W, 3], action [7, 5],
    state [7, 5]).  Used by the tests and to make small synthetic shards."""
    from PIL import Image
    payloads = []
    for frames, action, state in sequences:
        feats = {}
        for j, i in enumerate(FRAME_IDS):
            b = io.BytesIO()
            Image.fromarray(np.asarray(frames[j], np.uint8)).save(b, format='JPEG', quality=quality)
            feats['move/%d/image/encoded' % i] = b.getvalue()
            feats['move/%d/commanded_pose/vec_pitch_yaw' % i] = np.asarray(action[j], np.float32)
            feats['move/%d/endeffector/vec_pitch_yaw' % i] = np.asarray(state[j], np.float32)
        payloads.append(serialize_example(feats))
    write_records(path, payloads)
