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

    A feeder thread pulls ``(payload, tag)`` items from the (sequential, deterministic) stream, submits each payload to the pool
    (``None``: nothing to decode - the consumer has all it needs in the tag) and parks the future in a queue of ``capacity``
    entries; ``get()`` takes the oldest, waits for it and returns ``(result, tag)``.  Order of delivery = order of the stream
    whatever the workers' timing; at most ``capacity`` decoded records are held; an exception in the stream or in a decode
    surfaces from ``get()``; ``close()`` stops the feeder, cancels what has not started and joins everything."""

    class _Done:
        @staticmethod
        def result():
            return None

        @staticmethod
        def cancel():
            return False

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
            for payload, tag in self._stream:
                fut = self._Done if payload is None else self._pool.submit(self._decode, payload)
                if self._stop.is_set() or not self._put((fut, tag)):
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
        return item[0].result(), item[1]

    def close(self):
        self._stop.set()
        while True:                         # unblock the feeder and drop what is queued
            try:
                item = self._q.get_nowait()
                if not isinstance(item, BaseException):
                    item[0].cancel()
            except queue.Empty:
                break
        self._feeder.join(timeout=10)
        self._pool.shutdown(wait=True, cancel_futures=True)


class SparseFrames:
    """The frames of an ANNOUNCED batch: only the frames the caller said it would read exist (``rows[b]`` = {frame index:
    [S, S, 3] float32}).  Stands in for the dense ``[B, 7, S, S, 3]`` array where the training loop uses it - ``frames[mask]`` with
    a boolean ``[B, 7]`` mask (train.py:231-232: one frame per record, in record order) gathers straight from the decoded frames;
    a frame that was not asked for reads as NaN.  ``np.asarray(frames)`` (and with it any numpy function) materialises the dense
    array, NaN where nothing was decoded - 11 MB at batch 32 that the loop has no use for."""

    def __init__(self, rows, seq_len, img_size):
        self.rows = rows
        self.shape = (len(rows), seq_len, img_size, img_size, COLOR_CHAN)
        self.dtype, self.ndim = np.dtype(np.float32), 5

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        out = np.full(self.shape, np.nan, np.float32)
        for b, row in enumerate(self.rows):
            for t, frame in row.items():
                out[b, t] = frame
        return out if dtype is None else out.astype(dtype)

    def __getitem__(self, index):
        mask = index if isinstance(index, np.ndarray) else None
        if mask is not None and mask.dtype == bool and mask.shape == self.shape[:2]:
            picked = []
            for b, t in zip(*np.nonzero(mask)):                   # row-major, as numpy orders a boolean selection
                frame = self.rows[b].get(int(t))
                picked.append(frame if frame is not None else np.full(self.shape[2:], np.nan, np.float32))
            return np.stack(picked) if picked else np.zeros((0,) + self.shape[2:], np.float32)
        return np.asarray(self)[index]


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
    threads).  A worker reads its record from the file itself (the feeder walks the 12-byte frame headers only; with
    ``verify_crc`` the feeder reads and checks the payloads).  The order of the batches is the order of the record stream -
    deterministic per (seed, rank, world_size) whatever the worker count or kind; ``num_threads=0`` decodes inside
    ``get_batch`` as before round 5.  The workers start with the first ``get_batch`` or ``announce``.  ``close()`` (or the
    context manager) stops them; a dataset that is garbage-collected closes itself.

    Decoding less (the loop is decode-bound: one 512x640 JPEG is 2-3 ms of a core, a batch of 32 records holds 224 of them, and
    a training step consumes 2 frames of each record):
      * ``announce(need)`` - the caller says, batches ahead, WHICH frames of a coming batch it will read (``need`` [B, 7] bool;
        one call per future ``get_batch``, in order).  Only those are decoded; ``get_batch`` then returns the frames as a
        ``SparseFrames`` (``frames[mask]`` works as on the array; a frame that was not asked for reads as NaN, so that such a
        read cannot go unnoticed).  The decoded frames are the same bits as without the announcement.  A batch the workers
        reached before its announcement is decoded in full.  ``train()`` draws its frame-pair selections ahead of time to do
        this (train._PairSelections).
      * ``cache_bytes`` > 0 - decoded frames (and the pose vectors) are kept, up to that many bytes, and a record that comes
        round again (the stream repeats the files every epoch; 60 000 iterations at batch 32 are ~37 epochs of the push
        training set) is served from memory: no file read, no worker task.  What is kept is what the decoder produced - same
        bits.  First come, first kept (no eviction: under a cyclic stream an LRU of less than the whole set never hits).
      * ``decode='dct'`` - opt-in, approximate: the 8x reduction inside libjpeg's inverse DCT (``decode_frame``).
    """

    def __init__(self, data_dir, batch_size, train_val_split=0.95, use_state=True, training=True, img_size=IMG_HEIGHT,
                 seed=7, rank=0, world_size=1, verify_crc=False, num_threads=None, capacity=None, workers='thread', decode='exact',
                 cache_bytes=0):
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
        import functools
        self._task = functools.partial(_decode_task, use_state=self.use_state, img_size=self.img_size, dct=decode == 'dct')    # picklable
        self._lock = threading.Lock()
        self._plan = {}                                         # batch number -> need [B, 7], announced and not yet reached
        self._announced = 0                                     # batches announced so far
        self._served = 0                                        # records drawn from the stream so far
        self._spans = {}                                        # path -> [(offset, length)] of its records (walked once)
        self.cache_bytes, self._cached_bytes = int(cache_bytes), 0
        self._cache = {}                                        # (file, record number) -> [{frame: array}, action, state]
        self.cache_hits = self.cache_misses = 0                 # frames served from memory / decoded
        self._stream = self._items()
        self._prefetch, self._closed = None, False

    def _records(self):
        """The record stream of this rank: ((file number, record number), record) with the record as its payload bytes
        (``verify_crc``) or as a (path, offset, length) span that whoever decodes it reads itself - same files, same order."""
        n = 0
        while True:
            seen = False
            for k in self.rng.permutation(len(self.files)):
                path = self.files[k]
                if self.verify_crc:
                    recs = read_records(path, True)
                else:
                    if path not in self._spans:
                        self._spans[path] = list(record_spans(path))
                    recs = ((path,) + span for span in self._spans[path])
                for j, rec in enumerate(recs):
                    seen = True
                    if n % self.world_size == self.rank:
                        yield (int(k), j), rec
                    n += 1
            if not seen:
                raise RuntimeError('the data files hold no records')

    def _items(self):
        """(payload, tag) in stream order for the prefetcher: payload = (record, frames to decode) or None when the cache holds
        every frame that is needed; tag = (key, frames needed).  The need row is looked up when the record is DRAWN (by the
        feeder thread, up to ``capacity`` records ahead of the consumer) - a batch announced later than that is decoded in full."""
        everything = tuple(range(self.seq_len))
        for key, rec in self._records():
            with self._lock:
                batch, row = divmod(self._served, self.batch_size)
                need = self._plan.get(batch)
                if need is not None and row == self.batch_size - 1:
                    del self._plan[batch]
                self._served += 1
                entry = self._cache.get(key) if self.cache_bytes else None
                have = set(entry[0]) if entry is not None else ()
            wanted = everything if need is None else tuple(j for j in everything if need[row][j])
            missing = tuple(j for j in wanted if j not in have)
            if entry is not None and not missing:
                yield None, (key, wanted)
            else:
                yield (rec, tuple(j in missing for j in everything)), (key, wanted)

    def announce(self, need):
        """The frames of the next not-yet-announced batch that will be read: ``need`` [B, 7] bool.  One call per future
        ``get_batch``, in the same order; call it a few batches ahead (the workers run up to ``capacity`` records ahead)."""
        need = np.asarray(need, bool)
        if need.shape != (self.batch_size, self.seq_len):
            raise ValueError('announce: expected a [%d, %d] boolean array, got %s' % (self.batch_size, self.seq_len, need.shape))
        with self._lock:
            if self._announced * self.batch_size >= self._served:      # (a batch the workers already started on is past announcing)
                self._plan[self._announced] = need.copy()
            self._announced += 1
        self._start()

    def _start(self):
        if self._closed:
            raise RuntimeError('the dataset is closed')
        if self._prefetch is None and self.num_threads > 0:
            self._prefetch = _Prefetcher(self._stream, self._task, self.num_threads, self.capacity, processes=self.workers == 'process')

    def _next_record(self):
        """-> ({frame: [S, S, 3]} of the frames needed, action [7, 5], state [7, 5]) of the next record of the stream."""
        if self._prefetch is not None:
            decoded, (key, wanted) = self._prefetch.get()
        else:
            payload, (key, wanted) = next(self._stream)
            decoded = None if payload is None else self._task(payload)
        with self._lock:
            entry = self._cache.get(key) if self.cache_bytes else None
        fresh = {} if decoded is None else {j: decoded[1][i] for i, j in enumerate(decoded[0])}
        # (a frame both decoded for this visit and kept meanwhile by an earlier visit - the workers run ahead - counts once)
        frames = {j: entry[0][j] for j in wanted if j in entry[0] and j not in fresh} if entry is not None else {}
        self.cache_hits += len(frames)
        self.cache_misses += len(fresh)
        if decoded is None:
            return frames, entry[1], entry[2]
        acts, states = decoded[2], decoded[3]
        frames.update(fresh)
        if self.cache_bytes:
            size = sum(f.nbytes for j, f in fresh.items() if entry is None or j not in entry[0]) + (acts.nbytes + states.nbytes if entry is None else 0)
            with self._lock:
                if self._cached_bytes + size <= self.cache_bytes:       # first come, first kept
                    if entry is None:
                        entry = self._cache.setdefault(key, [{}, acts, states])
                    for j, f in fresh.items():
                        entry[0].setdefault(j, f)
                    self._cached_bytes += size
        return frames, acts, states

    def get_batch(self):
        """-> (frames, frames, action||state [B,T,10], state [B,T,5]), the tuple the training loop consumes.  ``frames``: the
        ``[B, 7, S, S, 3]`` array, or a ``SparseFrames`` when only announced frames were decoded."""
        self._start()
        rows, acts, states = zip(*[self._next_record() for _ in range(self.batch_size)])
        if all(len(r) == self.seq_len for r in rows):
            img = np.stack([np.stack([r[j] for j in range(self.seq_len)]) for r in rows])
        else:
            img = SparseFrames(list(rows), self.seq_len, self.img_size)
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
