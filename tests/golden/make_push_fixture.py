"""Makes tests/golden/push_tiny.tfrecord + push_tiny_expected.npz: one push record (7 frames 512x640 JPEG + poses)
written by push_data.write_push_tfrecord and the values push_data.decode_example must return for it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from action_conditioned_gans_amd import push_data as P   # noqa: E402

rng = np.random.default_rng(42)
yy, xx = np.mgrid[0:P.ORIGINAL_HEIGHT, 0:P.ORIGINAL_WIDTH].astype(np.float32)
frames = []
for j in range(7):
    a, b, c = rng.uniform(0.002, 0.01, 3)
    img = np.stack([127 + 100 * np.sin(a * xx + b * yy), 127 + 100 * np.cos(b * xx), 127 + 100 * np.sin(c * yy + j)], -1)
    frames.append(np.clip(img, 0, 255).astype(np.uint8))
action = rng.standard_normal((7, 5)).astype(np.float32)
state = rng.standard_normal((7, 5)).astype(np.float32)
path = os.path.join(HERE, 'push_tiny.tfrecord')
P.write_push_tfrecord(path, [(np.stack(frames), action, state)], quality=60)
img, a, s = P.decode_example(next(P.read_records(path, verify_crc=True)))
np.savez_compressed(os.path.join(HERE, 'push_tiny_expected.npz'), images=img, action=a, state=s)
print(os.path.getsize(path), 'bytes')
