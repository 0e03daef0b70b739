"""Host-side helpers of the training loop (util.py:10-16 of the reference)."""
import numpy as np


def build_all_mask(num_frame):
    """One-hot boolean rows selecting frame t for t in [0, num_frame-2]; ``np.roll(mask, 1, axis=1)``
    then selects frame t+1 (train.py:231-232)."""
    masks = np.zeros((num_frame - 1, num_frame), dtype=bool)
    for t in range(num_frame - 1):
        masks[t, t] = True
    return masks
