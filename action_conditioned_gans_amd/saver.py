"""Checkpoint save / resume (the reference uses ``tf.train.Saver().save`` every 100 iterations, train.py:215,274,
and ``saver.restore(latest_checkpoint)`` in test.py:29-30).

A checkpoint is one ``.npz``: every variable under its slim name (``g/conv1/weights`` ... - SURVEY Appendix C, so
a TF-name-compatible export is a rename-free dump) plus every optimizer slot and step counter under its graph name.
"""
import glob
import os
import re

import numpy as np
import torch


class Saver:
    def __init__(self, graph=None):
        from . import graph as G
        self.graph = graph or G.get_default_graph()

    def _tensors(self):
        items = {'var:' + n: v for n, v in self.graph.variables.items()}
        for s in self.graph.state:
            if s.name and not s.name.endswith(('/flat_grad', '/bf16_rm', '/bf16_tr')):     # gradients and derived copies are not state
                items['state:' + s.name] = s
        return items

    def save(self, sess, save_path):
        """Writes ``save_path + '.npz'``; returns that path."""
        out = {}
        for key, t in self._tensors().items():
            out[key] = sess._materialize(t).detach().cpu().numpy()
        os.makedirs(os.path.dirname(os.path.abspath(save_path)) or '.', exist_ok=True)
        path = save_path + '.npz'
        np.savez(path, **out)
        return path

    def restore(self, sess, save_path):
        path = save_path if save_path.endswith('.npz') else save_path + '.npz'
        data = np.load(path)
        items = self._tensors()
        missing = [k for k in items if k.startswith('var:') and k not in data]
        if missing:
            raise ValueError('checkpoint %s lacks variables: %s' % (path, ', '.join(missing)))
        for key, t in items.items():
            if key not in data:
                continue
            arr = data[key]
            if tuple(arr.shape) != t.shape:
                raise ValueError('checkpoint tensor %s has shape %s, graph expects %s' % (key, arr.shape, t.shape))
            sess._materialize(t).copy_(torch.from_numpy(arr).to(t.dtype))
        sess._weights_dirty = True          # bf16 sessions: the filter copies follow the restored master weights


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the ``model<N>`` prefix with the largest N (train.py:274 naming)."""
    best, best_n = None, -1
    for f in glob.glob(os.path.join(checkpoint_dir, '*.npz')):
        m = re.search(r'(\d+)\.npz$', f)
        n = int(m.group(1)) if m else 0
        if n > best_n:
            best, best_n = f[:-4], n
    return best
