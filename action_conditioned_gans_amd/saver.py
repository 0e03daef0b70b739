"""Checkpoint save / resume (the reference uses ``tf.train.Saver().save`` every 100 iterations, train.py:215,274,
and ``saver.restore(latest_checkpoint)`` in test.py:29-30).

A checkpoint is one ``.npz``: every variable under its slim name (``g/conv1/weights`` ... - SURVEY Appendix C, so
a TF-name-compatible export is a rename-free dump) plus every optimizer slot and step counter under its graph name.
As ``tf.train.Saver()`` does by default, only the ``max_to_keep`` = 5 most recent checkpoints of a Saver stay on disk
(a checkpoint of config 2 is ~90 MB; the reference's cadence makes 600 of them in a run).
"""
import glob
import os
import queue
import re
import threading

import numpy as np
import torch


class Saver:
    def __init__(self, graph=None, max_to_keep=5):
        from . import graph as G
        self.graph = graph or G.get_default_graph()
        self.max_to_keep = max_to_keep            # None or 0: keep everything
        self._kept = []                           # paths written by this Saver, oldest first
        self._jobs, self._writer, self._error = None, None, None

    def _tensors(self):
        items = {'var:' + n: v for n, v in self.graph.variables.items()}
        for s in self.graph.state:
            if s.name and not s.name.endswith(('/flat_grad', '/bf16_rm', '/bf16_tr')):     # gradients and derived copies are not state
                items['state:' + s.name] = s
        return items

    def _snapshot(self, sess):
        """{key: host array} of everything a checkpoint holds.  On a GPU the tensors are gathered into one buffer per dtype on
        the device and come over in ONE copy each (they are ~150 views of a few flat buffers: one synchronous copy apiece cost
        more than writing the file)."""
        items = self._tensors()
        bufs = {k: sess._materialize(t).detach() for k, t in items.items()}
        out = {}
        if not any(b.is_cuda for b in bufs.values()):
            return {k: b.cpu().numpy().copy() for k, b in bufs.items()}
        groups = {}
        for k, b in bufs.items():
            groups.setdefault(b.dtype, []).append(k)
        for dtype, keys in groups.items():
            flat = torch.cat([bufs[k].reshape(-1) for k in keys]).cpu().numpy()
            off = 0
            for k in keys:
                n = bufs[k].numel()
                out[k] = flat[off:off + n].reshape(tuple(bufs[k].shape))
                off += n
        return out

    @staticmethod
    def _write(path, out):
        tmp = path + '.tmp'
        with open(tmp, 'wb') as f:
            np.savez(f, **out)
        os.replace(tmp, path)                     # a reader (latest_checkpoint, restore) never sees half a file

    def _drain(self):
        while True:
            job = self._jobs.get()
            try:
                if job is None:
                    return
                kind, path, out = job
                if kind == 'write':
                    self._write(path, out)
                elif os.path.exists(path):
                    os.remove(path)
            except BaseException as e:            # reported by the next save() / wait()
                self._error = e
            finally:
                self._jobs.task_done()

    def _submit(self, job, background):
        if background:
            if self._writer is None:
                self._jobs = queue.Queue(maxsize=2)          # at most two snapshots (~200 MB) waiting for the disk
                self._writer = threading.Thread(target=self._drain, name='acg-checkpoint-writer', daemon=True)
                self._writer.start()
            self._jobs.put(job)
        else:
            self.wait()                                      # keep the order of writes and removals
            kind, path, out = job
            if kind == 'write':
                self._write(path, out)
            elif os.path.exists(path):
                os.remove(path)

    def save(self, sess, save_path, background=False):
        """Writes ``save_path + '.npz'``; returns that path.  ``background``: the state is copied to the host now, the file is
        written by a writer thread (in order; ``wait()`` returns when everything is on disk; an error of the writer is raised by
        the next ``save`` / ``wait``) - the training loop goes on meanwhile."""
        if self._error is not None:
            err, self._error = self._error, None
            raise err
        out = self._snapshot(sess)
        os.makedirs(os.path.dirname(os.path.abspath(save_path)) or '.', exist_ok=True)
        path = save_path + '.npz'
        self._submit(('write', path, out), background)
        if path in self._kept:
            self._kept.remove(path)
        self._kept.append(path)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            self._submit(('remove', self._kept.pop(0), None), background)
        return path

    def wait(self):
        """Everything handed to the writer thread is on disk when this returns."""
        if self._jobs is not None:
            self._jobs.join()
            self._jobs.put(None)                  # the writer thread ends; the next background save starts a new one
            self._writer.join()
            self._jobs = self._writer = None
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def restore(self, sess, save_path):
        self.wait()
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
