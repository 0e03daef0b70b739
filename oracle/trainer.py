"""torch-CPU restatement of the reference Trainer step functions (train.py:27-155).

TEST INFRASTRUCTURE (see oracle/__init__.py) - parity unpinned.

One instance owns the parameter dict and three independent optimizer slot sets
(g_opt, g_pretrain_opt, d_opt - train.py:100-102).  Optimizer formulas are TensorFlow's
(SURVEY A.6), not PyTorch's.  Documented choices for reference defects: D1 (`==`),
D4 (state head optional), D6 (update -> clip order).
"""
import math

import numpy as np
import torch

from . import models as M
from . import tf_ops as T

def _f32(v):
    """TF-1.0 holds optimizer hyper-parameters as float32 constants."""
    return float(np.float32(v))


ADAM_LR = 1e-3        # train.py:20
RMSPROP_LR = 5e-5     # train.py:93
L2_WEIGHT = 0.05      # train.py:22
CLIP = 0.01           # train.py:89


class TFAdam:
    """tf.train.AdamOptimizer(lr) defaults: beta1=.9 beta2=.999 eps=1e-8 (eps outside sqrt)."""

    def __init__(self, names, params, lr=ADAM_LR, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = _f32(lr), _f32(b1), _f32(b2), _f32(eps), 0
        self.m = {n: torch.zeros_like(params[n]) for n in names}
        self.v = {n: torch.zeros_like(params[n]) for n in names}

    def apply(self, params, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for n, g in grads.items():
            self.m[n] = self.b1 * self.m[n] + (1 - self.b1) * g
            self.v[n] = self.b2 * self.v[n] + (1 - self.b2) * g * g
            params[n] = params[n] - lr_t * self.m[n] / (self.v[n].sqrt() + self.eps)


class TFRMSProp:
    """tf.train.RMSPropOptimizer(lr) defaults: decay=.9 momentum=0 eps=1e-10, ms initialised to 1."""

    def __init__(self, names, params, lr=RMSPROP_LR, decay=0.9, eps=1e-10):
        self.lr, self.decay, self.eps = _f32(lr), _f32(decay), _f32(eps)
        self.ms = {n: torch.ones_like(params[n]) for n in names}

    def apply(self, params, grads):
        for n, g in grads.items():
            self.ms[n] = self.decay * self.ms[n] + (1 - self.decay) * g * g
            params[n] = params[n] - self.lr * g / torch.sqrt(self.ms[n] + self.eps)


class OracleTrainer:
    def __init__(self, params, arg_adv, arg_loss, arg_opt, arg_transform, ksize=5):
        self.p = {k: v.clone() for k, v in params.items()}
        self.adv, self.loss, self.opt, self.dna, self.ksize = arg_adv, arg_loss, arg_opt, arg_transform, ksize
        if arg_loss not in ('bce', 'wass'):
            raise ValueError('unexpected loss argument')
        self.g_names = [k for k in self.p if k.startswith('g/')]
        self.d_names = [k for k in self.p if k.startswith('d/')]
        if arg_opt == 'rmsprop':
            mk = lambda names: TFRMSProp(names, self.p)
        elif arg_opt == 'adam':
            mk = lambda names: TFAdam(names, self.p)
        else:
            raise ValueError('unexpected opt argument')
        self.g_opt, self.g_pretrain_opt, self.d_opt = mk(self.g_names), mk(self.g_names), mk(self.d_names)

    # ---- graph pieces (train.py:48-85)
    def _g(self, p, img, actions):
        if self.dna:
            return M.generator_transform(p, img, actions, self.ksize)
        return M.generator(p, img, actions), None

    def _d(self, p, img, frame, actions):
        return M.discriminator(p, torch.cat([img, frame], dim=3), actions)

    def _g_losses(self, p, img, next_frame, actions, state):
        b = img.shape[0]
        frame, st = self._g(p, img, actions)
        out = {'frame': frame, 'state': st}
        l2 = T.l1_over_batch(frame, next_frame, b)
        if self.dna:
            l2 = l2 * L2_WEIGHT + T.l2norm_over_batch(st, state, b)
        out['g_l2_loss'] = l2
        if self.adv:
            adv = T.g_adv_loss(self._d(p, img, frame, actions), self.loss)
            g = T.gdl(next_frame, frame)
            out['g_adv_loss'], out['gdl'] = adv, g
            out['g_loss'] = l2 + adv + g
        else:
            out['g_loss'] = l2
        return out

    def _with_grad(self, names):
        p = dict(self.p)
        for n in names:
            p[n] = self.p[n].detach().clone().requires_grad_(True)
        return p

    # ---- steps (train.py:114-155)
    def pretrain_g(self, img, next_frame, actions, state):
        p = self._with_grad(self.g_names)
        out = self._g_losses(p, img, next_frame, actions, state)
        grads = torch.autograd.grad(out['g_l2_loss'], [p[n] for n in self.g_names], allow_unused=True)
        self.last_grads = {n: g for n, g in zip(self.g_names, grads) if g is not None}
        self.g_pretrain_opt.apply(self.p, self.last_grads)
        return out['g_loss'].detach()

    def train_g(self, img, next_frame, actions, state, return_all=False):
        p = self._with_grad(self.g_names)
        out = self._g_losses(p, img, next_frame, actions, state)
        grads = torch.autograd.grad(out['g_loss'], [p[n] for n in self.g_names], allow_unused=True)
        gdict = {n: g for n, g in zip(self.g_names, grads) if g is not None}
        self.last_grads = gdict
        self.g_opt.apply(self.p, gdict)
        if return_all:
            return {k: (v.detach() if v is not None else None) for k, v in out.items()}
        return out['frame'].detach()

    def train_d(self, img, next_frame, actions, return_all=False):
        p = self._with_grad(self.d_names)
        with torch.no_grad():
            frame, _ = self._g(self.p, img, actions)
        d_gen = self._d(p, img, frame, actions)
        d_real = self._d(p, img, next_frame, actions)
        total, direct, gen = T.d_loss(d_real, d_gen, self.loss)
        grads = torch.autograd.grad(total, [p[n] for n in self.d_names], allow_unused=True)
        gdict = {n: g for n, g in zip(self.d_names, grads) if g is not None}
        self.last_grads = gdict
        self.d_opt.apply(self.p, gdict)
        for n in self.d_names:                      # train.py:89,140-143 (D6: after the update)
            self.p[n] = self.p[n].clamp(_f32(-CLIP), _f32(CLIP))
        if return_all:
            return {'d_loss': total.detach(), 'discriminator_direct_loss': direct.detach(),
                    'discriminator_gen_loss': gen.detach(), 'd_out_gen': d_gen.detach(),
                    'd_out_real': d_real.detach(), 'frame': frame}
        return None

    def test(self, img, next_frame, actions):
        with torch.no_grad():
            frame, st = self._g(self.p, img, actions)
            return frame, st, T.psnr(next_frame, frame)

    def test_sequence(self, frames, next_frames, actions, steps=None):
        """Recursive rollout (reference train.py:157-176 / the eval block at :285-298): the predicted frame and the
        predicted state are fed back in; the commanded action of step j comes from the data.
        frames, next_frames: [B,T,H,W,3]; actions: [B,T,10] (action 0:5, state 5:10).  Returns [B,steps,H,W,3], PSNRs."""
        if steps == 'literal':      # the reference's own test_sequence (train.py:157-176): six steps, stride-2 indexing
            cur, state, out = frames[:, 0], actions[:, 0, 5:], []
            for j in range(6):
                acs = torch.cat([actions[:, j * 2, :5], state], dim=1)
                frame, st, _ = self.test(cur, next_frames[:, j * 2], acs)
                out.append(frame)
                cur = frame
                state = st if st is not None else actions[:, j * 2, 5:]
            return torch.stack(out, dim=1), cur[1:7]
        steps = steps if steps is not None else next_frames.shape[1] - 1
        cur, state = frames[:, 0], actions[:, 0, 5:]
        out, psnrs = [], []
        for j in range(steps):
            acs = torch.cat([actions[:, j, :5], state], dim=1)
            frame, st, ps = self.test(cur, next_frames[:, j + 1], acs)
            out.append(frame)
            psnrs.append(float(ps))
            cur = frame
            state = st if st is not None else actions[:, j + 1, 5:]
        return torch.stack(out, dim=1), psnrs
