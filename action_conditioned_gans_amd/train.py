"""Adversarial training step and loop with the reference's Trainer / CLI surface (train.py:27-333).

``Trainer(sess, arg_adv, arg_loss, arg_opt, arg_transform)`` builds the static graph once -
placeholders, G, D(fake), D(real), losses, three optimizers, the D weight clip - and its step methods
are single ``sess.run`` calls exactly as in the reference; the session prunes each fetch to what it
needs and replays it as one HIP graph (graph.py).  Hyper-parameters the reference hard-codes as module
constants (train.py:16-25) are constructor arguments here so that the BASELINE configurations (batch
2/32/128/256, 128x128, ksize 11, n_critic) are expressible.

Reference defects resolved here: D1 `==`; D3 D actions tiled to H/4; D4 state head
optional in ``test``; D6 update -> clip; D7 eval uses its own states; D8 boolean flags accept
``--adv`` and ``--adv True|False``; D9 ksize is a parameter (default 5).
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import graph as G
from . import models as M
from . import ops as O
from . import optim
from .saver import Saver
from .util import build_all_mask

ADAM_LR = 1e-3          # train.py:20
RMSPROP_LR = 5e-5       # train.py:93
L2_WEIGHT = 0.05        # train.py:22
CLIP_VALUE = 0.01       # train.py:89
PRETRAIN_ITER = 20      # train.py:24
TRAIN_ITER = 60000      # train.py:25
ACTION_DIM, STATE_DIM = 10, 5


class Trainer:
    def __init__(self, sess, arg_adv, arg_loss, arg_opt, arg_transform, batch_size=64, img_size=64, ksize=5,
                 seed=0, batched_d=True, lookahead=True):
        """``lookahead`` (no reference counterpart, off the reference's call path unless asked for): builds a second generator
        instance on a batch of 2 B - the pair (generator-step samples ; discriminator-step samples), BatchNorm statistics per half -
        that ``train_d(..., next_g=...)`` runs INSTEAD of the batch-B instance; the ``train_g`` call that follows with the announced
        inputs then finds its generator forward pass done.  The two passes read the same generator weights (the D step does not
        touch them), so this is the reference's arithmetic - one D step, then one G step (train.py:241-263) - with the two
        generator forward passes of an iteration sharing their launches at twice the GEMM height."""
        self.sess = sess
        self.batch_size, self.img_size, self.ksize = batch_size, img_size, ksize
        self.arg_adv, self.arg_loss, self.arg_opt, self.arg_transform = arg_adv, arg_loss, arg_opt, arg_transform
        if arg_loss not in ('bce', 'wass'):
            raise ValueError('unexpected loss argument')
        B, S = batch_size, img_size
        O.set_random_seed(seed)

        self.img_ph = G.placeholder((B, S, S, 3), name='current_frame')
        # the same frames with a channel pitch of 4 (zero pad channel): lets g/conv1 gather with 16-byte loads
        self.img_ph.padded = self._img_pad = G.placeholder((B, S, S, 3), name='current_frame_conv', channel_pitch=O.cpad(3), act=True)
        self.next_frame_ph = G.placeholder((B, S, S, 3), name='next_frame')
        self.action_ph = G.placeholder((B, ACTION_DIM), name='action')
        self.next_state = G.placeholder((B, STATE_DIM), name='next_state')

        # generator (train.py:52-61); the action tile + concat is fused inside the model builders
        graph = G.get_default_graph()
        dp = graph.collections.get('data_parallel')
        # (synchronised BatchNorm builds other ops per layer and is a validation mode: it keeps the plain call path)
        self.lookahead = bool(lookahead) and batched_d and not (dp is not None and dp.active and dp.sync_bn)

        def build_g(images, actions, batch, reuse):
            if arg_transform:
                return M.build_generator_transform(images, actions, batch_size=batch, ksize=ksize, reuse=reuse)
            return M.build_generator(images, actions, reuse=reuse), None
        n0 = len(graph.ops)
        self.g_out, self.g_state_out = build_g(self.img_ph, self.action_ph, B, False)
        n1 = len(graph.ops)
        self.g_next_frame = self.g_out
        self.pair_img_ph = self.pair_action_ph = self._g_pair_out = None
        self._stash_copy, self._g_extra = None, []
        if self.lookahead:
            # the pair instance: rows [0, B) = the samples of the G step that follows, rows [B, 2 B) = this D step's samples
            self.pair_img_ph = G.placeholder((2 * B, S, S, 3), name='frame_pair')
            self.pair_img_ph.padded = self._pair_img_pad = G.placeholder((2 * B, S, S, 3), name='frame_pair_conv', channel_pitch=O.cpad(3), act=True)
            self.pair_action_ph = G.placeholder((2 * B, ACTION_DIM), name='action_pair')
            with O.arg_scope([O.batch_norm], groups=2):
                self._g_pair_out, _ = build_g(self.pair_img_ph, self.pair_action_ph, 2 * B, True)
            # every tensor of the batch-B instance IS the first half of its twin in the pair instance: what the pair pass
            # computes for the G step's samples is exactly what that step's backward pass reads
            self._g_ops, g_pair_ops = graph.ops[n0:n1], graph.ops[n1:len(graph.ops)]
            _alias_first_half(self._g_ops, g_pair_ops)

        # discriminator on (x_t, fake) then (x_t, real), sharing variables (train.py:63-70).
        # batched_d: the D step runs D ONCE on [fake ; real] stacked along the batch, BatchNorm statistics kept
        # per half (groups=2) - arithmetically the two reference calls, at twice the GEMM height and half the
        # launches.  The G step still uses the batch-B D(fake) graph (D(real) is pruned there anyway).
        self._pair_concat = None
        if batched_d:
            # ONE buffer of 3 B discriminator inputs [spare ; generated ; real], 6 channels at a pitch of 8 (16-byte gathers in
            # d/conv1): D(both) reads rows [B, 3 B), D(fake) rows [B, 2 B); the pair generator writes rows [0, 2 B) - its second
            # half, the D step's samples, lands where D(both) expects the generated frames
            n_part = B * S * S * 8
            d_in_all = O._new_act((3 * B, S, S, 8), 'd_in_all:0')
            win = lambda k, rows, name: d_in_all.view(k * n_part, (rows, S, S, 8), name=name)     # noqa: E731
            d_in_gen = O.concat([self.img_ph, self.g_next_frame], axis=3, name='d_in_gen', out=win(1, B, 'd_in_all/gen'), pitch=8, act=True)
            d_in_real = O.concat([self.img_ph, self.next_frame_ph], axis=3, name='d_in_real', out=win(2, B, 'd_in_all/real'), pitch=8, act=True)
            d_in_both = win(1, 2 * B, 'd_in_both:0')
            O.JoinOp([d_in_gen, d_in_real], d_in_both, 'd_in_both')
            d_in_both.valid_c = d_in_gen.valid_c
            self._stash_copy = None
            if self.lookahead:
                self._pair_concat = O.concat([self.pair_img_ph, self._g_pair_out], axis=3, name='d_in_pair', out=win(0, 2 * B, 'd_in_all/pair'),
                                             pitch=8, act=True).op
                if self._pair_concat.by_producer:
                    # the pair's DNA kernel writes whole discriminator-input pixels into rows [0, 2 B): the G step's own frames sit
                    # in the spare rows [0, B) afterwards, and the G step moves them to rows [B, 2 B) with one copy
                    self._stash_copy = O.CopyRowsOp(win(0, B, 'd_in_all/spare'), win(1, B, 'd_in_all/gen_copy'), 'd_in_gen/from_pair')
        else:
            d_in_gen = O.concat([self.img_ph, self.g_next_frame], axis=3, name='d_in_gen', pitch=8, act=True)
            d_in_real = O.concat([self.img_ph, self.next_frame_ph], axis=3, name='d_in_real', pitch=8, act=True)
        self.d_out_gen = M.build_discriminator(d_in_gen, self.action_ph, reuse=False)
        if batched_d:
            with O.arg_scope([O.batch_norm], groups=2):
                self.d_out_both = M.build_discriminator(d_in_both, O.repeat_batch(self.action_ph, 2), reuse=True)
            self.d_out_real = self.d_out_both.view(self.d_out_gen.numel, self.d_out_gen.shape, name='d_out_real')
        else:
            self.d_out_both = None
            self.d_out_real = M.build_discriminator(d_in_real, self.action_ph, reuse=True)

        # losses (train.py:72-85)
        self.g_psnr = O.build_psnr(self.next_frame_ph, self.g_next_frame)
        l1, gdl = O.frame_losses(self.g_out, self.next_frame_ph)
        g_l2_loss = l1 / B
        if arg_transform:
            with G.get_default_graph().side_branch():      # the state head's loss belongs to its side chain (models.py)
                state_loss = O.l2_norm(self.g_state_out, self.next_state, name='g_state_loss')
            g_l2_loss = g_l2_loss * L2_WEIGHT + state_loss / B
        self.g_l2_loss = g_l2_loss
        self.summaries = {}
        if arg_adv:
            self.g_adv_loss = O.build_g_adv_loss(self.d_out_gen, arg_loss)
            self.g_loss = g_l2_loss + self.g_adv_loss + gdl
            self.summaries['g_adv_loss'] = self.g_adv_loss
        else:
            self.g_loss = g_l2_loss
        if batched_d:
            self.d_loss = O.build_d_loss_batched(self.d_out_both, arg_loss, summaries=self.summaries)
        else:
            self.d_loss = O.build_d_loss(self.d_out_real, self.d_out_gen, arg_loss, summaries=self.summaries)

        graph = G.get_default_graph()
        self.g_vars = graph.trainable_variables('g')
        self.d_vars = graph.trainable_variables('d')
        self.clip_d = [optim.clip_by_value_assign(p, -CLIP_VALUE, CLIP_VALUE) for p in self.d_vars]

        if arg_opt == 'rmsprop':
            make = lambda name: optim.RMSPropOptimizer(RMSPROP_LR, name=name)
        elif arg_opt == 'adam':
            make = lambda name: optim.AdamOptimizer(ADAM_LR, name=name)
        else:
            raise ValueError('unexpected opt argument')
        self.g_opt_op = make('g_opt').minimize(self.g_loss, var_list=self.g_vars)
        self.g_pretrain_opt_op = make('g_pretrain_opt').minimize(g_l2_loss, var_list=self.g_vars)
        self.d_opt_op = make('d_opt').minimize(self.d_loss, var_list=self.d_vars)

        # the seven tf.summary scalars of ops.py:48-49 / train.py:104-111, names kept
        self.summaries.update({'discriminator_loss': self.d_loss, 'g_loss': self.g_loss, 'g_l2_loss': g_l2_loss,
                               'g_psnr': self.g_psnr})
        self._summary_names = sorted(self.summaries)
        self.merged_summaries = [self.summaries[k] for k in self._summary_names]
        self._zero_state = np.zeros((B, STATE_DIM), np.float32)
        self._announced = None          # (images, actions) of the G step a look-ahead D step has prepared
        self._skip_d = self._skip_g = None
        if self.lookahead:
            # what the pair pass replaces.  D step: the whole batch-B generator and the launch that puts its frame into D's input.
            # G step: the generator up to and including the frame (an alias of the pair's first half) and, where the DNA kernel
            # wrote the discriminator-input pixels too, the concatenation: those pixels are copied over from the spare rows
            trunk = _ancestors(self.g_out, set(map(id, self._g_ops)))
            self._skip_d = frozenset(trunk + [d_in_gen.op])
            if self._stash_copy is not None:        # DNA generator: the frame AND its copy in D(fake)'s input exist already (one 4 MB copy)
                self._skip_g, self._g_extra = frozenset(trunk + [d_in_gen.op]), [self._stash_copy]
            else:                                   # plain generator: its concat launch puts the (aliased) frame into D(fake)'s input as usual
                self._skip_g, self._g_extra = frozenset(trunk), []

    # ---- steps: one sess.run each (train.py:114-155)
    def _feed(self, input_images, next_frame, actions, state=None):
        return {self.img_ph: input_images, self._img_pad: input_images, self.next_frame_ph: next_frame,
                self.action_ph: actions, self.next_state: self._zero_state if state is None else state}

    def pretrain_g(self, input_images, next_frame, actions, state):
        self._announced = None
        _, g_res = self.sess.run([self.g_pretrain_opt_op, self.g_loss], self._feed(input_images, next_frame, actions, state))
        return float(g_res[0])

    def train_g(self, input_images, next_frame, actions, state, device_fetch=False):
        # the generator forward pass of these very inputs was run by the preceding train_d(..., next_g=(input_images, actions)):
        # the program then starts behind it (Session.run skip=)
        prepared, self._announced = self._announced, None
        if prepared is not None and prepared[0] is input_images and prepared[1] is actions:
            fd = self._feed(input_images, next_frame, actions, state)
            if len(prepared) == 4:      # host arrays the announcing D step already put on the device: no second upload
                fd[self.img_ph] = fd[self._img_pad] = prepared[2]
                fd[self.action_ph] = prepared[3]
            res = self.sess.run([self.g_opt_op, self.g_next_frame] + self._g_extra, fd, device_fetch=device_fetch, skip=self._skip_g)
            return res[1]
        _, gen_next_frames = self.sess.run([self.g_opt_op, self.g_next_frame],
                                           self._feed(input_images, next_frame, actions, state), device_fetch=device_fetch)
        return gen_next_frames

    def train_d(self, input_images, next_frame, actions, summarize=False, next_g=None, pair=None, next_d=None):
        """One discriminator step (train.py:132-144).  ``next_g`` / ``next_d`` = (input_images, actions) of the ``train_g`` /
        ``train_d`` call that follows (extension, see __init__ ``lookahead``): this step's generator pass then also covers that
        step's samples, and that step starts behind its generator forward pass (with n_critic > 1 the D steps alternate: one runs
        the pair pass for itself and its successor, the next runs no generator at all).  ``pair`` = the two batches already
        joined, (frames [2 B, H, W, 3], actions [2 B, 10]) with the FOLLOWING step's samples first - saves the concatenation
        here when the caller keeps its batches that way."""
        prepared, self._announced = self._announced, None
        fd = self._feed(input_images, next_frame, actions)
        if summarize:
            _, summ, _ = self.sess.run([self.d_opt_op, self.merged_summaries, self.clip_d], fd)
            return self._named(summ)
        if prepared is not None and prepared[0] is input_images and prepared[1] is actions:
            # the preceding D step ran the generator for these samples: their frames wait in the spare rows
            if len(prepared) == 4:
                fd[self.img_ph] = fd[self._img_pad] = prepared[2]
                fd[self.action_ph] = prepared[3]
            self.sess.run([self.d_opt_op, self.clip_d] + self._g_extra, fd, skip=self._skip_g)
            return None
        nxt = next_g if next_g is not None else next_d
        if nxt is not None and self.lookahead:
            self._announced = (nxt[0], nxt[1])
            if pair is None:
                if self.sess.rt.is_cuda and not torch.is_tensor(input_images) and not torch.is_tensor(nxt[0]):
                    # host arrays (the reference's numpy call path): each batch goes to the device ONCE - the pair is joined there,
                    # this step and the announced one are fed the device copies (round 5: -4.5 MB of uploads and a 3 MB host
                    # concatenation per iteration)
                    # (pinned staging + a copy stream of its own: neither the host nor the running step waits; one copy for the four)
                    x_d, a_d, x_n, a_n = self.sess.upload_many([input_images, actions, nxt[0], nxt[1]])
                    fd[self.img_ph] = fd[self._img_pad] = x_d
                    fd[self.action_ph] = a_d
                    pair = (torch.cat([x_n, x_d]), torch.cat([a_n, a_d]))
                    self._announced = (nxt[0], nxt[1], x_n, a_n)
                else:
                    pair = (_join(nxt[0], input_images), _join(nxt[1], actions))
            fd.update({self.pair_img_ph: pair[0], self._pair_img_pad: pair[0], self.pair_action_ph: pair[1]})
            self.sess.run([self.d_opt_op, self.clip_d, self._pair_concat], fd, skip=self._skip_d)
            return None
        self.sess.run([self.d_opt_op, self.clip_d], fd)
        return None

    def test(self, input_images, next_frame, actions):
        self._announced = None
        tensors = [self.g_next_frame] + ([self.g_state_out] if self.g_state_out is not None else []) + [self.merged_summaries]
        res = self.sess.run(tensors, self._feed(input_images, next_frame, actions))
        gen_next_frames, summ = res[0], res[-1]
        gen_next_state = res[1] if self.g_state_out is not None else None      # defect D4
        return gen_next_frames, gen_next_state, self._named(summ)

    def test_sequence(self, input_images, test_next_frame, test_actions, steps=None, literal=False, device_loop=None):
        """Recursive rollout: feed each prediction (and predicted state) back in.
        ``device_loop`` (default: on for a GPU session): from the second step on the prediction and the predicted state stay on the
        device between steps - the program of those steps fetches nothing but the two, so it carries no loss ops either - and
        all predictions come to the host in one copy at the end.  Same kernels on the same bits as the step-by-step numpy round
        trip (``device_loop=False``, the reference's own loop), which it replaces only in where the intermediate frames live.

        Default: the evaluation block of the reference's training loop (train.py:285-298) - T-1 steps, step j commanded
        by ``test_actions[:, j]`` and scored against ``test_next_frame[:, j + 1]`` (defect D7: own states); returns
        ``(predicted [B, steps, H, W, 3], summaries of step 0)``.
        ``literal=True``: the reference's method of this name exactly as written (train.py:157-176) - SIX steps, step j
        reads ``test_actions[:, 2 j, :5]`` and ``test_next_frame[:, 2 j]`` (the sequences must hold >= 11 frames),
        and the second return value is ``current_frame[1:7]``, samples 1..6 of the last prediction."""
        if literal:
            predicted = []
            current_frame = input_images[:, 0]
            current_state = test_actions[:, 0, 5:]
            for j in range(0, 6):
                acs = np.concatenate((test_actions[:, j * 2, :5], current_state), axis=1).astype(np.float32)
                out, st, _ = self.test(current_frame, test_next_frame[:, j * 2], acs)
                predicted.append(out)
                current_frame = out
                current_state = st if st is not None else test_actions[:, j * 2, 5:]      # plain generator: no state head (D4)
            return np.transpose(np.array(predicted), (1, 0, 2, 3, 4)), current_frame[1:7]
        steps = steps if steps is not None else test_next_frame.shape[1] - 1
        if device_loop is None:
            device_loop = self.sess.rt.is_cuda
        if device_loop and steps >= 1:
            out, st, summ0 = self.test(input_images[:, 0], test_next_frame[:, 1], np.asarray(test_actions[:, 0], np.float32))
            acts = self.sess.upload(np.asarray(test_actions[:, :steps + 1], np.float32))          # [B, steps + 1, 10], once
            frame = self.sess.upload(out)
            state = self.sess.upload(st) if st is not None else acts[:, 1, 5:]
            frames = [frame]
            fetches = [self.g_next_frame] + ([self.g_state_out] if self.g_state_out is not None else [])
            for j in range(1, steps):
                acs = torch.cat([acts[:, j, :5], state], dim=1).contiguous()
                fd = self._feed(frame, test_next_frame[:, j + 1], acs)      # (next_frame is not read by this program: checked, not uploaded)
                self._announced = None
                res = self.sess.run(fetches, fd, device_fetch=True)
                frame = res[0].float().clone()                             # the fetch is the tensor's own buffer: the next step overwrites it
                state = res[1].float().clone() if self.g_state_out is not None else acts[:, j + 1, 5:]
                frames.append(frame)
            predicted = torch.stack(frames, dim=1).cpu().numpy()
            return predicted, summ0
        predicted, summ0 = [], None
        current_frame = input_images[:, 0]
        current_state = test_actions[:, 0, 5:]
        for j in range(steps):
            acs = np.concatenate((test_actions[:, j, :5], current_state), axis=1).astype(np.float32)
            out, st, summ = self.test(current_frame, test_next_frame[:, j + 1], acs)
            summ0 = summ0 or summ
            predicted.append(out)
            current_frame = out
            current_state = st if st is not None else test_actions[:, j + 1, 5:]
        return np.transpose(np.array(predicted), (1, 0, 2, 3, 4)), summ0

    def _named(self, values):
        return {k: float(np.asarray(v).reshape(-1)[0]) for k, v in zip(self._summary_names, values)}


def _join(first, second):
    """[first ; second] along the batch axis, numpy arrays or (device) torch tensors."""
    if torch.is_tensor(first):
        return torch.cat([first, second.to(first.device) if torch.is_tensor(second) else torch.as_tensor(second, device=first.device)], dim=0)
    return np.concatenate([np.asarray(first), np.asarray(second.cpu() if torch.is_tensor(second) else second)], axis=0)


def _ancestors(tensor, within):
    """The ops (restricted to the ids in ``within``) that ``tensor`` depends on, its producer included, creation order."""
    seen, stack = {}, [tensor.op]
    while stack:
        op = stack.pop()
        if op is None or id(op) in seen or id(op) not in within:
            continue
        seen[id(op)] = op
        stack.extend(t.op for t in op.inputs)
    return sorted(seen.values(), key=lambda o: o.index)


def _alias_first_half(ops_small, ops_pair):
    """Two instances of one network built by the same code, the second on twice the batch: make every tensor the first
    instance's ops produce a window onto the FIRST HALF of its twin (batch-major storage: the first B samples; BatchNorm
    statistics [groups = 2, C]: group 0).  Tensors that already are windows (a BatchNorm output placed in its concatenation)
    follow through their base."""
    if len(ops_small) != len(ops_pair):
        raise RuntimeError('look-ahead: the two generator instances differ in structure (%d vs %d ops)' % (len(ops_small), len(ops_pair)))
    for a, b in zip(ops_small, ops_pair):
        if type(a) is not type(b) or len(a.outputs) != len(b.outputs):
            raise RuntimeError('look-ahead: %r has no twin in the pair instance (%r)' % (a, b))
        for ta, tb in zip(a.outputs, b.outputs):
            if ta.view_of is not None or ta.alias_of is not None or isinstance(ta, (G.Variable, G.Placeholder)):
                continue
            if tb.numel != 2 * ta.numel or ta.dtype != tb.dtype:
                raise RuntimeError('look-ahead: %r is not half of %r' % (ta, tb))
            ta.view_of = (tb, 0)


# ---- synthetic push-style data (rng(7), frames U(-1,1), action||state N(0,1)) ----------
class SyntheticPush:
    """Seeded random sequences in the shape of the push batches.  ``pool`` > 0: the first ``pool`` batches are kept and handed out
    round robin afterwards (drawing 3 M uniform numbers per batch costs a host core 10-50 ms - many training steps; the loop
    benchmark uses a pool so that what it times is the loop)."""

    def __init__(self, batch_size, seq_len=8, img_size=64, seed=7, rank=0, pool=0):
        self.rng = np.random.default_rng(seed + 1000 * rank)
        self.shape = (batch_size, seq_len, img_size, img_size, 3)
        self.batch_size, self.seq_len = batch_size, seq_len
        self.pool, self._kept, self._next = int(pool), [], 0

    def get_batch(self):
        """-> (frames, frames, action||state [B,T,10], state [B,T,5]) like ops.get_batch (ops.py:15-17)."""
        if self.pool and len(self._kept) == self.pool:
            img, acts = self._kept[self._next % self.pool]
            self._next += 1
            return img, img, acts, acts[:, :, 5:].copy()
        img = self.rng.uniform(-1.0, 1.0, self.shape).astype(np.float32)
        acts = self.rng.standard_normal((self.batch_size, self.seq_len, ACTION_DIM)).astype(np.float32)
        if self.pool:
            self._kept.append((img, acts))
        return img, img, acts, acts[:, :, 5:].copy()


def select_pairs(rng_randint, boolean_mask, batch_size):
    """(t, t+1) selection of train.py:231-232,249-250,258-259."""
    start_mask = boolean_mask[rng_randint(0, len(boolean_mask), size=batch_size)]
    return start_mask, np.roll(start_mask, 1, axis=1)


class _PairSelections:
    """The frame-pair selections of the coming iterations, drawn AHEAD of the loop in the reference's order (train.py:231-232
    per pretraining iteration; 249-250 per D step, then 258-259 once for the G step on the last D batch), from a private copy
    of numpy's global generator as it stands when the loop starts - the same numbers the loop would draw one call at a time,
    since nothing else in the loop draws from it.  Knowing them early is what lets a PushDataset decode only the frames a
    step will read (``data.announce``: 2-4 of a record's 7 JPEGs instead of all of them); a source without ``announce``
    (SyntheticPush) just gets its selections from here."""

    def __init__(self, boolean_mask, batch_size, d_per_g, pretrain_iter, train_iter, data, ahead=8):
        self.rng = np.random.RandomState()
        self.rng.set_state(np.random.get_state())
        self.mask, self.batch_size, self.d_per_g = boolean_mask, batch_size, d_per_g
        self.pretrain_iter, self.train_iter = pretrain_iter, train_iter
        self.announce = getattr(data, 'announce', None)
        self.ahead, self.drawn, self.queue = max(int(ahead), 1), 0, []

    def _draw(self):
        i = self.drawn
        n = 1 if i < self.pretrain_iter else self.d_per_g + 1
        sels = [select_pairs(self.rng.randint, self.mask, self.batch_size) for _ in range(n)]
        if self.announce is not None:
            if i < self.pretrain_iter:
                self.announce(sels[0][0] | sels[0][1])
            else:
                for j in range(self.d_per_g):
                    need = sels[j][0] | sels[j][1]
                    if j == self.d_per_g - 1:                       # the G step selects again on the last D batch
                        need = need | sels[-1][0] | sels[-1][1]
                    self.announce(need)
        self.queue.append(sels)
        self.drawn += 1

    def next(self):
        """The selections of the next iteration: [(start, end)] while pretraining, else [D_1, ..., D_n, G]."""
        while self.drawn < self.train_iter and len(self.queue) < self.ahead + 1:
            self._draw()
        return self.queue.pop(0)


def _log_jsonl(path, record):
    with open(path, 'a') as f:
        f.write(json.dumps(record) + '\n')


def train(input_path, output_path, test_output_path, log_dir, model_dir, arg_adv, arg_loss, arg_opt, arg_transform,
          batch_size=64, img_size=64, seq_len=8, ksize=5, train_iter=TRAIN_ITER, pretrain_iter=PRETRAIN_ITER,
          n_critic=None, device='cuda:0', world_size=1, rank=0, process_group=None, log_every=100, quiet=False,
          eval_every=500, resume=None, dtype='f32', sync_bn=False, exact_global_batch=False, dp_collectives=None, buckets=0,
          data_workers='thread', data_threads=None, data_decode='exact', data_frames='selected', data_cache_gb=0.0, synthetic_pool=0):
    """Training loop of train.py:179-309.  ``input_path``: 'synthetic' (seeded random sequences) or a directory of
    push-dataset TFRecords, read by push_data.PushDataset (the reference's build_tfrecord_input, ops.py:140-223).
    ``dtype``: 'f32', or 'bf16' for the bf16 pipeline of BASELINE configs 3 and 5 (bf16 activations, float32 master weights).
    Data parallel (world_size > 1): ``sync_bn`` - BatchNorm statistics of the global batch; ``exact_global_batch`` -
    the run reproduces one device at the global batch (SyncBN + GDL scaled by the world size + global state-loss norm);
    ``dp_collectives`` - 'side' (default with more than one rank: all-reduces on a second HIP stream, overlapping the rest of
    backward) or 'stream' (in program order on the compute stream); ``buckets`` - all-reduce buckets per optimizer (0 = 2 for
    'side', 1 for 'stream').  ``data_workers`` / ``data_threads``: the TFRecord decode workers (push_data.PushDataset: threads or
    spawned processes filling a bounded prefetch queue, the reference's tf.train.batch(num_threads=batch_size), ops.py:209-213).
    ``data_frames``: 'selected' (default) - the loop tells the dataset batches ahead which frames each step will read and only
    those JPEGs are decoded (same frames, same bits; a step reads 2 of a record's 7) - or 'all' (every frame of every record,
    as the reference's queue runners do).  ``data_decode``: 'exact' (decode -> crop -> box mean, the reference's arithmetic) or
    'dct' (opt-in, approximate: the reduction inside libjpeg's inverse DCT, push_data.decode_frame).  ``data_cache_gb``: keep up to
    that many GiB of decoded frames in host memory - a record that comes round again in a later epoch is not decoded again (same
    bits; 0 = off, as the reference).  ``synthetic_pool``: SyntheticPush(pool=...)."""
    if data_frames not in ('selected', 'all'):
        raise ValueError("data_frames must be 'selected' or 'all'")
    np.random.seed(7)                                           # train.py:14
    synthetic = input_path in (None, '', 'synthetic')
    if synthetic:
        data = SyntheticPush(batch_size, seq_len, img_size, rank=rank, pool=synthetic_pool)
    else:
        from .push_data import PushDataset
        data = PushDataset(input_path, batch_size, training=True, img_size=img_size, rank=rank, world_size=world_size,
                           workers=data_workers, num_threads=data_threads, decode=data_decode, cache_bytes=int(data_cache_gb * 2 ** 30))
        seq_len = data.seq_len
    boolean_mask = build_all_mask(seq_len)
    G.reset_default_graph()
    if dp_collectives is None:
        dp_collectives = 'side' if world_size > 1 else 'stream'
    optim.set_data_parallel(world_size, n_buckets=buckets or None, sync_bn=sync_bn, exact_global_batch=exact_global_batch,
                            collectives=dp_collectives)
    sess = G.Session(device=device, world_size=world_size, rank=rank, process_group=process_group, dtype=dtype)
    try:
        trainer = _train_loop(sess, data, input_path, synthetic, boolean_mask, log_dir, model_dir, arg_adv, arg_loss, arg_opt, arg_transform,
                              batch_size, img_size, seq_len, ksize, train_iter, pretrain_iter, n_critic, rank, log_every, quiet, eval_every, resume,
                              select_frames=data_frames == 'selected')
        sess.rt.check_exchange_flags()     # a last look at the device-side flags of the iterations since the last log interval
    except BaseException:
        sess.close(check=False)            # tear the transport down; the exception on its way out is the one to report
        raise
    finally:
        if hasattr(data, 'close'):
            data.close()                   # PushDataset: stop the decode workers and the feeder thread
    # The session stays OPEN: the returned Trainer is usable (evaluation, more steps, reading variables).  Its owner closes it -
    # `trainer.sess.close()` (main() does): ncclCommDestroy under data parallelism and a last check of the device-side flags.
    return trainer


def _train_loop(sess, data, input_path, synthetic, boolean_mask, log_dir, model_dir, arg_adv, arg_loss, arg_opt, arg_transform, batch_size,
                img_size, seq_len, ksize, train_iter, pretrain_iter, n_critic, rank, log_every, quiet, eval_every, resume, select_frames=True):
    trainer = Trainer(sess, arg_adv, arg_loss, arg_opt, arg_transform, batch_size, img_size, ksize)
    sess.run(G.global_variables_initializer())
    saver = Saver()                                                           # train.py:215
    if resume:
        saver.restore(sess, resume)
    if synthetic:
        eval_data = SyntheticPush(batch_size, seq_len, img_size, seed=1007, rank=rank)
    else:
        from .push_data import PushDataset
        try:                                                                  # validation files: the tail of the split
            eval_data = PushDataset(input_path, batch_size, training=False, img_size=img_size, seed=1007)
        except RuntimeError:
            eval_data = PushDataset(input_path, batch_size, training=True, img_size=img_size, seed=1007)
    D_per_G = n_critic if n_critic else (5 if arg_loss == 'wass' else 1)      # train.py:217-220
    log_file = os.path.join(log_dir, 'train.jsonl') if log_dir else None
    t0 = time.time()
    selections = _PairSelections(boolean_mask, batch_size, D_per_G, pretrain_iter, train_iter, data if select_frames else None)
    for i in range(train_iter):
        sels = selections.next()
        if i < pretrain_iter:
            inp, nxt, acts, states = data.get_batch()
            sm, em = sels[0]
            trainer.pretrain_g(inp[sm], nxt[em], acts[sm], states[em])
            if not quiet:
                print('pre-train iter: ' + str(i))
            continue
        # The iteration's sub-steps, drawn up front in the reference's order (train.py:241-259: per D step a fresh batch and a
        # fresh frame-pair selection, then a NEW selection on the last batch for the G step; the steps themselves draw nothing;
        # the selections come from _PairSelections, which drew them some iterations ago in that same order),
        # so that a step can announce its successor's inputs to Trainer.train_d (look-ahead generator pass): D1 runs the
        # generator for D1 and D2, D2 runs none, ... the last pair pass covers the G step.  Logging iterations keep the plain path
        # for their last D step (its summaries read that step's own generated frames).
        subs = []
        for j in range(D_per_G):
            inp, nxt, acts, states = data.get_batch()
            sm, em = sels[j]
            subs.append((inp[sm], nxt[em], acts[sm]))
        smg, emg = sels[-1]
        g_in, g_act = inp[smg], acts[smg]
        summ, carried = None, False
        for j, (x_d, y_d, a_d) in enumerate(subs):
            last = j == D_per_G - 1
            summarize = (i % log_every == 0) and last
            follow = None
            if not carried and not summarize:                    # this step runs the pair pass for itself and its successor
                follow = (g_in, g_act) if last else ((subs[j + 1][0], subs[j + 1][2]) if not ((i % log_every == 0) and j + 1 == D_per_G - 1) else None)
            summ = trainer.train_d(x_d, y_d, a_d, summarize=summarize, next_d=follow)
            carried = follow is not None and not carried
        # (the generated frames stay on the device: the reference fetches them every step only to dump samples at i % 100 == 0,
        # train.py:130,269-273, which this loop does not do - no D2H copy, no synchronisation per iteration)
        trainer.train_g(g_in, nxt[emg], g_act, states[emg], device_fetch=True)
        if i % log_every == 0:
            # the fetches above synchronised anyway: look at the device-side flags of the one-launch BatchNorm kernels HERE, on
            # every rank, so that a step that ran on wrong statistics fails now - before anything of it is logged or
            # checkpointed - and not at exit, thousands of iterations later (graph.Runtime.check_exchange_flags)
            sess.rt.check_exchange_flags()
        if i % log_every == 0 and rank == 0:
            if not quiet:
                print('Iteration {:d}'.format(i))
            if log_file and summ:
                _log_jsonl(log_file, dict(summ, iteration=i, wall_s=time.time() - t0))
            if model_dir:
                saver.save(sess, os.path.join(model_dir, 'model{:d}'.format(i)), background=True)      # train.py:274; written by a writer thread
        if eval_every and i % eval_every == 0 and rank == 0:
            # recursive rollout over T-1 steps on held-out sequences (train.py:278-309; defect D7: own states)
            t_img, _, t_acts, _ = eval_data.get_batch()
            predicted, e_summ = trainer.test_sequence(t_img, t_img, t_acts)
            psnr = [float(10.0 * np.log10(1.0 / max(np.mean((predicted[:, j] - t_img[:, j + 1]) ** 2), 1e-30)))
                    for j in range(predicted.shape[1])]
            if log_file:
                _log_jsonl(os.path.join(log_dir, 'test.jsonl'), dict(e_summ or {}, iteration=i, rollout_psnr=psnr))
    if hasattr(eval_data, 'close'):
        eval_data.close()
    saver.wait()                     # the last checkpoints are on disk when train() returns
    return trainer


def _flag(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('true', '1', 'yes'):
        return True
    if v.lower() in ('false', '0', 'no'):
        return False
    raise argparse.ArgumentTypeError('boolean expected')


def main(argv=None):
    parser = argparse.ArgumentParser(description='action-conditioned video-prediction GAN on MI355X')
    parser.add_argument('input_path', type=str)
    parser.add_argument('output_path', type=str)
    parser.add_argument('--adv', nargs='?', const=True, default=False, type=_flag)
    parser.add_argument('--loss', type=str, default='bce')
    parser.add_argument('--opt', type=str, default='adam')
    parser.add_argument('--dna', nargs='?', const=True, default=False, type=_flag)
    parser.add_argument('--batch_size', type=int, default=64)
    parser.add_argument('--img_size', type=int, default=64)
    parser.add_argument('--seq_len', type=int, default=8)
    parser.add_argument('--ksize', type=int, default=5)
    parser.add_argument('--n_critic', type=int, default=None)
    parser.add_argument('--train_iter', type=int, default=TRAIN_ITER)
    parser.add_argument('--pretrain_iter', type=int, default=PRETRAIN_ITER)
    parser.add_argument('--dtype', type=str, default='f32', choices=['f32', 'bf16'])
    # data parallel (one process per GPU under torch.distributed.run; no reference counterpart - SURVEY 8(e))
    parser.add_argument('--sync_bn', nargs='?', const=True, default=False, type=_flag,
                        help='BatchNorm statistics of the GLOBAL batch (one small all-reduce per BatchNorm layer and direction)')
    parser.add_argument('--exact_global_batch', nargs='?', const=True, default=False, type=_flag,
                        help='reproduce ONE device at the global batch: SyncBN + GDL scaled by the world size + global state-loss norm')
    parser.add_argument('--dp_collectives', type=str, default=None, choices=['stream', 'side'],
                        help="gradient all-reduces in program order on the compute stream, or on a side HIP stream overlapping "
                             "the rest of backward (default with more than one rank)")
    parser.add_argument('--buckets', type=int, default=0, help='all-reduce buckets per optimizer (0: 2 for side, 1 for stream)')
    parser.add_argument('--data_workers', type=str, default='process', choices=['thread', 'process'],
                        help='TFRecord decode workers: spawned processes (default here: 2.3x the rate of threads on the GPU box, '
                             'profiles/r5/d_train_loop.txt) or threads (the default of train() / PushDataset: no __main__ guard needed)')
    parser.add_argument('--data_threads', type=int, default=None, help='number of decode workers (default: batch size, at most 16)')
    parser.add_argument('--data_frames', type=str, default='selected', choices=['selected', 'all'],
                        help="decode only the frames a step will read (same frames, same bits) or every frame of every record")
    parser.add_argument('--data_cache_gb', type=float, default=8.0,
                        help='GiB of host memory for decoded frames: a record seen in an earlier epoch is served from memory (same bits; '
                             '0 = decode every time as the reference; the 64x64 push training set is ~17 GiB decoded)')
    parser.add_argument('--data_decode', type=str, default='exact', choices=['exact', 'dct'],
                        help="'dct': approximate 8x reduction inside libjpeg's inverse DCT (3x cheaper; within 2-3 levels of 255)")
    args = parser.parse_args(argv)
    if args.buckets < 0:
        parser.error('--buckets must be >= 0')
    model_dir = os.path.join(args.output_path, 'models')
    log_dir = os.path.join(args.output_path, 'logs')
    os.makedirs(args.output_path)
    os.makedirs(model_dir)
    os.makedirs(log_dir)
    world_size, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world_size > 1:
        # control plane only (bootstrap of the RCCL communicator, comm.py): gradients never go through torch.distributed
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group('gloo')
    trainer = train(args.input_path, os.path.join(args.output_path, 'train_output'), os.path.join(args.output_path, 'test_output'),
                    log_dir, model_dir, args.adv, args.loss, args.opt, args.dna, batch_size=args.batch_size, img_size=args.img_size,
                    seq_len=args.seq_len, ksize=args.ksize, train_iter=args.train_iter, pretrain_iter=args.pretrain_iter,
                    n_critic=args.n_critic, device='cuda:%d' % local_rank, world_size=world_size, rank=rank, dtype=args.dtype,
                    sync_bn=args.sync_bn, exact_global_batch=args.exact_global_batch, dp_collectives=args.dp_collectives, buckets=args.buckets,
                    data_workers=args.data_workers, data_threads=args.data_threads, data_decode=args.data_decode, data_frames=args.data_frames,
                    data_cache_gb=args.data_cache_gb)
    if trainer is not None:
        trainer.sess.close()        # ncclCommDestroy under data parallelism + a last check of the device-side flags


if __name__ == '__main__':
    main()
