#!/usr/bin/env python
"""Where does the run-to-run sensitivity of the bf16 discriminator gradient come from?  (CPU, test-side: it runs the oracle.)

Two bf16 runs of ONE discriminator step whose arithmetic differs at rounding level (a BatchNorm statistic summed in another
order) differ by several per cent in the D gradient (tests/test_gpu_train.py::test_epilogue_statistics_match_the_statistics_pass).
The round-3 review suspected the BatchNorm'd, activation-free head d/conv6 (models.py:87-88) and asked for its neighbourhood
to be kept in float32.  This script measures that on the oracle's bf16-storage emulation (oracle.tf_ops.bf16_storage), batch 8,
fp64 arithmetic: the D step is run twice, the second time with a 1e-7 relative perturbation of d/conv1's conv output (what a
different summation order does), for variants that keep more and more of the discriminator's tail in float32 STORAGE AND
OPERANDS (no rounding of that layer's conv output, BatchNorm output, or of the next conv's input operand):

  python tests/bf16_head_sensitivity.py > profiles/r4/n_bf16_head_sensitivity.txt

Reading (profiles/r4/n_bf16_head_sensitivity.txt): keeping the head's neighbourhood in float32 (conv5 -> conv6, what the review
proposed) changes nothing - 3.6e-2 either way; the difference only shrinks as the NUMBER of bf16-stored layers between the
perturbation and the loss shrinks (1.8e-2 with float32 from conv4's output on, 0.9e-2 from conv3's, 0.4e-2 with no bf16 layer in D
at all), i.e. it is storage-rounding chaos (every bf16 layer re-rounds ~0.5 % of its elements by a whole ulp once its input moved
by 2e-5) amplified by the loss geometry: the head's BatchNorm backward removes the constant and the linear part of the bce
gradient, which is ~97 % of it, so a 2e-3 relative change of the logits is several per cent of what is left.  The error against
EXACT arithmetic (last columns) is 6.8 % whatever is kept in float32 inside D - it is set by the bf16 rounding of D's INPUT frames
and of the generator that produced them.  float32 operands for conv4 / conv5 cost 4-9 % of the config-3 step (their fp32 matrix-core
kernels run at 1/16 of the bf16 rate on launch-bound 0.8-GFLOP contractions) and still leave 1-2 %; not taken."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import models as OM, tf_ops as T   # noqa: E402

KEEP = {}            # 'd/convN' -> subset of {'in', 'conv', 'out'} kept unrounded
PERTURB = [0.0]


def _layer(params, net, spec, x, create=None):
    """oracle.models._layer with per-layer exemptions from bf16 storage and the perturbation hook."""
    scope, kind, cout, k, s, pad, norm, act = spec
    name = net + '/' + scope
    keep = KEEP.get(name, ())
    w = T.q_weight(params[name + '/weights'])
    x = x if 'in' in keep else T.q_act(x)
    y = T.conv2d(x, w, s, pad) if kind == 'c' else T.conv2d_transpose(x, w, s, pad)
    if PERTURB[0] and name == 'd/conv1':
        g = torch.Generator().manual_seed(5)
        y = y * (1 + PERTURB[0] * torch.randn(y.shape, generator=g, dtype=y.dtype))
    y = T.q_grad(y) if ((norm and act is None) or 'conv' in keep) else T.q_act(y)
    y = T.batch_norm_train(y, params[name + '/BatchNorm/beta']) if norm else y + params[name + '/biases']
    out = OM._ACT[act](y)
    return out if (not (norm and act is not None) or 'out' in keep) else T.q_act(out)


def run_to_run_floor(params, x, y, a, ksize=5, loss='bce', perturb=1e-7, dtype=torch.float32):
    """The oracle's own answer to "how far apart are two bf16 runs of this D step that differ at rounding level": relative L2
    difference of the whole D gradient between the bf16-storage emulation and the same with d/conv1's conv output perturbed
    by `perturb` (relative, random sign and size).  Used by tests/test_gpu_train.py to bound what two HIP runs may differ by."""
    prev = OM._layer
    OM._layer = _layer
    KEEP.clear()
    try:
        outs = []
        for pert in (0.0, perturb):
            PERTURB[0] = pert
            p = {k: torch.as_tensor(v, dtype=dtype).clone().requires_grad_(k.startswith('d/')) for k, v in params.items()}
            xs, ys, as_ = (torch.as_tensor(t, dtype=dtype) for t in (x, y, a))
            with T.bf16_storage():
                with torch.no_grad():
                    fake, _ = OM.generator_transform(p, xs, as_, ksize)
                total = T.d_loss(OM.discriminator(p, T.q_act(torch.cat([xs, ys], 3)), as_),
                                 OM.discriminator(p, T.q_act(torch.cat([xs, fake], 3)), as_), loss)[0]
            total.backward()
            outs.append(torch.cat([p[k].grad.reshape(-1).double() for k in sorted(p) if k.startswith('d/')]))
        return float((outs[0] - outs[1]).norm() / outs[0].norm())
    finally:
        PERTURB[0] = 0.0
        OM._layer = prev


def main():
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    B = 8
    params = OM.init_params(True, batch=2, img=64, ksize=5, seed=3, dtype=torch.float64)
    rng = np.random.default_rng(21)
    x = torch.tensor(rng.uniform(-1, 1, (B, 64, 64, 3)))
    y = torch.clamp(torch.roll(x, 2, 2) + 0.05 * torch.tensor(rng.standard_normal(x.shape)), -1, 1)
    a = torch.tensor(rng.standard_normal((B, 10)))
    OM._layer = _layer

    def d_grad(emulate):
        p = {k: v.clone().requires_grad_(k.startswith('d/')) for k, v in params.items()}
        q = T.q_act if emulate else (lambda t: t)
        if emulate:
            with T.bf16_storage():
                with torch.no_grad():
                    fake, _ = OM.generator_transform(p, x, a, 5)
                loss = T.d_loss(OM.discriminator(p, q(torch.cat([x, y], 3)), a), OM.discriminator(p, q(torch.cat([x, fake], 3)), a), 'bce')[0]
        else:
            with torch.no_grad():
                fake, _ = OM.generator_transform(p, x, a, 5)
            loss = T.d_loss(OM.discriminator(p, torch.cat([x, y], 3), a), OM.discriminator(p, torch.cat([x, fake], 3), a), 'bce')[0]
        loss.backward()
        return torch.cat([p[k].grad.reshape(-1) for k in sorted(p) if k.startswith('d/')])

    exact = d_grad(False)
    full = ('in', 'conv', 'out')
    variants = [
        ('as shipped (conv6 conv output float32)', {}),
        ('float32 around the head: conv5 output .. conv6', {'d/conv5': ('conv', 'out'), 'd/conv6': ('in',)}),
        ('float32 from conv4\'s output on', {'d/conv4': ('conv', 'out'), 'd/conv5': full, 'd/conv6': ('in',)}),
        ('float32 from conv3\'s output on', {'d/conv3': ('conv', 'out'), 'd/conv4': full, 'd/conv5': full, 'd/conv6': ('in',)}),
        ('no bf16 tensor inside D (bf16 input frames only)', {'d/conv%d' % i: full for i in range(1, 7)}),
    ]
    print('# D step of the DNA GAN, batch 8, bf16-storage emulation (fp64 arithmetic); relative L2 difference of the whole D gradient')
    print('# %-52s %-30s %s' % ('variant', 'two runs, 1e-7 apart at d/conv1', 'against exact arithmetic (diff, cosine)'))
    for name, keep in variants:
        KEEP.clear()
        KEEP.update(keep)
        PERTURB[0] = 0.0
        g0 = d_grad(True)
        PERTURB[0] = 1e-7
        g1 = d_grad(True)
        print('%-54s %-30.3e %.3e  %.5f' % (name, float((g0 - g1).norm() / g0.norm()), float((g0 - exact).norm() / exact.norm()),
                                            float(torch.dot(g0, exact) / g0.norm() / exact.norm())))


if __name__ == '__main__':
    main()
