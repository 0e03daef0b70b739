"""Generates tests/golden/*.npz from the fp64 torch restatement (oracle/), seeded.

PARITY UNPINNED: the reference has no fixtures and cannot run (TF-1.0 absent), so these vectors pin
the ORACLE (against accidental change) and give the GPU suite fixed targets; they are not outputs of
the reference itself.  Usage:  python tests/golden/make_golden.py   (from the repo root)

Each case stores small tensors only: generated frame / state / logits / loss scalars with the initial
parameters; per variable the gradient L2 norm AND a strided elementwise sample (sample_index) of the gradient
of one D step, one G step and one pre-training step; the same sample of every parameter after 1 D step +
1 G step (RMSProp: plus the L2 norm; Adam: compared where the gradient is clearly signed, see
tests/train_cases.py).  Parameters themselves are re-created from the seed (oracle.models.init_params).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import models as OM            # noqa: E402
from oracle.trainer import OracleTrainer   # noqa: E402

CASES = {
    # name: (arg_adv, arg_loss, arg_opt, arg_transform, batch, ksize)
    'c1_plain_l1': (False, 'bce', 'adam', False, 2, 5),          # BASELINE config 1
    'c2_dna_bce_adam': (True, 'bce', 'adam', True, 2, 5),        # BASELINE config 2 at batch 2
    'c4_dna_wass_rmsprop': (True, 'wass', 'rmsprop', True, 2, 5),
    'plain_adv_bce_rmsprop': (True, 'bce', 'rmsprop', False, 2, 5),
    'dna_k6_bce_rmsprop': (True, 'bce', 'rmsprop', True, 2, 6),  # train.py:54 passes ksize=6
}
PARAM_SEED = 11


def inputs(batch, img=64):
    """default_rng(7); frames U(-1,1); action||state N(0,1)."""
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (batch, img, img, 3)).astype(np.float32)
    y = rng.uniform(-1, 1, (batch, img, img, 3)).astype(np.float32)
    a = rng.standard_normal((batch, 10)).astype(np.float32)
    s = rng.standard_normal((batch, 5)).astype(np.float32)
    return x, y, a, s


def sample_index(numel, n=193):
    """<= n element indices of a flattened variable: a stride coprime to the usual power-of-two extents, so the sample
    walks through every axis (taps, input channels, output channels) instead of one column."""
    if numel <= n:
        return np.arange(numel)
    stride = numel // n
    stride += 1 - stride % 2            # odd
    while stride % 3 == 0 or stride % 5 == 0:
        stride += 2
    return (np.arange(n) * stride) % numel


def sample(t):
    flat = t.detach().reshape(-1).numpy()
    return flat[sample_index(flat.size)].astype(np.float64)


def make_case(name):
    adv, loss, opt, dna, batch, ksize = CASES[name]
    params = OM.init_params(dna, batch=batch, ksize=ksize, seed=PARAM_SEED, dtype=torch.float32)
    params = {k: v.double() for k, v in params.items()}
    x, y, a, s = inputs(batch)
    td = lambda t: torch.from_numpy(t).double()
    out = {}
    tr = OracleTrainer(params, adv, loss, opt, dna, ksize)
    frame, state, psnr = tr.test(td(x), td(y), td(a))
    out['frame'], out['psnr'] = frame.numpy(), np.float64(psnr)
    if state is not None:
        out['state'] = state.numpy()
    d = tr.train_d(td(x), td(y), td(a), return_all=True)
    out['d_loss'] = np.float64(d['d_loss'])
    out['d_direct_loss'] = np.float64(d['discriminator_direct_loss'])
    out['d_gen_loss'] = np.float64(d['discriminator_gen_loss'])
    out['d_out_gen'], out['d_out_real'] = d['d_out_gen'].numpy(), d['d_out_real'].numpy()
    for k, g in tr.last_grads.items():
        out['dgrad_norm/' + k] = np.float64(g.norm())
        out['dgrad_sample/' + k] = sample(g)
    g = tr.train_g(td(x), td(y), td(a), td(s), return_all=True)
    for k in ('g_loss', 'g_l2_loss', 'g_adv_loss', 'gdl'):
        if g.get(k) is not None:
            out[k] = np.float64(g[k])
    for k, gr in tr.last_grads.items():
        out['ggrad_norm/' + k] = np.float64(gr.norm())
        out['ggrad_sample/' + k] = sample(gr)
    for k, v in tr.p.items():                    # parameters after 1 D step + 1 G step
        out['param_sample/' + k] = sample(v)
        if opt == 'rmsprop':
            out['param_norm/' + k] = np.float64(v.norm())
    # pre-training step (train.py:114-121: g_pretrain_opt on g_l2_loss) from the initial parameters, own optimizer state
    tp = OracleTrainer(params, adv, loss, opt, dna, ksize)
    out['pretrain_g_loss'] = np.float64(tp.pretrain_g(td(x), td(y), td(a), td(s)))
    for k, gr in tp.last_grads.items():
        out['pretrain_grad_sample/' + k] = sample(gr)
    for k, v in tp.p.items():
        if k.startswith('g/'):
            out['pretrain_param_sample/' + k] = sample(v)
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        out = make_case(name)
        np.savez_compressed(os.path.join(here, name + '.npz'), **out)
        print(name, 'frame', out['frame'].shape, 'g_loss', out.get('g_loss'), 'd_loss', out['d_loss'])


if __name__ == '__main__':
    main()
