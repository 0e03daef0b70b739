#!/usr/bin/env python
"""Times the few-MFLOP conv layers (d/conv6, g/sconv5, g/sconv4) entry by entry through the C ABI: forward, input gradient,
weight gradient and the paired backward launch, 50 back-to-back launches each between two events.
  python tools/time_small_convs.py [--dtype bf16] [--lib other.so]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--lib', default=None)
    args = ap.parse_args()
    from action_conditioned_gans_amd import _lib
    if args.lib:
        _lib._LIB = _lib.Library(args.lib)
    import abi_call
    abi = abi_call.Abi(_lib.get(), 'cuda:0', conv_dtype=_lib.ACG_BF16 if args.dtype == 'bf16' else _lib.ACG_F32)
    dev = torch.device('cuda:0')

    def timed(fn, reps=50):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    for name, (b, h, w, cin, cout, k, s, pad) in [('d/conv6 (D step)', (64, 2, 2, 512, 1, 2, 1, 'SAME')), ('g/sconv5', (32, 4, 4, 16, 5, 4, 1, 'VALID')),
                                                  ('g/sconv4', (32, 8, 8, 32, 16, 3, 2, 'SAME')), ('g/sconv3', (32, 16, 16, 128, 32, 3, 2, 'SAME'))]:
        x = torch.randn(b, h, w, cin, device=dev)
        wt = torch.randn(k, k, cin, cout, device=dev) * 0.1
        y = abi.conv2d_fwd(x, wt, s, pad)
        dy = torch.randn_like(y)
        t_f = timed(lambda: abi.conv2d_fwd(x, wt, s, pad))
        t_d = timed(lambda: abi.conv2d_dgrad(dy, wt, tuple(x.shape), s, pad))
        t_w = timed(lambda: abi.conv2d_wgrad(x, dy, tuple(wt.shape), s, pad))
        print('%-18s %s  fwd %6.1f us  dgrad %6.1f us  wgrad %6.1f us   (eager launches incl. the wrapper\'s allocations)' % (name, args.dtype, t_f, t_d, t_w))


if __name__ == '__main__':
    main()
