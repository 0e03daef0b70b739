"""Per-op parity cases shared by the CPU suite (C oracle vs torch-fp64 restatement) and the
GPU suite (libacgan_hip.so vs torch-fp64 restatement).  Every case takes an ``Abi`` (tests/abi_call.py)
and a tolerance, builds seeded inputs, and returns nothing or raises AssertionError."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T


def _rng(seed):
    return torch.Generator().manual_seed(seed)


def randn(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=_rng(seed), dtype=torch.float64) * scale).float()


def uniform(shape, seed):
    return (torch.rand(shape, generator=_rng(seed), dtype=torch.float64) * 2 - 1).float()


def close(got, want, tol, what, elementwise=True):
    """Two bars.  (1) max abs error relative to the reference tensor's scale (north_star: 1e-3 rel, fp32).  (2) element
    by element: |err| <= 4 tol |want| + 2 tol rms(want) - relative to each element, with an absolute floor at the
    tensor's RMS instead of its maximum (accumulation error scales with the sum of |products|, so a cancelled, near-zero
    output cannot be held to its own magnitude); outliers, scale or sign errors confined to small elements fail here
    while passing (1)."""
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, '%s: shape %s vs %s' % (what, tuple(got.shape), tuple(want.shape))
    assert torch.isfinite(got).all(), '%s: non-finite values' % what
    scale = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item() / scale
    assert err <= tol, '%s: rel err %.3e > %.1e (scale %.3e)' % (what, err, tol, scale)
    if elementwise and tol > 0 and want.numel() > 1:
        rms = max(want.pow(2).mean().sqrt().item(), 1e-30)
        bad = (got - want).abs() > 4 * tol * want.abs() + 2 * tol * rms
        assert not bad.any(), '%s: %d of %d elements beyond the elementwise bar (tol %.1e, rms %.3e)' % (what, int(bad.sum()), want.numel(), tol, rms)
    return err


# (B, H, W, Cin, Cout, k, stride, padding)
CONV_SHAPES = [
    (2, 64, 64, 3, 32, 5, 2, 'SAME'),      # g/conv1 (DNA)
    (2, 32, 32, 32, 64, 5, 2, 'SAME'),     # g/conv2
    (2, 8, 8, 128, 256, 5, 2, 'SAME'),     # g/conv4
    (2, 64, 64, 6, 64, 5, 2, 'SAME'),      # d/conv1
    (2, 16, 16, 138, 128, 5, 2, 'SAME'),   # d/conv3 (128 + 10 action channels)
    (3, 4, 4, 256, 512, 5, 2, 'SAME'),     # d/conv5, odd batch
    (2, 2, 2, 512, 1, 2, 1, 'SAME'),       # d/conv6
    (2, 16, 16, 128, 32, 3, 2, 'SAME'),    # g/sconv3
    (2, 8, 8, 32, 16, 3, 2, 'SAME'),       # g/sconv4
    (2, 4, 4, 16, 5, 4, 1, 'VALID'),       # g/sconv5
    (1, 7, 9, 5, 7, 3, 1, 'SAME'),         # ragged: odd sizes, stride 1
    (2, 9, 7, 4, 33, 5, 2, 'SAME'),        # ragged: odd spatial, stride 2, Cout not a tile multiple
    (1, 5, 5, 2, 3, 5, 1, 'VALID'),        # single output pixel
]
CONV_SHAPES_SMALL = [CONV_SHAPES[i] for i in (0, 4, 6, 9, 10, 11, 12)]

# (B, IH, IW, Cin, Cout, k, stride)   TF conv2d_transpose SAME
DECONV_SHAPES = [
    (2, 4, 4, 266, 128, 5, 2),     # g/tconv1 (DNA)
    (2, 8, 8, 128, 128, 5, 2),     # g/tconv2
    (2, 32, 32, 128, 25, 5, 2),    # g/tconv4 (DNA, k=5)
    (2, 32, 32, 64, 3, 5, 2),      # g/tconv4 (plain)
    (2, 4, 4, 522, 256, 5, 2),     # g/tconv1 (plain)
    (1, 3, 5, 6, 7, 5, 2),         # ragged
    (1, 4, 4, 8, 36, 5, 2),        # DNA k=6 logits
    (2, 3, 3, 4, 5, 3, 1),         # stride 1
]
DECONV_SHAPES_SMALL = [DECONV_SHAPES[i] for i in (0, 3, 5, 7)]


def case_conv(abi, shape, tol, seed=0):
    b, h, w, cin, cout, k, s, pad = shape
    x = uniform((b, h, w, cin), seed)
    wt = randn((k, k, cin, cout), seed + 1, 0.1)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y_ref = T.conv2d(xd, wd, s, pad)
    dy = randn(tuple(y_ref.shape), seed + 2)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, [xd, wd], dy.double())
    dev = abi.device
    xg, wg, dyg = x.to(dev), wt.to(dev), dy.to(dev)
    tag = 'conv%s' % (shape,)
    close(abi.conv2d_fwd(xg, wg, s, pad), y_ref, tol, tag + ' fwd')
    close(abi.conv2d_dgrad(dyg, wg, tuple(x.shape), s, pad), dx_ref, tol, tag + ' dgrad')
    close(abi.conv2d_wgrad(xg, dyg, tuple(wt.shape), s, pad), dw_ref, tol, tag + ' wgrad')
    # accumulate semantics: dw = 0.5*dw0 + grad
    dw0 = randn(tuple(wt.shape), seed + 3).to(dev)
    got = abi.conv2d_wgrad(xg, dyg, tuple(wt.shape), s, pad, dw=dw0.clone(), accumulate=0.5)
    close(got, 0.5 * dw0.double().cpu() + dw_ref, tol, tag + ' wgrad accumulate')


MERGED_LAYERS = [   # (x shape, w shape, stride, padding, transposed, merged?)
    ((4, 16, 16, 6), (5, 5, 6, 64), 2, 'SAME', False, True),       # d/conv1-like: 3 x 3 window, N = 24
    ((2, 16, 12, 3), (3, 3, 3, 16), 2, 'SAME', False, True),       # 3 x 3 filter, pad 0 / 1: a 2 x 2 window (8 output channels would take the direct kernels)
    ((2, 10, 6, 8), (5, 5, 8, 32), 2, 'SAME', False, True),        # 8 channels: all 32 columns
    ((32, 32, 40, 1), (5, 5, 1, 16), 2, 'SAME', False, True),      # one channel: N = 4
    ((3, 12, 20, 1), (5, 5, 1, 16), 2, 'SAME', False, None),       # the same with a few MFLOP: float32 stays with the class-wise kernel (bf16 has no direct kernels: merged)
    ((2, 8, 8, 16), (5, 5, 3, 16), 2, None, True, True),           # transposed layer with a 3-channel output (a plain generator's last)
    ((2, 12, 12, 4), (5, 5, 4, 8), 2, 'VALID', False, False),      # output extent is not half the input: the class-wise kernel
    ((2, 9, 9, 4), (5, 5, 4, 8), 2, 'SAME', False, False),         # odd extents
    ((2, 16, 16, 12), (5, 5, 12, 8), 2, 'SAME', False, False),     # 48 columns would not fit the narrow tile
    ((2, 16, 16, 6), (5, 5, 6, 8), 1, 'SAME', False, False),       # stride 1
]


def case_merged_dgrad(abi, tol):
    """The merged input gradient (acgan_hip.h at acg_conv2d_dgrad: the four stride-parity classes of a stride-2 layer with at
    most 8 input channels as ONE stride-1 contraction with a derived filter) against the float64 gradient, next to shapes
    that must NOT take it; the workspace query tells which path a shape takes (the derived filter lives there)."""
    from action_conditioned_gans_amd import _lib as L
    import ctypes
    dev = abi.device
    r = (lambda t: t.bfloat16().float()) if abi.half else (lambda t: t)
    for i, (xs, ws_, stride, padding, transposed, merged) in enumerate(MERGED_LAYERS):
        merged = abi.half if merged is None else merged
        w = randn(ws_, 910 + i, 0.1)
        if transposed:
            x = uniform(xs, 900 + i)
            want = T.conv2d_transpose(r(x).double(), r(w).double(), stride, 'SAME')
            d = abi._adj(xs, ws_, stride)
            got = abi.deconv2d_fwd(x.to(dev), w.to(dev), stride)
        else:
            xd = uniform(xs, 900 + i).double().requires_grad_(True)
            y = T.conv2d(xd, r(w).double(), stride, padding)
            dy = randn(tuple(y.shape), 920 + i)
            want, = torch.autograd.grad(y, [xd], r(dy).double())
            b, h, wd, c = xs
            d = abi.desc(b, h, wd, c, ws_[0], ws_[1], ws_[3], stride, padding)
            got = abi.conv2d_dgrad(dy.to(dev), w.to(dev), xs, stride, padding)
        if abi.half:
            d.in_pitch = d.out_pitch = 0
        splits = abi.lib.conv2d_splits(ctypes.byref(d), L.CONV_DGRAD, abi.conv_dtype)
        wsb = abi.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_DGRAD, abi.conv_dtype)
        numel = want.numel() if not abi.half else want.numel() // want.shape[-1] * ((want.shape[-1] + 7) // 8 * 8)
        took = wsb != (splits * numel * 4 if splits > 1 else 0)           # the derived filter sits behind the slabs
        assert took == merged, ('merged layer %d' % i, took, merged, splits, wsb)
        close(got, want, tol, 'merged dgrad layer %d%s' % (i, ' (merged)' if merged else ''))


def case_dgrad_channel_limit(abi, tol):
    """acg_conv_desc dgrad_c / adj_dgrad_c: the input gradient of a layer fed by an action-concatenated map (d/conv3: 128 + 10
    channels at a pitch of 140; g/tconv1: 256 + 10 at 268) computes the feature channels only.  The limited columns must equal
    the unlimited result to rounding (fewer column tiles can mean another split-K factor), the others must not be written; single
    entries and the paired launch."""
    dev = abi.device
    for i, (xs, ws_, stride, transposed, keep) in enumerate([
            ((4, 16, 16, 140), (5, 5, 138, 128), 2, False, 128),      # d/conv3-like: 138 logical channels at pitch 140
            ((2, 8, 8, 44), (3, 3, 42, 24), 1, False, 32),            # two of three 16.. column tiles on the narrow tile
            ((4, 4, 4, 268), (5, 5, 128, 266), 2, True, 256),         # g/tconv1-like (transposed: dx is the adjoint's out side)
            ((2, 8, 8, 24), (5, 5, 16, 20), 2, True, 12),
            ((2, 8, 8, 8), (5, 5, 8, 16), 2, False, 4)]):               # a shape the merged input gradient would take: the limit wins
        cphys, clog = xs[3], (ws_[3] if transposed else ws_[2])
        x = torch.zeros(xs)
        x[..., :clog] = uniform(xs[:3] + (clog,), 1100 + i)
        w = randn(ws_, 1110 + i, 0.1).to(dev)
        if transposed:
            dy = randn((xs[0], xs[1] * stride, xs[2] * stride, ws_[2]), 1120 + i).to(dev)
            full = abi.deconv2d_dgrad(dy, w, xs, stride)
            part = abi.deconv2d_dgrad(dy, w, xs, stride, grad_c=keep)
        else:
            yshape = tuple(T.conv2d(x[..., :clog].double(), w.double().cpu(), stride, 'SAME').shape)
            dy = randn(yshape, 1120 + i).to(dev)
            full = abi.conv2d_dgrad(dy, w, xs, stride, 'SAME')
            part = abi.conv2d_dgrad(dy, w, xs, stride, 'SAME', grad_c=keep)
        tag = 'dgrad channel limit %d' % i
        close(part[..., :keep], full[..., :keep], tol, tag + ': limited columns')
        assert (part[..., keep:] == 0).all(), tag + ': columns beyond the limit were written'
        assert float(full[..., keep:clog].abs().max()) > 0, tag
        if not abi.half:
            dxp, dwp = abi.bwd_pair(x.to(dev), dy, w, stride, 'SAME', transposed, grad_c=keep)
            dxf, dwf = abi.bwd_pair(x.to(dev), dy, w, stride, 'SAME', transposed)
            close(dxp[..., :keep], dxf[..., :keep], tol, tag + ' (pair)')
            assert (dxp[..., keep:] == 0).all(), tag + ' (pair): columns beyond the limit were written'
            assert torch.equal(dwp, dwf), tag + ' (pair): the weight gradient must not change'


def case_conv_bf16(abi, shape, tol, tol_w, seed=0, transposed=False):
    """ACG_BF16: tensors stored as bfloat16 (RNE), products exact, fp32 accumulation - so the reference is the fp64
    conv of the bf16-rounded operands; bf16 outputs (y, dx) carry one more rounding (``tol``), the fp32 weight
    gradient stays at accumulation-order level (``tol_w``)."""
    r16 = lambda t: t.bfloat16().float()
    if transposed:
        b, ih, iw, cin, cout, k, s = shape
        x, wt = uniform((b, ih, iw, cin), seed), randn((k, k, cout, cin), seed + 1, 0.1)
        fwd = lambda xx, ww: T.conv2d_transpose(xx, ww, s, 'SAME')
    else:
        b, h, w, cin, cout, k, s, pad = shape
        x, wt = uniform((b, h, w, cin), seed), randn((k, k, cin, cout), seed + 1, 0.1)
        fwd = lambda xx, ww: T.conv2d(xx, ww, s, pad)
    y_shape = tuple(fwd(x.double(), wt.double()).shape)
    dy = randn(y_shape, seed + 2)
    xr, wr, dyr = r16(x).double(), r16(wt).double(), r16(dy).double()
    y_ref = fwd(xr, wr)
    xg_, wg_ = xr.clone().requires_grad_(True), wr.clone().requires_grad_(True)
    dx_ref, = torch.autograd.grad(fwd(xg_, wr), [xg_], dyr)          # dgrad uses rounded dy and w
    dw_ref, = torch.autograd.grad(fwd(xr, wg_), [wg_], dyr)          # wgrad uses rounded x and dy
    dev = abi.device
    xg, wg, dyg = x.to(dev), wt.to(dev), dy.to(dev)
    tag = 'bf16 %s%s' % ('deconv' if transposed else 'conv', shape)
    if transposed:
        close(abi.deconv2d_fwd(xg, wg, s), y_ref, tol, tag + ' fwd')
        close(abi.deconv2d_dgrad(dyg, wg, tuple(x.shape), s), dx_ref, tol, tag + ' dgrad')
        close(abi.deconv2d_wgrad(xg, dyg, tuple(wt.shape), s), dw_ref, tol_w, tag + ' wgrad')
    else:
        close(abi.conv2d_fwd(xg, wg, s, pad), y_ref, tol, tag + ' fwd')
        close(abi.conv2d_dgrad(dyg, wg, tuple(x.shape), s, pad), dx_ref, tol, tag + ' dgrad')
        close(abi.conv2d_wgrad(xg, dyg, tuple(wt.shape), s, pad), dw_ref, tol_w, tag + ' wgrad')


def case_bwd_pair_bf16(abi, tol, tol_w):
    """bf16 acg_(de)conv2d_bwd_pair (conv_pair_bf16: both gradients of a layer out of one grid) against the separate
    entries, for every tile pairing the planner produces (64/64, 64/128, 128/64, 128/128) and an accumulating dw."""
    dev = abi.device
    layers = [((2, 16, 16, 32), (5, 5, 32, 64), 2, 'SAME', False),       # 64x64 / 64x64
              ((8, 32, 32, 64), (5, 5, 64, 128), 2, 'SAME', False),      # larger, split wgrad
              ((4, 9, 7, 24), (3, 3, 24, 40), 1, 'SAME', False),         # ragged edges
              ((2, 8, 8, 64), (5, 5, 32, 64), 2, None, True),            # transposed layer (FWD on the adjoint + WGRAD)
              ((32, 32, 32, 128), (5, 5, 128, 128), 1, 'SAME', False)]   # 128x128 tiles on both sides
    for i, (xs, ws_, stride, padding, transposed) in enumerate(layers):
        x = uniform(xs, 170 + i).to(dev)
        w = randn(ws_, 180 + i, 0.1).to(dev)
        if transposed:
            dy = randn((xs[0], xs[1] * stride, xs[2] * stride, ws_[2]), 190 + i).to(dev)
            dx_ref = abi.deconv2d_dgrad(dy, w, tuple(xs), stride)
            dw_plain = lambda dw, acc: abi.deconv2d_wgrad(x, dy, ws_, stride, dw=dw, accumulate=acc)    # noqa: E731
        else:
            d = abi.desc(xs[0], xs[1], xs[2], ws_[2], ws_[0], ws_[1], ws_[3], stride, padding)
            dy = randn((xs[0], d.out_h, d.out_w, ws_[3]), 190 + i).to(dev)
            dx_ref = abi.conv2d_dgrad(dy, w, tuple(xs), stride, padding)
            dw_plain = lambda dw, acc: abi.conv2d_wgrad(x, dy, ws_, stride, padding, dw=dw, accumulate=acc)    # noqa: E731
        init = randn(ws_, 200 + i).to(dev)
        for acc in (0.0, 1.0):
            dw_ref = dw_plain(init.clone(), acc)
            dx, dw = abi.bwd_pair(x, dy, w, stride, padding, transposed, accumulate=acc, dw=init.clone())
            abi.sync()
            tag = 'bf16 pair layer %d acc %g' % (i, acc)
            close(dx, dx_ref.double().cpu(), tol, tag + ' dx')
            close(dw, dw_ref.double().cpu(), tol_w, tag + ' dw')


STATS_LAYERS = [   # (x shape, w shape, stride, padding, transposed, groups, act)
    ((4, 16, 16, 8), (5, 5, 8, 32), 2, 'SAME', False, 1, 'relu'),            # (16 output channels would take the direct kernels: no statistics)
    ((8, 32, 32, 3), (5, 5, 3, 32), 2, 'SAME', False, 1, 'relu'),        # g/conv1-like: N = 32 (the 128 x 32 tile in fp32)
    ((8, 32, 32, 6), (5, 5, 6, 64), 2, 'SAME', False, 2, 'lrelu'),       # d/conv1-like: two groups (fake | real)
    ((6, 18, 14, 12), (3, 3, 12, 40), 1, 'SAME', False, 1, None),        # ragged rows and columns
    ((8, 16, 16, 32), (5, 5, 24, 32), 2, None, True, 1, 'relu'),         # transposed layer: four stride classes
    ((32, 32, 32, 64), (5, 5, 64, 128), 2, 'SAME', False, 2, 'lrelu'),   # d/conv2-like
]


def case_conv_bn_stats(abi, tol, tol_stat, min_fused=1, layers=None):
    """acg_(de)conv2d_fwd_stats + acg_bn_act_fwd_partials against the float64 conv followed by slim batch_norm + activation:
    the conv output, the normalised output and the saved mean / rstd.  ``min_fused``: how many of the layers must actually
    take the fused path (acg_conv2d_stats_blocks > 0) - the split ones legitimately do not."""
    fused = 0
    r = (lambda t: t.bfloat16().float()) if abi.half else (lambda t: t)
    for i, (xs, ws_, stride, padding, transposed, groups, act) in enumerate(layers or STATS_LAYERS):
        x, w = uniform(xs, 300 + i), randn(ws_, 310 + i, 0.1)
        c = ws_[2] if transposed else ws_[3]
        beta = randn((c,), 320 + i, 0.5)
        got = abi.conv_bn_fused(x.to(abi.device), w.to(abi.device), beta.to(abi.device), stride, padding, act, groups, transposed)
        if got is None:
            continue
        fused += 1
        conv, y, mean, rstd = got
        abi.sync()
        xr, wr = r(x).double(), r(w).double()
        ref = T.conv2d_transpose(xr, wr, stride, 'SAME') if transposed else T.conv2d(xr, wr, stride, padding)
        tag = 'conv+bn stats layer %d' % i
        close(conv, ref, tol, tag + ' conv')
        stored = r(ref.float()).double()                      # BatchNorm sees the tensor as stored
        rows = stored.reshape(groups, -1, c)
        m, v = rows.mean(1), rows.var(1, unbiased=False)
        close(mean, m.reshape(-1), tol_stat, tag + ' mean')
        close(rstd, (1.0 / torch.sqrt(v + 1e-3)).reshape(-1), tol_stat, tag + ' rstd')
        pre = (rows - m[:, None, :]) / torch.sqrt(v[:, None, :] + 1e-3) + beta.double()
        want = {'relu': torch.relu, 'lrelu': lambda t: torch.where(t > 0, t, 0.2 * t), None: lambda t: t}[act](pre).reshape(stored.shape)
        close(y, want, tol * 4 if abi.half else tol, tag + ' y')
    assert fused >= min_fused, 'only %d of %d layers took the fused path' % (fused, len(layers or STATS_LAYERS))


HANDOFF_LAYERS = [   # (x shape, w shape, stride, padding, transposed, groups, act): small layers the planner splits over K
    ((8, 8, 8, 64), (5, 5, 64, 128), 2, 'SAME', False, 1, 'relu'),           # g/conv3-like
    ((16, 8, 8, 128), (5, 5, 128, 256), 2, 'SAME', False, 2, 'lrelu'),       # d/conv4-like: two groups
    ((8, 4, 4, 256), (5, 5, 256, 512), 2, 'SAME', False, 2, 'lrelu'),        # d/conv5-like: 32 rows
    ((8, 4, 4, 256), (5, 5, 128, 256), 2, None, True, 1, 'relu'),            # g/tconv1-like (transposed)
    ((4, 8, 8, 138), (5, 5, 138, 128), 2, 'SAME', False, 1, 'relu'),         # 138 gathered channels
    ((6, 6, 6, 64), (3, 3, 64, 44), 1, 'SAME', False, 1, None),              # ragged rows, 44 channels (bf16: pitch 48)
    ((32, 32, 32, 32), (5, 5, 32, 64), 2, 'SAME', False, 1, 'relu'),         # g/conv2 at config-2 size: 8192 rows (two-launch BatchNorm: rows layout; float32 splits it)
]
# bf16 only (test_conv_bn_stats_wide_bf16): large enough for the 256 x 128 LDS-DMA kernel (conv_bf16_glds.h), whose tiles leave
# the partials: forward with one group and with two (D on [fake ; real]), and a transposed layer (four stride classes)
STATS_LAYERS_WIDE = [
    ((48, 64, 64, 64), (5, 5, 64, 128), 2, 'SAME', False, 1, 'relu'),
    ((48, 64, 64, 64), (5, 5, 64, 128), 2, 'SAME', False, 2, 'lrelu'),
    ((12, 32, 32, 160), (5, 5, 128, 160), 2, None, True, 1, 'relu'),
]


def tile_rows(abi, which, b, h, w, cin, k, cout, stride, padding):
    """Rows of the tile the planner runs this contraction on (acg_conv2d_tile)."""
    import ctypes
    d = abi.desc(b, h, w, cin, k, k, cout, stride, padding)
    rows, cols = ctypes.c_int32(0), ctypes.c_int32(0)
    assert abi.lib.conv2d_tile(ctypes.byref(d), which, abi.conv_dtype, ctypes.byref(rows), ctypes.byref(cols)) > 0
    return rows.value


def case_slab_handoff(abi, tol, min_quads=0, min_rows=5):
    """Split-K hand-off, forward and backward, in both slab layouts (acgan_hip.h ACG_SLABS_ROWS / ACG_SLABS_QUADS) against
    the separate-reduction path on the same tensors: the conv output written back must be bit-identical (same slabs, same
    summation order and rounding), the BatchNorm results equal to rounding level (another instantiation of the same
    kernel); the layout acg_bn_slabs_layout asks for must be QUADS exactly where the one-launch kernels run."""
    dev = abi.device
    quads = rows_l = bwd_done = pair_done = 0
    for i, (xs, ws_, stride, padding, transposed, groups, act) in enumerate(HANDOFF_LAYERS):
        x, w = uniform(xs, 700 + i).to(dev), randn(ws_, 710 + i, 0.1).to(dev)
        c = ws_[2] if transposed else ws_[3]
        beta = randn((c,), 720 + i, 0.5).to(dev)
        conv_ref = abi.deconv2d_fwd(x, w, stride) if transposed else abi.conv2d_fwd(x, w, stride, padding)
        st = abi.to16(conv_ref) if abi.half else conv_ref
        y_ref, mean_ref, rstd_ref = abi.bn_act_fwd(st, beta, act, groups=groups, c=c)
        y_ref = y_ref[..., :c].float()
        asked = None
        for layout in (None, 0):
            got = abi.conv_bn_handoff(x, w, beta, stride, padding, act, groups, transposed, layout=layout)
            if got is None:          # the planner does not split this layer in this arithmetic
                break
            conv, y, mean, rstd, used = got
            asked = used if layout is None else asked
            tag = 'hand-off layer %d layout %d' % (i, used)
            assert torch.equal(conv, conv_ref), tag + ': conv output written back differs from the separate reduction'
            close(mean, mean_ref, tol, tag + ' mean'); close(rstd, rstd_ref, tol * 4, tag + ' rstd')
            close(y, y_ref, 8e-3 if abi.half else tol * 4, tag + ' y')
        if asked is None:
            continue
        rows_per_group = conv_ref.numel() // c // groups
        # round 4: the one-launch grid kernels take slabs laid out like the tensor (ROWS) wherever their grid is resident - every
        # layer here with a channel count that is a multiple of 4; the quad layout remains for the resident kernels behind them
        assert asked == (0 if c % 4 == 0 else (1 if rows_per_group <= 2048 and c % 4 == 0 else 0)), (i, asked, rows_per_group, c)
        quads += asked == 1; rows_l += asked == 0
        # backward: this BatchNorm's dy is the split input gradient of a following 5x5 / stride-2 layer
        if transposed or c % 8 or conv_ref.shape[1] % 2:
            continue
        w2 = randn((5, 5, c, 2 * c), 730 + i, 0.05).to(dev)
        dy2 = randn((conv_ref.shape[0], conv_ref.shape[1] // 2, conv_ref.shape[2] // 2, 2 * c), 740 + i).to(dev)
        for pair in (False, True):
            got = [abi.dgrad_bn_bwd_handoff(conv_ref, beta, mean_ref, rstd_ref, act, dy2, w2, 2, 'SAME', groups, pair_x=y_ref if pair else None,
                                            layout=lay) for lay in (None, 0)]
            if got[0] is None:
                continue
            dyb = abi.conv2d_dgrad(dy2, w2, tuple(conv_ref.shape), 2, 'SAME')
            xs_, dys_ = (abi.to16(conv_ref), abi.to16(dyb)) if abi.half else (conv_ref, dyb)
            dx_ref, dbeta_ref = abi.bn_act_bwd(xs_, dys_, beta, mean_ref, rstd_ref, act, groups=groups)
            for dx, dbeta, used in got:
                tag = 'hand-off bwd layer %d layout %d%s' % (i, used, ' (pair)' if pair else '')
                close(dx, dx_ref[..., :c].float(), 8e-3 if abi.half else tol * 8, tag + ' dx')
                close(dbeta, dbeta_ref, 2e-3 if abi.half else tol * 8, tag + ' dbeta')
            bwd_done += not pair; pair_done += pair
    assert quads >= min_quads and rows_l >= min_rows and bwd_done >= 2 and pair_done >= 2, (quads, rows_l, bwd_done, pair_done)


def case_conv_bn_stats_large_mean(abi, tol_stat):
    """the statistics out of the conv epilogue must survive |mean| >> std (a drifted d/conv layer late
    in GAN training).  1x1 convolutions with positive weights over inputs offset by 300: per-channel mean / std of the
    output ~ 1e3, where E[x^2] - E[x]^2 from float32 sums (round 2) has no digits left.  Checked: the saved mean and rstd
    against float64 statistics of the tensor as stored."""
    r = (lambda t: t.bfloat16().float()) if abi.half else (lambda t: t)
    for i, (b, h, w, cin, cout, groups) in enumerate([(8, 32, 32, 8, 64, 1), (16, 16, 16, 16, 32, 2), (4, 24, 20, 8, 40, 1)]):
        x = uniform((b, h, w, cin), 700 + i) + 300.0
        wt = randn((1, 1, cin, cout), 710 + i, 0.1).abs() + 0.02
        beta = randn((cout,), 720 + i, 0.5)
        got = abi.conv_bn_fused(x.to(abi.device), wt.to(abi.device), beta.to(abi.device), 1, 'SAME', 'relu', groups, False)
        assert got is not None, 'large-mean layer %d did not take the fused path' % i
        conv, y, mean, rstd = got
        abi.sync()
        rows = conv.detach().double().cpu().reshape(groups, -1, cout)          # the tensor as the kernel stored it
        m, v = rows.mean(1), rows.var(1, unbiased=False)
        ratio = (m.abs() / v.sqrt()).min().item()
        assert ratio > 100, 'test premise: mean / std = %.1f' % ratio
        close(mean, m.reshape(-1), 1e-6, 'large-mean layer %d mean' % i)
        close(rstd, (1.0 / torch.sqrt(v + 1e-3)).reshape(-1), tol_stat, 'large-mean layer %d rstd' % i)
        assert torch.isfinite(y.double()).all()


def case_bn_large_tensor(abi, tol_stat):
    """BatchNorm at config 5's tensor sizes (4-8 M elements, more than 512 partial blocks per group): forward behind a conv
    epilogue's statistics (acg_conv2d_fwd_stats -> bn_partials_finalize -> apply) and the two-launch backward, against float64
    BatchNorm of the tensor as stored."""
    dev = abi.device
    half = abi.half
    r = (lambda t: t.bfloat16().float()) if half else (lambda t: t)
    cout, groups, act = (32, 2, 'lrelu') if half else (64, 1, 'relu')
    b, h, w, cin = 8, 128, 128, 8
    x, wt = uniform((b, h, w, cin), 800), randn((1, 1, cin, cout), 801, 0.3)
    beta = randn((cout,), 802, 0.5)
    got = abi.conv_bn_fused(x.to(dev), wt.to(dev), beta.to(dev), 1, 'SAME', act, groups, False)
    assert got is not None
    conv, y, mean, rstd = got
    abi.sync()
    rows = conv.detach().double().cpu().reshape(groups, -1, cout)
    m, v = rows.mean(1), rows.var(1, unbiased=False)
    close(mean, m.reshape(-1), tol_stat, 'large bn mean'); close(rstd, (1.0 / torch.sqrt(v + 1e-3)).reshape(-1), tol_stat, 'large bn rstd')
    pre = (rows - m[:, None, :]) / torch.sqrt(v[:, None, :] + 1e-3) + beta.double()
    want = {'relu': torch.relu, 'lrelu': lambda t: torch.where(t > 0, t, 0.2 * t)}[act](pre).reshape(conv.shape)
    close(y, want, 8e-3 if half else 2e-5, 'large bn y')
    # backward on the same stored tensor
    cp = (cout + 7) // 8 * 8 if half else cout
    store = torch.bfloat16 if half else torch.float32
    xs = torch.zeros(b, h, w, cp, dtype=store, device=dev)
    xs[..., :cout] = conv.to(dev).to(store)
    dy = torch.zeros(b, h, w, cp, dtype=store, device=dev)
    dy[..., :cout] = randn((b, h, w, cout), 803).to(dev).to(store)
    dx, dbeta = abi.bn_act_bwd(xs, dy, beta.to(dev), mean, rstd, act, groups)
    abi.sync()
    xd = xs[..., :cout].double().cpu().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    dx64, db64 = torch.autograd.grad(_bn_ref(xd, bd, act, groups), [xd, bd], dy[..., :cout].double().cpu())
    close(dx[..., :cout].float(), dx64, 8e-3 if half else 2e-5, 'large bn dx'); close(dbeta, db64, 3e-4, 'large bn dbeta')
    if half:
        assert (dx[..., cout:] == 0).all() and (y.shape[-1] == cout or (y[..., cout:] == 0).all())


# Every conv / deconv layer of the DNA generator and the discriminator at BASELINE config 2's per-GPU size (batch 32, 64x64;
# Appendix B of SURVEY.md) plus config 5's two largest (128x128): (kind, B, H, W, Cin, Cout, k, stride, padding)
BASELINE_LAYERS = [
    ('c', 32, 64, 64, 3, 32, 5, 2, 'SAME'), ('c', 32, 32, 32, 32, 64, 5, 2, 'SAME'), ('c', 32, 16, 16, 64, 128, 5, 2, 'SAME'),
    ('c', 32, 8, 8, 128, 256, 5, 2, 'SAME'), ('d', 32, 4, 4, 266, 128, 5, 2, 'SAME'), ('d', 32, 8, 8, 128, 128, 5, 2, 'SAME'),
    ('c', 32, 16, 16, 128, 32, 3, 2, 'SAME'), ('c', 32, 8, 8, 32, 16, 3, 2, 'SAME'), ('c', 32, 4, 4, 16, 5, 4, 1, 'VALID'),
    ('d', 32, 16, 16, 128, 128, 5, 2, 'SAME'), ('d', 32, 32, 32, 128, 25, 5, 2, 'SAME'),
    ('c', 64, 64, 64, 6, 64, 5, 2, 'SAME'), ('c', 64, 32, 32, 64, 128, 5, 2, 'SAME'), ('c', 64, 16, 16, 138, 128, 5, 2, 'SAME'),
    ('c', 64, 8, 8, 128, 256, 5, 2, 'SAME'), ('c', 64, 4, 4, 256, 512, 5, 2, 'SAME'), ('c', 64, 2, 2, 512, 1, 2, 1, 'SAME'),
    ('d', 32, 64, 64, 128, 121, 5, 2, 'SAME'), ('c', 32, 128, 128, 6, 64, 5, 2, 'SAME'),
]


def case_conv_adjoint_identities(abi, tol):
    """Size-independent properties at BASELINE's full per-GPU sizes, where no float64 oracle run fits a test: a conv layer is
    linear in x and in w, so for random x, w, dy
        <fwd(x, w), dy>  =  <x, dgrad(dy, w)>  =  <w, wgrad(x, dy)>
    - one identity ties the three contractions of a layer (and the split-K, tile and pairing choices the planner makes at
    these sizes) together.  Inner products in float64 on the device.  bf16: outputs are rounded once (2^-9 per element,
    unbiased), which the sums average out; the bar there is 3e-3."""
    dev = abi.device
    dot = lambda a, b: float((a.double() * b.double()).sum().item())      # noqa: E731
    for i, (kind, b, h, w, cin, cout, k, s_, pad) in enumerate(BASELINE_LAYERS):
        x = uniform((b, h, w, cin), 1000 + i).to(dev)
        if kind == 'c':
            wt = randn((k, k, cin, cout), 1100 + i, 0.1).to(dev)
            y = abi.conv2d_fwd(x, wt, s_, pad)
            dy = randn(tuple(y.shape), 1200 + i).to(dev)
            dx = abi.conv2d_dgrad(dy, wt, tuple(x.shape), s_, pad)
            dw = abi.conv2d_wgrad(x, dy, tuple(wt.shape), s_, pad)
        else:
            wt = randn((k, k, cout, cin), 1100 + i, 0.1).to(dev)
            y = abi.deconv2d_fwd(x, wt, s_)
            dy = randn(tuple(y.shape), 1200 + i).to(dev)
            dx = abi.deconv2d_dgrad(dy, wt, tuple(x.shape), s_)
            dw = abi.deconv2d_wgrad(x, dy, tuple(wt.shape), s_)
        abi.sync()
        if abi.half:      # the kernels saw the operands rounded to bf16
            x, wt, dy = x.bfloat16().float(), wt.bfloat16().float(), dy.bfloat16().float()
        a, bq, c = dot(y, dy), dot(x, dx), dot(wt, dw)
        scale = float(y.double().norm().item() * dy.double().norm().item())
        tag = 'layer %d %s' % (i, (kind, b, h, w, cin, cout, k, s_))
        assert abs(a - bq) <= tol * scale, '%s: <y,dy> %.6e vs <x,dx> %.6e (scale %.3e)' % (tag, a, bq, scale)
        assert abs(a - c) <= tol * scale, '%s: <y,dy> %.6e vs <w,dw> %.6e (scale %.3e)' % (tag, a, c, scale)
        del x, wt, y, dy, dx, dw


def case_full_size_properties(abi):
    """DNA stencil and BatchNorm at BASELINE's full per-GPU sizes through properties that need no oracle:
      DNA   softmax weights sum to one: a constant image comes back unchanged wherever the k x k window lies inside the
            frame (no border renormalisation, models.py:60-72); the frame is linear in the image; the logits' gradient
            sums to zero over the taps of every pixel (softmax), and dbias is its sum over pixels;
      BN    (no activation) every output channel has mean beta and variance var / (var + eps); the input gradient sums to
            zero over the batch and is orthogonal to the normalised input, per channel."""
    dev = abi.device
    for (b, hw, k, dt) in [(32, 64, 5, torch.float32), (32, 64, 5, torch.bfloat16), (32, 128, 11, torch.bfloat16), (8, 128, 11, torch.float32)]:
        kk = k * k
        lp = (kk + 7) // 8 * 8 if dt == torch.bfloat16 else kk
        lg = torch.zeros(b, hw, hw, lp, dtype=dt, device=dev)
        lg[..., :kk] = (randn((b, hw, hw, kk), 1300 + k).to(dev) * 2).to(dt)
        bias = randn((kk,), 1301, 0.3).to(dev)
        const = torch.tensor([0.7, -0.2, 0.4], device=dev).expand(b, hw, hw, 3).contiguous()
        out = abi.dna_fwd(lg, const, k, bias=bias)
        p = (k - 1) // 2
        inner = out[:, p:hw - (k - 1 - p), p:hw - (k - 1 - p)]
        assert float((inner - const[:, p:hw - (k - 1 - p), p:hw - (k - 1 - p)]).abs().max()) <= 2e-6, ('dna constant image', k, dt)
        i1, i2 = uniform((b, hw, hw, 3), 1302).to(dev), uniform((b, hw, hw, 3), 1303).to(dev)
        lin = abi.dna_fwd(lg, 0.3 * i1 - 1.7 * i2, k, bias=bias)
        ref = 0.3 * abi.dna_fwd(lg, i1, k, bias=bias) - 1.7 * abi.dna_fwd(lg, i2, k, bias=bias)
        assert float((lin - ref).abs().max()) <= 1e-5, ('dna linearity', k, dt)
        dout = randn((b, hw, hw, 3), 1304).to(dev)
        dl, dbias = abi.dna_bwd(lg, i1, dout, k, bias=bias, want_dbias=True)
        abi.sync()
        dlf = dl[..., :kk].float()
        tap_sum = dlf.sum(-1).abs().max().item()
        assert tap_sum <= (2e-2 if dt == torch.bfloat16 else 1e-5) * max(dlf.abs().max().item(), 1e-6) * kk ** 0.5, ('dna sum_t dlogits', k, dt, tap_sum)
        if dt == torch.float32:        # (bf16: dbias sums the float32 values before they are rounded into dlogits)
            assert float((dbias - dlf.double().sum((0, 1, 2)).float()).abs().max()) <= 1e-4 * max(float(dbias.abs().max()), 1e-6) + 1e-5, ('dna dbias', k)
        if dt == torch.bfloat16:
            assert (dl[..., kk:] == 0).all()
        del lg, out, lin, ref, dl
    for lead, c, groups, dt in [((64, 32, 32), 64, 2, torch.float32), ((32, 64, 64), 128, 1, torch.bfloat16), ((64, 64, 64), 64, 2, torch.bfloat16)]:
        x = (randn(lead + (c,), 1400, 1.5) + 0.7).to(dev).to(dt)
        beta = randn((c,), 1401, 0.3).to(dev)
        y, mean, rstd = abi.bn_act_fwd(x, beta, None, groups, y_dtype=None)
        yg = y.float().reshape(groups, -1, c).double()
        xg = x.float().reshape(groups, -1, c).double()
        var = xg.var(1, unbiased=False)
        tolm = 2e-3 if dt == torch.bfloat16 else 1e-5
        assert float((yg.mean(1) - beta.double()).abs().max()) <= tolm, ('bn mean', lead, dt)
        assert float((yg.var(1, unbiased=False) - var / (var + 1e-3)).abs().max()) <= 4 * tolm, ('bn var', lead, dt)
        dy = randn(lead + (c,), 1402).to(dev).to(dt)
        dx, dbeta = abi.bn_act_bwd(x, dy, beta, mean, rstd, None, groups)
        abi.sync()
        dxg = dx.float().reshape(groups, -1, c).double()
        xh = (xg - xg.mean(1, keepdim=True)) / torch.sqrt(var + 1e-3)[:, None, :]
        n = dxg.shape[1]
        scale = float(dxg.abs().mean()) * n
        tolb = 2e-3 if dt == torch.bfloat16 else 1e-5
        assert float(dxg.sum(1).abs().max()) <= tolb * scale, ('bn sum dx', lead, dt, float(dxg.sum(1).abs().max()) / scale)
        assert float((dxg * xh).sum(1).abs().max()) <= tolb * scale, ('bn sum dx xhat', lead, dt)
        assert float((dbeta.double() - dy.float().double().reshape(-1, c).sum(0)).abs().max()) <= 1e-4 * n ** 0.5 * groups, ('bn dbeta', lead, dt)
        del x, y, dy, dx


def case_dna_second(abi, tol):
    """acg_dna_fwd out2 / acg_dna_bwd dout2 (the frame's second home, train.py:63-66): forward writes the frame also into channels
    [3, 6) of an 8-pitched tensor (float32 and bfloat16), leaving the other channels alone; backward with a gradient window equals
    backward of the explicit sum."""
    dev = abi.device
    for (b, h, w, c, k) in [(2, 16, 12, 3, 5), (1, 9, 20, 3, 6), (1, 8, 8, 3, 11)]:
        lg, img = randn((b, h, w, k * k), 500 + k).to(dev), uniform((b, h, w, c), 510 + k).to(dev)
        bias = randn((k * k,), 520 + k, 0.3).to(dev)
        for dt in ([torch.float32, torch.bfloat16] if dev.type == 'cuda' else [torch.float32]):
            ref = abi.dna_fwd(lg, img, k, bias=bias)
            # (a) the discriminator-input layout, concat(image, frame) at a pitch of 8: the whole pixel is written
            out2 = torch.full((b, h, w, 8), 7.0, dtype=dt, device=dev)
            out = abi.dna_fwd(lg, img, k, bias=bias, out2=out2, out2_off=3)
            abi.sync()
            assert torch.equal(out.cpu(), ref.cpu()), 'dna second output changed the frame'
            assert torch.equal(out2[..., 3:6].float().cpu(), ref.to(dt).float().cpu()), 'dna second output %s' % dt
            assert torch.equal(out2[..., :3].float().cpu(), img.to(dt).float().cpu()) and bool((out2[..., 6:] == 0).all()), 'dna: concat(image, frame) pixel'
            # (b) any other layout: the frame channels only
            out2 = torch.full((b, h, w, 12), 7.0, dtype=dt, device=dev)
            abi.dna_fwd(lg, img, k, bias=bias, out2=out2, out2_off=5)
            abi.sync()
            assert torch.equal(out2[..., 5:8].float().cpu(), ref.to(dt).float().cpu()), 'dna second output %s (pitch 12)' % dt
            assert bool((out2[..., :5] == 7).all()) and bool((out2[..., 8:] == 7).all()), 'dna second output wrote outside its channels'
            dout = randn((b, h, w, c), 530 + k).to(dev)
            d2 = torch.zeros(b, h, w, 8, dtype=dt, device=dev)
            d2[..., 3:6] = randn((b, h, w, c), 540 + k).to(dev).to(dt)
            d2[..., :3] = 5.0                                        # must not be read
            got = abi.dna_bwd(lg, img, dout, k, bias=bias, dout2=d2, dout2_off=3)
            want = abi.dna_bwd(lg, img, dout + d2[..., 3:6].float(), k, bias=bias)
            abi.sync()
            close(got, want.double().cpu(), tol, 'dna second gradient %s k=%d' % (dt, k))


def case_conv_pitched(abi, tol, seed=0):
    """3- and 6-channel inputs stored with a channel pitch of 4 / 8 (in_pitch): same results as the dense tensor,
    pad channels of dx untouched."""
    for (b, h, w, cin, cout, pitch) in [(2, 16, 16, 3, 32, 4), (2, 12, 10, 6, 64, 8), (1, 9, 7, 5, 7, 8)]:
        x = uniform((b, h, w, cin), seed)
        wt = randn((5, 5, cin, cout), seed + 1, 0.1)
        xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        y_ref = T.conv2d(xd, wd, 2, 'SAME')
        dy = randn(tuple(y_ref.shape), seed + 2)
        dx_ref, dw_ref = torch.autograd.grad(y_ref, [xd, wd], dy.double())
        dev = abi.device
        xp = abi.concat_channels(x.to(dev), None, pitch=pitch)
        assert xp.shape[-1] == pitch and torch.equal(xp[..., :cin].cpu(), x) and torch.all(xp[..., cin:] == 0)
        tag = 'conv pitched %s' % ((b, h, w, cin, cout, pitch),)
        close(abi.conv2d_fwd(xp, wt.to(dev), 2, 'SAME'), y_ref, tol, tag + ' fwd')
        dxp = abi.conv2d_dgrad(dy.to(dev), wt.to(dev), tuple(xp.shape), 2, 'SAME')
        close(dxp[..., :cin], dx_ref, tol, tag + ' dgrad')
        assert torch.all(dxp[..., cin:] == 0), tag + ': pad channels of dx were written'
        close(abi.conv2d_wgrad(xp, dy.to(dev), tuple(wt.shape), 2, 'SAME'), dw_ref, tol, tag + ' wgrad')


def case_deconv_pitched(abi, tol, seed=0):
    """conv2d_transpose whose 266- / 10-channel input is stored at a pitch of 268 / 12 (out_pitch of the adjoint)."""
    for (b, ih, iw, cin, cout, pitch) in [(2, 4, 4, 266, 128, 268), (1, 3, 5, 10, 7, 12)]:
        x = uniform((b, ih, iw, cin), seed)
        wt = randn((5, 5, cout, cin), seed + 1, 0.1)
        xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        y_ref = T.conv2d_transpose(xd, wd, 2, 'SAME')
        dy = randn(tuple(y_ref.shape), seed + 2)
        dx_ref, dw_ref = torch.autograd.grad(y_ref, [xd, wd], dy.double())
        dev = abi.device
        xp = abi.concat_channels(x.to(dev), None, pitch=pitch)
        tag = 'deconv pitched %s' % ((b, ih, iw, cin, cout, pitch),)
        close(abi.deconv2d_fwd(xp, wt.to(dev), 2), y_ref, tol, tag + ' fwd')
        dxp = abi.deconv2d_dgrad(dy.to(dev), wt.to(dev), tuple(xp.shape), 2)
        close(dxp[..., :cin], dx_ref, tol, tag + ' dgrad')
        assert torch.all(dxp[..., cin:] == 0), tag + ': pad channels of dx were written'
        close(abi.deconv2d_wgrad(xp, dy.to(dev), tuple(wt.shape), 2), dw_ref, tol, tag + ' wgrad')


def case_deconv(abi, shape, tol, seed=0):
    b, ih, iw, cin, cout, k, s = shape
    x = uniform((b, ih, iw, cin), seed)
    wt = randn((k, k, cout, cin), seed + 1, 0.1)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y_ref = T.conv2d_transpose(xd, wd, s, 'SAME')
    dy = randn(tuple(y_ref.shape), seed + 2)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, [xd, wd], dy.double())
    dev = abi.device
    xg, wg, dyg = x.to(dev), wt.to(dev), dy.to(dev)
    tag = 'deconv%s' % (shape,)
    close(abi.deconv2d_fwd(xg, wg, s), y_ref, tol, tag + ' fwd')
    close(abi.deconv2d_dgrad(dyg, wg, tuple(x.shape), s), dx_ref, tol, tag + ' dgrad')
    close(abi.deconv2d_wgrad(xg, dyg, tuple(wt.shape), s), dw_ref, tol, tag + ' wgrad')


# (rows-shape, C, groups, act)
BN_SHAPES = [
    ((2, 32, 32), 32, 1, 'relu'), ((2, 4, 4), 256, 1, 'lrelu'), ((4, 2, 2), 1, 2, None),
    ((2, 16, 16), 138, 1, 'lrelu'), ((6, 8, 8), 16, 2, 'relu'), ((2, 3, 5), 7, 1, 'lrelu'),
    ((2, 64, 64), 64, 1, 'lrelu'), ((8, 1, 1), 5, 1, None),
]


def _bn_ref(x, beta, act, groups):
    outs = []
    for xg in x.chunk(groups, dim=0):
        u = T.batch_norm_train(xg, beta)
        outs.append({'relu': T.relu, 'lrelu': T.lrelu, None: lambda t: t}[act](u))
    return torch.cat(outs, dim=0)


def case_bn(abi, shape, tol, seed=0):
    lead, c, groups, act = shape
    x = randn(lead + (c,), seed, 1.5) + 0.7       # non-zero mean exercises the variance formula
    beta = randn((c,), seed + 1, 0.3)
    xd, bd = x.double().requires_grad_(True), beta.double().requires_grad_(True)
    y_ref = _bn_ref(xd, bd, act, groups)
    dy = randn(tuple(y_ref.shape), seed + 2)
    dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
    dev = abi.device
    xg, bg, dyg = x.to(dev), beta.to(dev), dy.to(dev)
    tag = 'bn%s' % (shape,)
    y, mean, rstd = abi.bn_act_fwd(xg, bg, act, groups)
    close(y, y_ref, tol, tag + ' fwd')
    mref = torch.stack([t.mean(dim=(0, 1, 2)) for t in x.double().chunk(groups, dim=0)]).reshape(-1)
    close(mean, mref, tol, tag + ' mean')
    dx, dbeta = abi.bn_act_bwd(xg, dyg, bg, mean, rstd, act, groups)
    close(dx, dx_ref, tol * 4, tag + ' dx')
    close(dbeta, db_ref, tol * 4, tag + ' dbeta')


def case_bn_large_mean(abi, tol):
    """Variance must survive a mean that dwarfs the spread (catastrophic cancellation check)."""
    x = randn((4, 8, 8, 6), 5, 0.01) + 100.0
    beta = torch.zeros(6)
    y_ref = _bn_ref(x.double(), beta.double(), None, 1)
    y, _, _ = abi.bn_act_fwd(x.to(abi.device), beta.to(abi.device), None, 1)
    close(y, y_ref, max(tol, 2e-2), 'bn large-mean fwd')


def case_bias(abi, tol, seed=0):
    for lead, c, act in [((2, 64, 64), 3, 'tanh'), ((2, 64, 64), 25, None), ((2, 1, 1), 5, None),
                         ((2, 5, 3), 7, 'relu'), ((2, 5, 3), 7, 'lrelu')]:
        x = randn(lead + (c,), seed)
        bias = randn((c,), seed + 1, 0.5)
        xd, bd = x.double().requires_grad_(True), bias.double().requires_grad_(True)
        y_ref = {'tanh': torch.tanh, 'relu': T.relu, 'lrelu': T.lrelu, None: lambda t: t}[act](xd + bd)
        dy = randn(tuple(y_ref.shape), seed + 2)
        dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
        dev = abi.device
        y = abi.bias_act_fwd(x.to(dev), bias.to(dev), act)
        close(y, y_ref, tol, 'bias %s fwd' % act)
        dx, db = abi.bias_act_bwd(y, dy.to(dev), act, want_dx=True)
        close(dx, dx_ref, tol * 4, 'bias %s dx' % act)
        close(db, db_ref, tol * 4, 'bias %s dbias' % act)
        if act is None:
            _, db2 = abi.bias_act_bwd(y, dy.to(dev), act, want_dx=False)
            close(db2, db_ref, tol * 4, 'bias none dbias (dx NULL)')


DNA_SHAPES = [(2, 64, 64, 3, 5), (1, 16, 16, 3, 6), (1, 24, 20, 3, 11), (2, 7, 5, 3, 5), (1, 3, 3, 1, 5), (1, 8, 8, 4, 3),
              # the 16-lane-row kernel (k >= 6): widths beyond / not a multiple of its 64-pixel block, C != 3
              (1, 9, 70, 3, 7), (2, 5, 130, 1, 6), (1, 6, 33, 4, 9), (1, 4, 4, 2, 8)]


def case_sync_bn_entries(abi, shape, act, groups, tol):
    """The four synchronised-BatchNorm entries on ONE rank (global = local) must reproduce acg_bn_act_fwd / _bwd, and
    with doubled total_rows and sums of two identical halves they must give the two-rank answer for identical shards."""
    g = torch.Generator().manual_seed(9)
    dev = abi.device
    x = (torch.randn(*shape, generator=g) * 2 + 0.5).to(dev)
    dy = torch.randn(*shape, generator=g).to(dev)
    beta = torch.randn(shape[-1], generator=g).to(dev)
    y_ref, mean_ref, rstd_ref = abi.bn_act_fwd(x, beta, act, groups=groups)
    dx_ref, dbeta_ref = abi.bn_act_bwd(x, dy, beta, mean_ref, rstd_ref, act, groups=groups)
    mom = abi.bn_moments(x, groups=groups)
    y, mean, rstd = abi.bn_act_fwd_moments(x, beta, mom, act, groups=groups)
    tag = 'sync_bn%s %s g%d' % (shape, act, groups)
    close(mean, mean_ref, tol, tag + ' mean'); close(rstd, rstd_ref, tol * 4, tag + ' rstd'); close(y, y_ref, tol * 4, tag + ' y')
    sums = abi.bn_bwd_sums(x, dy, beta, mean_ref, rstd_ref, act, groups=groups)
    rows_per_group = x.numel() // shape[-1] // groups
    dx, dbeta = abi.bn_act_bwd_sums(x, dy, beta, mean_ref, rstd_ref, sums, sums, rows_per_group, act, groups=groups)
    close(dx, dx_ref, tol * 8, tag + ' dx'); close(dbeta, dbeta_ref, tol * 8, tag + ' dbeta')
    # two ranks holding the SAME shard: global sums double, total rows double -> dx unchanged, dbeta still the local sum
    dx2, dbeta2 = abi.bn_act_bwd_sums(x, dy, beta, mean_ref, rstd_ref, sums * 2, sums, 2 * rows_per_group, act, groups=groups)
    close(dx2, dx_ref, tol * 8, tag + ' dx (two identical ranks)'); close(dbeta2, dbeta_ref, tol * 8, tag + ' dbeta (two identical ranks)')


def case_sync_bn_entries_bf16(abi, tol):
    """The same four entries in the storage types of a bf16 network (round 3: BASELINE config 3 can run the N-rank ==
    1-rank validation mode): bf16 x / y / dy / dx at the pitch round8(C), and the float32 head (x, dy float32 at a pitch
    of 8, dense float32 y, bf16 dx).  One rank (global = local): they must reproduce acg_bn_act_fwd / _bwd on the same
    tensors - bit for bit where the arithmetic is the same kernel, to rounding level otherwise."""
    dev = abi.device
    g = torch.Generator().manual_seed(19)
    for lead, c, groups, act in [((4, 16, 16), 32, 2, 'lrelu'), ((2, 8, 8), 138, 1, 'relu'), ((6, 5, 5), 12, 1, 'relu')]:
        cp = (c + 7) // 8 * 8
        x = torch.zeros(*lead, cp, dtype=torch.bfloat16, device=dev)
        x[..., :c] = (torch.randn(*lead, c, generator=g) * 2 + 0.5).to(dev).to(torch.bfloat16)
        dy = torch.zeros(*lead, cp, dtype=torch.bfloat16, device=dev)
        dy[..., :c] = torch.randn(*lead, c, generator=g).to(dev).to(torch.bfloat16)
        beta = torch.randn(c, generator=g).to(dev)
        y_ref, mean_ref, rstd_ref = abi.bn_act_fwd(x, beta, act, groups=groups, c=c)
        mom = abi.bn_moments(x, groups=groups, c=c)
        y, mean, rstd = abi.bn_act_fwd_moments(x, beta, mom, act, groups=groups, c=c)
        tag = 'sync_bn bf16 %s c=%d %s g%d' % (lead, c, act, groups)
        close(mean, mean_ref, tol, tag + ' mean'); close(rstd, rstd_ref, tol * 4, tag + ' rstd')
        close(y[..., :c].float(), y_ref[..., :c].float(), 8e-3, tag + ' y')      # one bf16 ulp where mean / rstd differ in the last bit
        assert (y[..., c:] == 0).all()
        rows_per_group = x.numel() // cp // groups
        # backward against the fp64 BatchNorm gradient of the same stored tensors
        xd, bd = x[..., :c].double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
        dx64, db64 = torch.autograd.grad(_bn_ref(xd, bd, act, groups), [xd, bd], dy[..., :c].double().cpu())
        sums = abi.bn_bwd_sums(x, dy, beta, mean_ref, rstd_ref, act, groups=groups, c=c)
        dx, dbeta = abi.bn_act_bwd_sums(x, dy, beta, mean_ref, rstd_ref, sums, sums, rows_per_group, act, groups=groups, c=c)
        assert dx.dtype == torch.bfloat16 and (dx[..., c:] == 0).all()
        close(dx[..., :c].float(), dx64, 8e-3, tag + ' dx'); close(dbeta, db64, 2e-4, tag + ' dbeta')
    # the float32 head (d/conv6): x float32 at a pitch of 8, y / dy dense float32, dx bf16
    for lead, c, groups in [((8, 2, 2), 1, 2), ((6, 3, 3), 5, 1)]:
        x32 = torch.zeros(*lead, 8, device=dev)
        x32[..., :c] = (torch.randn(*lead, c, generator=g) * 1.5 + 0.3).to(dev)
        beta = torch.randn(c, generator=g).to(dev)
        dy = torch.randn(*lead, c, generator=g).to(dev)
        y_ref, mean_ref, rstd_ref = abi.bn_act_fwd(x32, beta, None, groups, y_dtype=torch.float32, c=c)
        dx_ref, dbeta_ref = abi.bn_act_bwd(x32, dy, beta, mean_ref, rstd_ref, None, groups, dx_dtype=torch.bfloat16)
        mom = abi.bn_moments(x32, groups=groups, c=c)
        y, mean, rstd = abi.bn_act_fwd_moments(x32, beta, mom, None, groups=groups, c=c, y_dtype=torch.float32)
        y = y[..., :c] if y.shape[-1] != c else y
        close(mean, mean_ref, tol, 'sync_bn head mean'); close(rstd, rstd_ref, tol * 4, 'sync_bn head rstd'); close(y, y_ref, tol * 4, 'sync_bn head y')
        sums = abi.bn_bwd_sums(x32, dy, beta, mean_ref, rstd_ref, None, groups=groups, c=c)
        rows_per_group = x32.numel() // 8 // groups
        dx, dbeta = abi.bn_act_bwd_sums(x32, dy, beta, mean_ref, rstd_ref, sums, sums, rows_per_group, None, groups=groups, c=c, dx_dtype=torch.bfloat16)
        assert dx.dtype == torch.bfloat16
        close(dx[..., :c].float(), dx_ref[..., :c].float(), 8e-3, 'sync_bn head dx'); close(dbeta, dbeta_ref, tol * 8, 'sync_bn head dbeta')


PAIR_LAYERS = [   # (x shape, w shape, stride, padding, transposed)
    ((4, 16, 16, 8), (5, 5, 8, 24), 2, 'SAME', False),        # both contractions on the 128x32 tile, weight gradient split
    ((4, 16, 16, 64), (5, 5, 64, 128), 2, 'SAME', False),     # 64x64 tiles (g/conv3-like), both split
    ((2, 9, 7, 4), (5, 5, 4, 36), 2, 'SAME', False),          # ragged extents, mixed tiles
    ((4, 12, 12, 12), (3, 3, 12, 8), 1, 'VALID', False),
    ((4, 8, 8, 16), (5, 5, 8, 16), 2, None, True),            # transposed layer (adjoint FWD + WGRAD)
    ((2, 4, 4, 128), (5, 5, 64, 128), 2, None, True),         # g/tconv2-like
    ((2, 8, 8, 6), (5, 5, 6, 16), 2, 'SAME', False),          # Cin = 6: not float4-able -> two launches inside, same results
    ((2, 16, 16, 32), (5, 5, 25, 32), 2, None, True),         # g/tconv4-like: 25-channel gathers (ragged variant of the pair)
    ((2, 8, 8, 8), (3, 3, 8, 5), 1, 'SAME', False),           # Cout = 5: dense rows not float4-able -> two launches inside
]


def case_bwd_pair(abi, tol, exact):
    """acg_(de)conv2d_bwd_pair == the separate dgrad and wgrad entries (bit for bit on the HIP side), with an
    accumulating weight gradient and with the slabs-only variant feeding acg_splitk_reduce_many."""
    dev = abi.device
    for i, (xs, ws_, stride, padding, transposed) in enumerate(PAIR_LAYERS):
        x = randn(xs, 70 + i).to(dev)
        w = randn(ws_, 80 + i, 0.1).to(dev)
        if transposed:
            dy = randn((xs[0], xs[1] * stride, xs[2] * stride, ws_[2]), 90 + i).to(dev)
            dx_ref = abi.deconv2d_dgrad(dy, w, tuple(xs), stride)
            dw_plain = lambda dw, acc: abi.deconv2d_wgrad(x, dy, ws_, stride, dw=dw, accumulate=acc)    # noqa: E731
        else:
            d = abi.desc(xs[0], xs[1], xs[2], ws_[2], ws_[0], ws_[1], ws_[3], stride, padding)
            dy = randn((xs[0], d.out_h, d.out_w, ws_[3]), 90 + i).to(dev)
            dx_ref = abi.conv2d_dgrad(dy, w, tuple(xs), stride, padding)
            dw_plain = lambda dw, acc: abi.conv2d_wgrad(x, dy, ws_, stride, padding, dw=dw, accumulate=acc)    # noqa: E731
        init = randn(ws_, 100 + i).to(dev)
        for acc in (0.0, 1.0):
            dw_ref = dw_plain(init.clone(), acc)
            dx, dw = abi.bwd_pair(x, dy, w, stride, padding, transposed, accumulate=acc, dw=init.clone())
            abi.sync()
            tag = 'pair layer %d acc %g' % (i, acc)
            if exact:
                assert torch.equal(dx.cpu(), dx_ref.cpu()), tag + ': dx differs from the separate dgrad'
                assert torch.equal(dw.cpu(), dw_ref.cpu()), tag + ': dw differs from the separate wgrad'
            else:
                close(dx, dx_ref.double().cpu(), tol, tag + ' dx')
                close(dw, dw_ref.double().cpu(), tol, tag + ' dw')
        slabs, splits = abi.wgrad_slabs(x, dy, ws_, stride, padding, transposed)
        if slabs is not None:                       # the weight gradient is split: slabs-only pair + deferred reduction
            dx, (ws2, sp2) = abi.bwd_pair(x, dy, w, stride, padding, transposed, slabs_only=True)
            assert sp2 == splits
            out = init.clone()
            abi.splitk_reduce_many([(ws2, out, sp2, 1.0)])
            abi.sync()
            dw_ref = dw_plain(init.clone(), 1.0)
            if exact:
                assert torch.equal(dx.cpu(), dx_ref.cpu()) and torch.equal(out.cpu(), dw_ref.cpu()), 'pair layer %d (slabs only)' % i
            else:
                close(out, dw_ref.double().cpu(), tol, 'pair layer %d slabs' % i)


def case_wgrad_deferred(abi, tol, exact):
    """acg_(de)conv2d_wgrad_slabs + ONE acg_splitk_reduce_many over several layers == the per-layer acg_(de)conv2d_wgrad
    (bit for bit on the HIP side: same slabs, same summation order), including an accumulating entry and a ragged size."""
    dev = abi.device
    # (more than 16 output channels each: smaller layers take the direct kernels, which never split)
    layers = [((4, 16, 16, 8), (5, 5, 8, 32), 2, 'SAME', False), ((4, 18, 14, 3), (5, 5, 3, 40), 1, 'SAME', False),
              ((4, 12, 12, 12), (3, 3, 12, 24), 1, 'VALID', False), ((4, 8, 8, 32), (5, 5, 8, 32), 2, None, True)]
    entries, want, got = [], [], []
    for i, (xs, ws_, stride, padding, transposed) in enumerate(layers):
        x = randn(xs, 20 + i).to(dev)
        if transposed:
            dy = randn((xs[0], xs[1] * stride, xs[2] * stride, ws_[2]), 40 + i).to(dev)
            plain = lambda dw, acc: abi.deconv2d_wgrad(x, dy, ws_, stride, dw=dw, accumulate=acc)    # noqa: E731
        else:
            d = abi.desc(xs[0], xs[1], xs[2], ws_[2], ws_[0], ws_[1], ws_[3], stride, padding)
            dy = randn((xs[0], d.out_h, d.out_w, ws_[3]), 40 + i).to(dev)
            plain = lambda dw, acc: abi.conv2d_wgrad(x, dy, ws_, stride, padding, dw=dw, accumulate=acc)    # noqa: E731
        acc = 1.0 if i == 2 else 0.0                       # one entry accumulates into an existing gradient
        init = randn(ws_, 60 + i).to(dev)
        want.append(plain(init.clone(), acc))
        slabs, splits = abi.wgrad_slabs(x, dy, ws_, stride, padding, transposed)
        assert splits >= 1
        if slabs is None:                                  # planner does not split this shape: nothing to defer
            got.append(plain(init.clone(), acc))
            continue
        out = init.clone()
        got.append(out)
        entries.append((slabs, out, splits, acc))
    assert len(entries) >= 2, 'the case must exercise a multi-entry launch'
    step = torch.full((1,), 41, dtype=torch.int32, device=dev)     # acg_reduce_list::step_inc: the launch advances an optimizer's step counter
    abi.splitk_reduce_many(entries, step=step)
    abi.sync()
    assert int(step[0]) == 42, 'splitk_reduce_many did not increment the step counter exactly once'
    for i, (w_, g_) in enumerate(zip(want, got)):
        if exact:
            assert torch.equal(w_.cpu(), g_.cpu()), 'deferred reduction of layer %d is not bit-identical' % i
        else:
            close(g_, w_.double().cpu(), tol, 'deferred wgrad layer %d' % i)
    # two entries sharing an output would race: rejected
    with pytest.raises(Exception):
        abi.splitk_reduce_many([entries[0], (entries[1][0], entries[0][1], entries[1][2], 0.0)])


def case_copy_many(abi):
    """acg_copy_many: eight segments of different sizes in one launch - dense float4-able, dense ragged, pitched."""
    g = torch.Generator().manual_seed(3)
    dev = abi.device
    specs = [(64, 64, 64), (7, 10, 10), (1, 5, 5), (33, 3, 4), (129, 6, 8), (2, 138, 140), (1000, 1, 1), (16, 12, 12)]
    pairs, want = [], []
    for rows, cols, pitch in specs:
        src = torch.randn(rows, cols, generator=g).to(dev)
        dst = torch.full((rows, pitch), -7.0, device=dev)
        pairs.append((src, dst))
        ref = torch.full((rows, pitch), -7.0)
        ref[:, :cols] = src.cpu()
        want.append(ref)
    abi.copy_many(pairs)
    abi.sync()
    for (src, dst), ref in zip(pairs, want):
        assert torch.equal(dst.cpu(), ref), (tuple(src.shape), tuple(dst.shape))


def case_cdna(abi, shape, tol, seed=0):
    """shape = (B, H, W, C, masks, k).  Forward pieces and both gradients against autograd on the torch restatement;
    some raw parameters are negative (clamped by the relu: zero gradient there)."""
    b, h, w, c, m, k = shape
    g = torch.Generator().manual_seed(seed)
    params = (torch.randn(b, k * k * m, generator=g) * 0.7 + 0.3).float()
    img = (torch.rand(b, h, w, c, generator=g) * 2 - 1).float()
    dout = torch.randn(m, b, h, w, c, generator=g).float()
    pd, im = params.double().requires_grad_(True), img.double().requires_grad_(True)
    pieces = T.cdna_transform(pd, im, m, k)
    ref = torch.stack(pieces)
    (ref * dout.double()).sum().backward()
    dev = abi.device
    tag = 'cdna%s' % (shape,)
    out, kn = abi.cdna_fwd(params.to(dev), img.to(dev), m, k)
    close(out, ref.detach(), tol, tag + ' fwd')
    dpar, dimg = abi.cdna_bwd(params.to(dev), kn, img.to(dev), dout.to(dev), m, k)
    close(dimg, im.grad, tol * 4, tag + ' dimg')
    close(dpar, pd.grad, tol * 8, tag + ' dparams')
    assert (dpar.cpu()[params <= 1e-12] == 0).all(), tag + ' clamped parameters must have zero gradient'


def case_dna(abi, shape, tol, seed=0):
    b, h, w, c, k = shape
    logits = randn((b, h, w, k * k), seed, 2.0)
    img = uniform((b, h, w, c), seed + 1)
    ld = logits.double().requires_grad_(True)
    out_ref = T.dna_gather(ld, img.double(), k)
    dout = randn(tuple(out_ref.shape), seed + 2)
    dl_ref, = torch.autograd.grad(out_ref, [ld], dout.double())
    dev = abi.device
    tag = 'dna%s' % (shape,)
    close(abi.dna_fwd(logits.to(dev), img.to(dev), k), out_ref, tol, tag + ' fwd')
    close(abi.dna_bwd(logits.to(dev), img.to(dev), dout.to(dev), k), dl_ref, tol * 4, tag + ' bwd')


def case_dna_bias(abi, shape, tol, seed=0):
    """softmax(logits + bias) with the bias of the producing layer folded into the kernel (models.py:54-72), and
    dbias = sum over pixels of dlogits, accumulated into a running gradient."""
    b, h, w, c, k = shape
    logits = randn((b, h, w, k * k), seed, 2.0)
    bias = randn((k * k,), seed + 3, 1.0)
    img = uniform((b, h, w, c), seed + 1)
    ld, bd = logits.double().requires_grad_(True), bias.double().requires_grad_(True)
    out_ref = T.dna_gather(ld + bd, img.double(), k)
    dout = randn(tuple(out_ref.shape), seed + 2)
    dl_ref, db_ref = torch.autograd.grad(out_ref, [ld, bd], dout.double())
    dev = abi.device
    tag = 'dna+bias%s' % (shape,)
    close(abi.dna_fwd(logits.to(dev), img.to(dev), k, bias=bias.to(dev)), out_ref, tol, tag + ' fwd')
    dl, db = abi.dna_bwd(logits.to(dev), img.to(dev), dout.to(dev), k, bias=bias.to(dev), want_dbias=True)
    close(dl, dl_ref, tol * 4, tag + ' dlogits')
    close(db, db_ref, tol * 8, tag + ' dbias')


def r16(t):
    """Round to bfloat16 and back (what a bf16 tensor holds)."""
    return t.to(torch.bfloat16).float()


def case_dna_bf16(abi, shape, tol_f32, tol_bf16, seed=0):
    """bf16 logits at the pitch round8(k*k), float32 image / frame; dlogits come back as bf16 with zero pad taps."""
    b, h, w, c, k = shape
    kk, lp = k * k, (k * k + 7) // 8 * 8
    logits = r16(randn((b, h, w, kk), seed, 2.0))
    bias = randn((kk,), seed + 3, 1.0)
    img = uniform((b, h, w, c), seed + 1)
    ld, bd = logits.double().requires_grad_(True), bias.double().requires_grad_(True)
    out_ref = T.dna_gather(ld + bd, img.double(), k)
    dout = randn(tuple(out_ref.shape), seed + 2)
    dl_ref, db_ref = torch.autograd.grad(out_ref, [ld, bd], dout.double())
    dev = abi.device
    l16 = torch.zeros(b, h, w, lp, dtype=torch.bfloat16, device=dev)
    l16[..., :kk] = logits.to(dev).to(torch.bfloat16)
    tag = 'dna bf16%s' % (shape,)
    close(abi.dna_fwd(l16, img.to(dev), k, bias=bias.to(dev)), out_ref, tol_f32, tag + ' fwd')
    dl, db = abi.dna_bwd(l16, img.to(dev), dout.to(dev), k, bias=bias.to(dev), want_dbias=True)
    assert dl.dtype == torch.bfloat16 and (dl[..., kk:] == 0).all(), tag + ': pad taps of dlogits must stay zero'
    close(dl[..., :kk].float(), dl_ref, tol_bf16, tag + ' dlogits')
    close(db, db_ref, tol_bf16, tag + ' dbias')      # sums of float32 values formed before the bf16 rounding of dlogits


def case_bn_bf16(abi, shape, tol, seed=0):
    """bf16 x / y / dy / dx (BASELINE configs 3 and 5): statistics, beta and dbeta stay float32.  Reference: fp64
    BatchNorm of the bf16-rounded inputs; outputs carry one bf16 rounding."""
    lead, c, groups, act = shape
    x = r16(randn(lead + (c,), seed, 1.5) + 0.7)
    beta = randn((c,), seed + 1, 0.3)
    xd, bd = x.double().requires_grad_(True), beta.double().requires_grad_(True)
    y_ref = _bn_ref(xd, bd, act, groups)
    dy = r16(randn(tuple(y_ref.shape), seed + 2))
    dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
    dev = abi.device
    xg, bg, dyg = x.to(dev).to(torch.bfloat16), beta.to(dev), dy.to(dev).to(torch.bfloat16)
    tag = 'bn bf16%s' % (shape,)
    y, mean, rstd = abi.bn_act_fwd(xg, bg, act, groups)
    assert y.dtype == torch.bfloat16
    close(y.float(), y_ref, tol, tag + ' fwd')
    dx, dbeta = abi.bn_act_bwd(xg, dyg, bg, mean, rstd, act, groups)
    close(dx.float(), dx_ref, tol, tag + ' dx')
    close(dbeta, db_ref, 2e-4, tag + ' dbeta')


def case_bn_head_bf16(abi, tol, seed=0):
    """The loss-facing layer of a bf16 network (d/conv6: one channel at a pitch of 8): bf16 x / dx with pad channels,
    dense float32 y / dy."""
    for lead, c, groups in [((4, 2, 2), 1, 2), ((6, 3, 3), 5, 1)]:
        x = r16(randn(lead + (c,), seed, 1.5) + 0.3)
        beta = randn((c,), seed + 1, 0.3)
        xd, bd = x.double().requires_grad_(True), beta.double().requires_grad_(True)
        y_ref = _bn_ref(xd, bd, None, groups)
        dy = randn(tuple(y_ref.shape), seed + 2)
        dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
        dev = abi.device
        xp = torch.zeros(*lead, 8, dtype=torch.bfloat16, device=dev)
        xp[..., :c] = x.to(dev).to(torch.bfloat16)
        y, mean, rstd = abi.bn_act_fwd(xp, beta.to(dev), None, groups, y_dtype=torch.float32, c=c)
        assert y.dtype == torch.float32 and y.shape[-1] == c
        close(y, y_ref, 2e-5, 'bn head fwd')
        dx, dbeta = abi.bn_act_bwd(xp, dy.to(dev), beta.to(dev), mean, rstd, None, groups)
        assert dx.dtype == torch.bfloat16 and (dx[..., c:] == 0).all()
        close(dx[..., :c].float(), dx_ref, tol, 'bn head dx')
        close(dbeta, db_ref, 2e-4, 'bn head dbeta')


def case_head_f32_in_bf16_network(abi, tol_w, tol_dx, seed=0):
    """The head layer of a bf16 network kept in float32 (d/conv6, models.py:87-88): acg_conv2d_fwd with
    ACG_DTYPE2(ACG_BF16, ACG_F32) - bf16 operands, float32 result at the pitch round8, split over K or not - and
    acg_bn_act_bwd with ACG_DTYPE2(ACG_F32, ACG_BF16): float32 x / dy at a pitch of 8, bf16 dx."""
    dev = abi.device
    for i, (b, h, w, cin, cout, k, s_, pad) in enumerate([(8, 2, 2, 512, 1, 2, 1, 'SAME'), (2, 16, 16, 16, 5, 3, 1, 'SAME'), (4, 32, 32, 8, 40, 1, 1, 'SAME')]):
        x, wt = uniform((b, h, w, cin), seed + i), randn((k, k, cin, cout), seed + 10 + i, 0.1)
        y = abi.conv2d_fwd(x.to(dev), wt.to(dev), s_, pad, out_f32=True)
        abi.sync()
        y_ref = T.conv2d(r16(x).double(), r16(wt).double(), s_, pad)
        close(y, y_ref, tol_w, 'head conv (bf16 operands, f32 result) %d' % i)       # no rounding of the result: accumulation level
    for lead, c, groups in [((8, 2, 2), 1, 2), ((6, 3, 3), 5, 1)]:
        x = randn(lead + (c,), seed, 1.5) + 0.3
        beta = randn((c,), seed + 1, 0.3)
        xd, bd = x.double().requires_grad_(True), beta.double().requires_grad_(True)
        y_ref = _bn_ref(xd, bd, None, groups)
        dy = randn(tuple(y_ref.shape), seed + 2)
        dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
        xp = torch.zeros(*lead, 8, dtype=torch.float32, device=dev)
        xp[..., :c] = x.to(dev)
        y, mean, rstd = abi.bn_act_fwd(xp, beta.to(dev), None, groups, y_dtype=torch.float32, c=c)
        close(y, y_ref, 2e-5, 'f32 head bn fwd')
        dx, dbeta = abi.bn_act_bwd(xp, dy.to(dev), beta.to(dev), mean, rstd, None, groups, dx_dtype=torch.bfloat16)
        assert dx.dtype == torch.bfloat16 and dx.shape[-1] == 8 and (dx[..., c:] == 0).all()
        close(dx[..., :c].float(), dx_ref, tol_dx, 'f32 head bn dx (bf16)')
        close(dbeta, db_ref, 2e-4, 'f32 head bn dbeta')


def case_deconv_bias_act(abi, tol):
    """acg_deconv2d_fwd_bias_act: tanh(conv2d_transpose(x, w) + b) in one launch (models.py:20-21, the plain generator's frame),
    against the float64 composition; shapes the planner runs unsplit on its 128x32 tile, float32 and bf16 operands."""
    dev = abi.device
    r = (lambda t: t.bfloat16().float()) if abi.half else (lambda t: t)
    fused = 0
    for i, (b, ih, iw, cin, cout, k, s_, act) in enumerate([(16, 32, 32, 64, 3, 5, 2, 'tanh'), (8, 32, 32, 32, 25, 5, 2, None), (32, 16, 16, 16, 8, 3, 2, 'relu')]):
        x, wt = uniform((b, ih, iw, cin), 900 + i), randn((k, k, cout, cin), 910 + i, 0.1)
        bias = randn((cout,), 920 + i, 0.5)
        y = abi.deconv2d_fwd_bias_act(x.to(dev), wt.to(dev), bias.to(dev), s_, act)
        if y is None:
            continue
        fused += 1
        abi.sync()
        pre = T.conv2d_transpose(r(x).double(), r(wt).double(), s_, 'SAME') + bias.double()
        want = {'tanh': torch.tanh, 'relu': torch.relu, None: lambda t: t}[act](pre)
        assert y.dtype == torch.float32 and y.shape == want.shape
        close(y, want, tol, 'deconv + bias + %s in the epilogue, layer %d' % (act, i))
    assert fused >= 2, 'only %d layers took the fused epilogue' % fused


def case_bias_bf16(abi, tol, seed=0):
    """bias (+ tanh) heads of a bf16 network: bf16 conv output at pitch 8 in, dense float32 frame / state out; backward
    returns the bf16 gradient at pitch 8 and the float32 bias gradient."""
    for lead, c, act in [((2, 16, 16), 3, 'tanh'), ((4, 1, 1), 5, None)]:
        x = r16(randn(lead + (c,), seed))
        bias = randn((c,), seed + 1, 0.5)
        xd, bd = x.double().requires_grad_(True), bias.double().requires_grad_(True)
        y_ref = {'tanh': torch.tanh, None: lambda t: t}[act](xd + bd)
        dy = randn(tuple(y_ref.shape), seed + 2)
        dx_ref, db_ref = torch.autograd.grad(y_ref, [xd, bd], dy.double())
        dev = abi.device
        xp = torch.zeros(*lead, 8, dtype=torch.bfloat16, device=dev)
        xp[..., :c] = x.to(dev).to(torch.bfloat16)
        y = abi.bias_act_fwd(xp, bias.to(dev), act, c=c, y_dtype=torch.float32)
        close(y, y_ref, 2e-5, 'bias bf16 %s fwd' % act)
        dx, db = abi.bias_act_bwd(y, dy.to(dev), act, x_pitch=8, x_dtype=torch.bfloat16)
        assert dx.dtype == torch.bfloat16 and (dx[..., c:] == 0).all()
        close(dx[..., :c].float(), dx_ref, tol, 'bias bf16 %s dx' % act)
        close(db, db_ref, 1e-4, 'bias bf16 %s dbias' % act)


def case_plumbing_bf16(abi):
    """The channel plumbing of a bf16 network: exact copies / roundings, no arithmetic."""
    dev = abi.device
    x = r16(randn((2, 4, 4, 256), 0))
    a = randn((2, 10), 1)
    ref = torch.cat([x, r16(a).reshape(2, 1, 1, 10).expand(2, 4, 4, 10)], dim=3)
    y = abi.concat_actions(x.to(dev).to(torch.bfloat16), a.to(dev), pitch=272)
    assert y.dtype == torch.bfloat16 and (y[..., 266:] == 0).all()
    assert torch.equal(y[..., :266].float().cpu(), ref), 'concat_actions bf16'
    p, q = randn((2, 8, 8, 3), 2), randn((2, 8, 8, 3), 3)
    cat = abi.concat_channels(p.to(dev), q.to(dev), pitch=8, y_dtype=torch.bfloat16)      # float32 frames -> bf16 D input
    assert cat.dtype == torch.bfloat16 and torch.equal(cat[..., :6].float().cpu(), r16(torch.cat([p, q], dim=3))) and (cat[..., 6:] == 0).all()
    back = abi.slice_channels(cat, 3, 3, dst_dtype=torch.float32)                         # bf16 gradient -> float32 frame gradient
    assert back.dtype == torch.float32 and torch.equal(back.cpu(), r16(q)), 'slice bf16 -> f32'
    same = abi.slice_channels(y, 0, 256)
    assert same.dtype == torch.bfloat16 and torch.equal(same.float().cpu(), x), 'slice bf16 -> bf16'
    u, v = r16(randn((3, 5, 7, 8), 5)), r16(randn((3, 5, 7, 8), 6))
    s2 = abi.add(u.to(dev).to(torch.bfloat16), v.to(dev).to(torch.bfloat16))
    assert torch.equal(s2.float().cpu(), r16(u + v)), 'add bf16'
    src = randn((6, 3), 7).to(dev)
    dst = torch.zeros(6, 8, dtype=torch.bfloat16, device=dev)
    abi.copy_many([(src, dst)])
    assert torch.equal(dst[:, :3].float().cpu(), r16(src.cpu())) and (dst[:, 3:] == 0).all(), 'copy_many f32 -> bf16'


def case_weights_prepare(abi):
    """acg_weights_prepare_bf16: both operand layouts, zero padded to multiples of 8."""
    for shape in [(5, 5, 3, 32), (5, 5, 266, 128), (2, 2, 512, 1), (4, 4, 16, 5), (5, 5, 25, 128)]:
        w = randn(shape, 11, 0.1)
        rm, tr = abi.prep_weights(w)
        abi.sync()
        kh, kw, a, b = shape
        wr = r16(w).reshape(kh * kw, a, b)
        assert torch.equal(rm[:, :, :b].float().cpu(), wr) and (rm[:, :, b:] == 0).all(), 'rm %s' % (shape,)
        assert torch.equal(tr[:, :, :a].float().cpu(), wr.permute(0, 2, 1)) and (tr[:, :, a:] == 0).all(), 'tr %s' % (shape,)


def case_opt_step_prepared(abi):
    """acg_opt_step_prepare_bf16 (optimizer update + refresh of the bf16 filter copies in one launch) against the two launches
    it replaces, bit for bit: parameters, slots and both copies of every filter, over two steps; filters of awkward shapes
    (3 and 138 gathered channels, one and five output channels: the scalar path) with beta / bias vectors between them."""
    import ctypes
    from action_conditioned_gans_amd import _lib as L
    dev = abi.device
    shapes = [(5, 5, 6, 64), (5, 5, 138, 128), (2, 2, 512, 1), (4, 4, 16, 5), (5, 5, 25, 128), (3, 3, 32, 16)]
    offs, total = [], 0
    for sh in shapes:
        offs.append(total)
        n = sh[0] * sh[1] * sh[2] * sh[3]
        total += -(-n // 4) * 4 + 4 * (1 + sh[3] // 4)          # the filter (16-byte aligned) and a vector behind it
    total += 25
    for kind in ('adam', 'rmsprop'):
        p0 = randn((total,), 40, 0.02)
        bufs = {}
        for path in ('two', 'one'):
            p = p0.clone().to(dev)
            s1 = (torch.zeros(total) if kind == 'adam' else torch.ones(total)).to(dev)
            s2 = torch.zeros(total, device=dev)
            step = torch.zeros(1, dtype=torch.int32, device=dev)
            copies, pl = [], L.PrepList()
            for i, (sh, off) in enumerate(zip(shapes, offs)):
                kh, kw, a, b = sh
                rm = torch.full((kh * kw, a, (b + 7) // 8 * 8), float('nan'), dtype=torch.bfloat16, device=dev)
                tr = torch.full((kh * kw, b, (a + 7) // 8 * 8), float('nan'), dtype=torch.bfloat16, device=dev)
                copies.append((rm, tr))
                pl.src[i], pl.rm[i], pl.tr[i] = p.data_ptr() + 4 * off, rm.data_ptr(), tr.data_ptr()
                pl.taps[i], pl.a[i], pl.b[i] = kh * kw, a, b
            for t in range(2):
                g = randn((total,), 50 + t, 0.1).to(dev)
                clip = (-0.03, 0.03) if t == 1 else None
                lo, hi = clip if clip else (0.0, 0.0)
                if path == 'two':
                    if kind == 'adam':
                        abi.adam_step(p, g, s1, s2, step, gs=0.5, clip=clip)
                    else:
                        abi.rmsprop_step(p, g, s1, gs=0.5, clip=clip)
                    abi.lib.weights_prepare_bf16(ctypes.byref(pl), len(shapes), abi.stream())
                else:
                    if kind == 'adam':
                        abi.lib.step_inc(_ptr(step), abi.stream())
                        oa = L.OptArgs(0, 1e-3, 0.9, 0.999, 1e-8, 0.5, 1 if clip else 0, lo, hi)
                    else:
                        oa = L.OptArgs(1, 5e-5, 0.9, 0.0, 1e-10, 0.5, 1 if clip else 0, lo, hi)
                    abi.lib.opt_step_prepare_bf16(_ptr(p), _ptr(g), _ptr(s1), _ptr(s2), _ptr(step), total, ctypes.byref(oa), ctypes.byref(pl),
                                                  len(shapes), abi.stream())
                abi.sync()
            bufs[path] = (p, s1, s2, copies)
        (pa, s1a, s2a, ca), (pb, s1b, s2b, cb) = bufs['two'], bufs['one']
        assert torch.equal(pa, pb), '%s: parameters differ: %d elements, max %.3g' % (kind, int((pa != pb).sum()), float((pa - pb).abs().max()))
        assert torch.equal(s1a, s1b) and torch.equal(s2a, s2b), kind + ': slots differ'
        assert float((pa.cpu() - p0).abs().max()) > 0, 'test premise: nothing was updated'
        for i, ((rma, tra), (rmb, trb)) in enumerate(zip(ca, cb)):
            assert torch.equal(rma.view(torch.int16), rmb.view(torch.int16)), '%s: rm copy of filter %d differs' % (kind, i)
            assert torch.equal(tra.view(torch.int16), trb.view(torch.int16)), '%s: tr copy of filter %d differs' % (kind, i)


def _ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def case_dna_extreme_logits(abi, tol):
    """softmax must be max-subtracted: logits of +-80 overflow a naive exp in fp32."""
    logits = randn((1, 6, 6, 25), 3, 1.0)
    logits[0, 2, 3, 7] = 90.0
    logits[0, 4, 1, :] = -90.0
    img = uniform((1, 6, 6, 3), 4)
    ref = T.dna_gather(logits.double(), img.double(), 5)
    close(abi.dna_fwd(logits.to(abi.device), img.to(abi.device), 5), ref, tol, 'dna extreme logits')


def case_plumbing(abi, tol):
    dev = abi.device
    x = randn((2, 4, 4, 256), 0)
    a = randn((2, 10), 1)
    ref = torch.cat([x, a.reshape(2, 1, 1, 10).expand(2, 4, 4, 10)], dim=3)
    close(abi.concat_actions(x.to(dev), a.to(dev)), ref, 0, 'concat_actions')
    p, q = randn((2, 8, 8, 3), 2), randn((2, 8, 8, 3), 3)
    cat = abi.concat_channels(p.to(dev), q.to(dev))
    close(cat, torch.cat([p, q], dim=3), 0, 'concat_channels')
    close(abi.slice_channels(cat, 3, 3), q, 0, 'slice_channels')
    dst = randn((2, 8, 8, 3), 4).to(dev)
    close(abi.slice_channels(cat, 0, 3, dst=dst.clone(), accumulate=1.0), dst.cpu().double() + p.double(), 1e-6,
          'slice_channels accumulate')
    close(abi.add(p.to(dev), q.to(dev)), p.double() + q.double(), 1e-6, 'add')


def case_losses(abi, tol, seed=0):
    dev = abi.device
    for shp in [(2, 64, 64, 3), (1, 5, 7, 3), (3, 2, 2, 1), (2, 16, 32, 4), (1, 12, 16, 2)]:
        gen, gt = uniform(shp, seed), uniform(shp, seed + 1)
        gd = gen.double().requires_grad_(True)
        l1 = (gd - gt.double()).abs().sum()
        g = T.gdl(gd, gt.double())
        w1, w2 = 0.05 / shp[0], 1.0
        dref, = torch.autograd.grad(w1 * l1 + w2 * g, [gd])
        out, dgen = abi.frame_loss(gen.to(dev), gt.to(dev), w1, w2)
        close(out, torch.stack([l1, g]).detach(), tol, 'frame_loss values %s' % (shp,))
        # sign-type gradients are exact except where |a|-|b| ties to rounding: compare by mismatch count
        bad = ((dgen.double().cpu() - dref).abs() > 1e-6).sum().item()
        assert bad <= max(2, dgen.numel() // 5000), 'frame_loss grad: %d mismatching elements %s' % (bad, shp)
        # symmetric in its arguments (reference passes them swapped, train.py:81 vs ops.py:100; defect D10)
        out_sw, _ = abi.frame_loss(gt.to(dev), gen.to(dev), w1, w2, want_grad=False)
        close(out_sw, out, 1e-6, 'frame_loss symmetry')
        # the gradient alone (what the training step launches: no values, no finalize): the very same gradient
        _, dgen2 = abi.frame_loss(gen.to(dev), gt.to(dev), w1, w2, want_values=False)
        assert torch.equal(dgen2.cpu(), dgen.cpu()), 'frame_loss gradient-only mode %s' % (shp,)
    pred, tgt = randn((32, 5), seed + 2), randn((32, 5), seed + 3)
    pd = pred.double().requires_grad_(True)
    n2 = torch.sqrt(((pd - tgt.double()) ** 2).sum())
    dref, = torch.autograd.grad(n2 / 32, [pd])
    out, d = abi.l2norm_loss(pred.to(dev), tgt.to(dev), 1.0 / 32)
    close(out, n2.detach().reshape(1), tol, 'l2norm value')
    close(d, dref, tol * 4, 'l2norm grad')
    z = torch.zeros(4, 5)
    out, d = abi.l2norm_loss(z.to(dev), z.to(dev), 1.0)
    assert out.item() == 0 and torch.all(d == 0), 'l2norm at zero must give zero gradient, not NaN'
    for label in (0.0, 0.9, 1.0):
        x = randn((32, 2, 2, 1), seed + 4, 3.0)
        x[0, 0, 0, 0], x[1, 0, 0, 0] = 60.0, -60.0
        xd = x.double().requires_grad_(True)
        ce = T.sigmoid_cross_entropy(torch.full_like(xd, label), xd)
        dref, = torch.autograd.grad(ce, [xd])
        out, d = abi.sigmoid_ce_loss(x.to(dev), label, 1.0)
        close(out, ce.detach().reshape(1), tol, 'sigmoid_ce value label=%g' % label)
        close(d, dref, tol * 4, 'sigmoid_ce grad label=%g' % label)
    x = randn((32, 2, 2, 1), seed + 5)
    out, d = abi.mean_loss(x.to(dev), -1.0)
    close(out, x.double().mean().reshape(1), tol, 'mean value')
    close(d, torch.full_like(x.double(), -1.0 / x.numel()), 1e-6, 'mean grad')
    a, b = uniform((2, 64, 64, 3), seed + 6), uniform((2, 64, 64, 3), seed + 7)
    close(abi.psnr(a.to(dev), b.to(dev)), T.psnr(a.double(), b.double()).reshape(1), tol, 'psnr')
    s0, s1 = torch.tensor([3.0]), torch.tensor([-2.0])
    close(abi.scalar_combine([(s0.to(dev), 0.5), (s1.to(dev), 2.0)]), torch.tensor([-2.5]), 1e-6, 'scalar_combine')


def case_optimizers(abi, tol, seed=0):
    dev = abi.device
    n = 10007
    p0 = randn((n,), seed, 0.02)
    grads = [randn((n,), seed + 1 + i, 0.1) for i in range(3)]
    # Adam, TF formulas, 3 steps with and without clip + grad_scale
    f32 = lambda v: float(np.float32(v))      # TF holds its hyper-parameters as float32 constants
    lr, b1, b2, eps = f32(1e-3), f32(0.9), f32(0.999), f32(1e-8)
    for clip, gs in ((None, 1.0), ((-0.01, 0.01), 0.5)):
        p, m, v = p0.double().clone(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        pg, mg, vg = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        for t, g in enumerate(grads, 1):
            gd = g.double() * gs
            lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            m = b1 * m + (1 - b1) * gd
            v = b2 * v + (1 - b2) * gd * gd
            p = p - lr_t * m / (v.sqrt() + eps)
            if clip:
                p = p.clamp(f32(clip[0]), f32(clip[1]))
            abi.adam_step(pg, g.to(dev), mg, vg, step, gs=gs, clip=clip)
        assert step.item() == 3
        close(pg, p, tol, 'adam param clip=%s' % (clip,))
        close(mg, m, tol, 'adam m')
        close(vg, v, tol, 'adam v')
    # RMSProp: ms starts at ONE
    p, ms = p0.double().clone(), torch.ones(n, dtype=torch.float64)
    pg, msg = p0.clone().to(dev), torch.ones(n, device=dev)
    for g in grads:
        gd = g.double()
        ms = f32(0.9) * ms + (1 - f32(0.9)) * gd * gd
        p = (p - f32(5e-5) * gd / torch.sqrt(ms + f32(1e-10))).clamp(f32(-0.01), f32(0.01))
        abi.rmsprop_step(pg, g.to(dev), msg, clip=(-0.01, 0.01))
    close(pg, p, tol, 'rmsprop param')
    close(msg, ms, tol, 'rmsprop ms')
    q = randn((n,), seed + 9, 0.05).to(dev)
    ref = q.double().cpu().clamp(f32(-0.01), f32(0.01))
    abi.clip(q, -0.01, 0.01)
    close(q, ref, 1e-7, 'clip')


def case_repeatable_launches(abi, reps=60):
    """Race screen for the kernels that synchronise by hand (cdna_hip_programming.md, section 5: "place reads by the vmcnt / barrier
    count, never by clean runs" - and then screen): the LDS-DMA convolution (counted vmcnt, raw s_barrier, two wave groups one
    phase apart) and the one-launch BatchNorm kernels (in-launch exchange of partial sums, epoch reused across launches).  Each is
    deterministic by construction, so `reps` launches on the same operands must give bit-identical results; a DMA landing after
    the fragment read, or a stale exchange granule, shows up as a run that differs."""
    import ctypes
    from action_conditioned_gans_amd import _lib as L
    from abi_call import _p
    dev = abi.device
    # ---- LDS-DMA convolution: forward (ragged rows and columns) and the stride classes of an input gradient
    for (b, h, w_, cin, cout, k, s, which) in [(52, 62, 62, 56, 96, 5, 2, 0), (13, 63, 61, 96, 160, 5, 2, 1)]:
        assert tile_rows(abi, which, b, h, w_, cin, k, cout, s, 'SAME') == 256, 'test premise: the planner does not pick the wide kernel'
        d = abi.desc(b, h, w_, cin, k, k, cout, s, 'SAME')
        x16 = abi.to16(uniform((b, h, w_, cin), 900))
        rm, tr = abi.prep_weights(randn((k, k, cin, cout), 901, 0.05))
        dy16 = abi.to16(randn((b, d.out_h, d.out_w, cout), 902))
        code = L.CONV_FWD if which == 0 else L.CONV_DGRAD
        ws, n = abi.ws(abi.lib.conv2d_workspace_bytes(ctypes.byref(d), code, L.ACG_BF16))
        outs = []
        for r in range(reps):
            if which == 0:
                y = torch.zeros(b, d.out_h, d.out_w, (cout + 7) // 8 * 8, dtype=torch.bfloat16, device=dev)
                abi.lib.conv2d_fwd(_p(x16), _p(tr), _p(y), ctypes.byref(d), L.ACG_BF16, _p(ws), n, abi.stream())
            else:
                y = torch.zeros(b, h, w_, (cin + 7) // 8 * 8, dtype=torch.bfloat16, device=dev)
                abi.lib.conv2d_dgrad(_p(dy16), _p(rm), _p(y), ctypes.byref(d), L.ACG_BF16, _p(ws), n, abi.stream())
            outs.append(y)
        for r in range(1, reps):
            assert torch.equal(outs[r], outs[0]), 'wide conv (which=%d): launch %d differs from launch 0' % (which, r)
    # ---- one-launch BatchNorm, forward and backward, ONE workspace per call site reused by every launch (as a captured graph does)
    for (rows, c, groups, tdt) in [(32768, 128, 1, torch.float32), (65536, 64, 2, torch.bfloat16), (8192, 128, 2, torch.float32)]:
        x = (randn((rows, c), 910, 1.5) + 0.3).to(dev).to(tdt)
        dy = randn((rows, c), 911).to(dev).to(tdt)
        beta = randn((c,), 912, 0.1).to(dev)
        code = L.ACG_BF16 if tdt == torch.bfloat16 else L.ACG_F32
        wsf, nf = abi.bn_ws(rows, c, groups)
        wsb, nb = abi.bn_ws(rows, c, groups)
        mean, rstd, dbeta = abi.empty(groups * c), abi.empty(groups * c), abi.empty(c)
        first = None
        for r in range(reps):
            y, dx = torch.zeros_like(x), torch.zeros_like(x)
            abi.lib.bn_act_fwd(_p(x), _p(beta), _p(y), _p(mean), _p(rstd), rows, c, 0, 0, groups, 1e-3, L.ACT_LRELU, 0.2, code, 0, _p(wsf), nf, abi.stream())
            abi.lib.bn_act_bwd(_p(x), _p(dy), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dbeta), 0.0, rows, c, 0, 0, groups, L.ACT_LRELU, 0.2, code, 0, _p(wsb), nb, abi.stream())
            got = (y, dx, mean.clone(), rstd.clone(), dbeta.clone())
            if first is None:
                first = got
            else:
                for a_, b_, name in zip(got, first, ('y', 'dx', 'mean', 'rstd', 'dbeta')):
                    assert torch.equal(a_, b_), 'BatchNorm %d x %d x %d groups: %s of launch %d differs from launch 0' % (rows // groups, c, groups, name, r)
        abi.no_timeout(wsf); abi.no_timeout(wsb)
        # the epoch word counts the launches: proof that the one-launch kernels (not the two-launch path) ran
        assert int(wsb[0:4].view(torch.int32)[0]) == reps and int(wsf[0:4].view(torch.int32)[0]) == reps, 'the one-launch kernels did not run'
