"""CPU suite: the two independent restatements (torch-fp64 composition vs brute-force C loops)
must agree on every op of the hot path.  This is what stands in for reference golden vectors
(parity unpinned: the reference has none)."""
import pytest

import op_cases as C

TOL = 2e-6   # the C oracle stores float32; the torch side is float64


@pytest.mark.parametrize('shape', C.CONV_SHAPES_SMALL, ids=str)
def test_conv(oracle_abi, shape):
    C.case_conv(oracle_abi, shape, TOL)


def test_deconv_pitched(oracle_abi):
    C.case_deconv_pitched(oracle_abi, TOL)


def test_conv_pitched(oracle_abi):
    C.case_conv_pitched(oracle_abi, TOL)


@pytest.mark.parametrize('shape', C.DECONV_SHAPES_SMALL, ids=str)
def test_deconv(oracle_abi, shape):
    C.case_deconv(oracle_abi, shape, TOL)


@pytest.mark.parametrize('shape', C.BN_SHAPES, ids=str)
def test_bn(oracle_abi, shape):
    C.case_bn(oracle_abi, shape, TOL)


def test_bn_large_mean(oracle_abi):
    C.case_bn_large_mean(oracle_abi, 1e-4)


def test_bias(oracle_abi):
    C.case_bias(oracle_abi, TOL)


@pytest.mark.parametrize('shape', C.DNA_SHAPES, ids=str)
def test_dna(oracle_abi, shape):
    C.case_dna(oracle_abi, shape, TOL)


@pytest.mark.parametrize('shape', [(2, 9, 7, 3, 4, 5), (1, 16, 16, 1, 1, 3), (2, 20, 17, 3, 10, 5), (1, 8, 8, 4, 3, 7)])
def test_cdna(oracle_abi, shape):
    C.case_cdna(oracle_abi, shape, TOL)


@pytest.mark.parametrize('shape,act,groups', [((4, 6, 5, 8), 'relu', 1), ((6, 3, 3, 12), 'lrelu', 2), ((2, 5, 7, 3), None, 1)])
def test_sync_bn_entries(oracle_abi, shape, act, groups):
    C.case_sync_bn_entries(oracle_abi, shape, act, groups, TOL)


def test_conv_bn_stats(oracle_abi):
    C.case_conv_bn_stats(oracle_abi, TOL, TOL, min_fused=6)


def test_conv_bn_stats_large_mean(oracle_abi):
    C.case_conv_bn_stats_large_mean(oracle_abi, 1e-4)


def test_dna_second(oracle_abi):
    C.case_dna_second(oracle_abi, TOL)


def test_bwd_pair(oracle_abi):
    C.case_bwd_pair(oracle_abi, TOL, exact=False)


def test_wgrad_deferred_reduction(oracle_abi):
    C.case_wgrad_deferred(oracle_abi, TOL, exact=False)


def test_copy_many(oracle_abi):
    C.case_copy_many(oracle_abi)


@pytest.mark.parametrize('shape', [C.DNA_SHAPES[i] for i in (0, 1, 3, 4, 8)], ids=str)
def test_dna_bias(oracle_abi, shape):
    """The C restatement of softmax(logits + bias) and its dbias against the torch restatement (autograd)."""
    C.case_dna_bias(oracle_abi, shape, TOL)


def test_dna_extreme(oracle_abi):
    C.case_dna_extreme_logits(oracle_abi, TOL)


def test_plumbing(oracle_abi):
    C.case_plumbing(oracle_abi, TOL)


def test_losses(oracle_abi):
    C.case_losses(oracle_abi, TOL)


def test_optimizers(oracle_abi):
    C.case_optimizers(oracle_abi, TOL)


# ---- pinning the restatement itself: hand-computed values and the published TF-1.0 index algebra --------
def test_same_padding_known_answers():
    """TF 'SAME': out = ceil(in / s), pad_total = max((out-1) s + k - in, 0), pad_before = pad_total // 2 (A.1)."""
    from oracle import tf_ops as T
    assert T.same_pads(64, 5, 2) == (32, 1, 2)        # every 5x5 / stride-2 layer: one before, two after
    assert T.same_pads(16, 3, 2) == (8, 0, 1)         # g/sconv3
    assert T.same_pads(2, 2, 1) == (2, 0, 1)          # d/conv6
    assert T.same_pads(64, 5, 1) == (64, 2, 2)        # DNA patches k = 5
    assert T.same_pads(64, 6, 1) == (64, 2, 3)        # k = 6 (train.py:54)
    assert T.same_pads(128, 11, 1) == (128, 5, 5)     # k = 11
    assert T.same_pads(7, 5, 2) == (4, 2, 2) and T.same_pads(9, 3, 1) == (9, 1, 1) and T.same_pads(8, 5, 2) == (4, 1, 2)


def test_conv2d_hand_computed():
    """3x3 input, 3x3 all-ones filter, stride 2, SAME (pad (1,1)): each output is the sum of the window that lies inside."""
    import torch
    from oracle import tf_ops as T
    x = torch.arange(1, 10, dtype=torch.float64).reshape(1, 3, 3, 1)       # 1 2 3 / 4 5 6 / 7 8 9
    y = T.conv2d(x, torch.ones(3, 3, 1, 1, dtype=torch.float64), 2, 'SAME')
    assert y.reshape(2, 2).tolist() == [[1 + 2 + 4 + 5, 2 + 3 + 5 + 6], [4 + 5 + 7 + 8, 5 + 6 + 8 + 9]]
    # 4x4 input, k 3, stride 2: pad (0,1) - the window of output 0 starts AT the first pixel
    x = torch.arange(16, dtype=torch.float64).reshape(1, 4, 4, 1)
    w = torch.zeros(3, 3, 1, 1, dtype=torch.float64)
    w[0, 0] = 1.0                                                           # picks the window's top-left pixel
    assert T.conv2d(x, w, 2, 'SAME').reshape(2, 2).tolist() == [[0.0, 2.0], [8.0, 10.0]]


def test_conv2d_transpose_is_backprop_input_not_the_textbook_idiom():
    """slim.conv2d_transpose (models.py:17-21,39-40,53-59) equals conv2d_backprop_input of the forward conv on the output
    size: y[p] = sum x[i] w[a] with p = i*s - pad_before + a, pad_before = 1 for k 5 / s 2 (A.2).  Checked against the
    brute-force index definition, against the gradient of the forward conv, and NEGATIVELY against the look-alike
    PyTorch idiom conv_transpose2d(padding=2, output_padding=1), which shifts the result by one pixel."""
    import torch
    import torch.nn.functional as F
    from oracle import tf_ops as T
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 4, 5, generator=g, dtype=torch.float64)          # [B, ih, iw, Cin]
    w = torch.randn(5, 5, 6, 5, generator=g, dtype=torch.float64)          # [kh, kw, Cout, Cin]
    y = T.conv2d_transpose(x, w, 2, 'SAME')
    assert tuple(y.shape) == (2, 6, 8, 6)
    ref = torch.zeros_like(y)                                               # brute force from the index formula
    for i in range(3):
        for j in range(4):
            for a in range(5):
                for c in range(5):
                    p, q = i * 2 - 1 + a, j * 2 - 1 + c
                    if 0 <= p < 6 and 0 <= q < 8:
                        ref[:, p, q, :] += x[:, i, j, :] @ w[a, c].T
    assert (y - ref).abs().max().item() < 1e-12
    # it is the input gradient of the forward conv (filter [kh,kw,Cout_deconv,Cin_deconv] read as HWIO of that conv)
    z = torch.zeros(2, 6, 8, 6, dtype=torch.float64, requires_grad=True)
    gz, = torch.autograd.grad(T.conv2d(z, w, 2, 'SAME'), [z], x)
    assert (y - gz).abs().max().item() < 1e-12
    wt = w.permute(3, 2, 0, 1).contiguous()
    right = F.conv_transpose2d(x.permute(0, 3, 1, 2), wt, stride=2, padding=1)[:, :, :-1, :-1].permute(0, 2, 3, 1)
    wrong = F.conv_transpose2d(x.permute(0, 3, 1, 2), wt, stride=2, padding=2, output_padding=1).permute(0, 2, 3, 1)
    assert (y - right).abs().max().item() < 1e-12
    assert tuple(wrong.shape) == tuple(y.shape) and (y - wrong).abs().max().item() > 0.1      # same shape, different tensor


def test_optimizer_known_answers():
    """TF-1.0 formulas (A.6) on scalars: Adam's first step is lr * g / (|g| + eps / sqrt(1 - beta2)); RMSProp starts its
    mean square at ONE."""
    import math
    import numpy as np
    import torch
    from oracle.trainer import TFAdam, TFRMSProp
    f32 = lambda v: float(np.float32(v))
    p = {'w': torch.tensor([1.0, -2.0], dtype=torch.float64)}
    adam = TFAdam(['w'], p)
    adam.apply(p, {'w': torch.tensor([0.5, -4.0], dtype=torch.float64)})
    lr_t = f32(1e-3) * math.sqrt(1 - f32(0.999)) / (1 - f32(0.9))
    for i, g0 in enumerate([0.5, -4.0]):
        m, v = (1 - f32(0.9)) * g0, (1 - f32(0.999)) * g0 * g0
        assert abs(float(p['w'][i]) - ([1.0, -2.0][i] - lr_t * m / (math.sqrt(v) + f32(1e-8)))) < 1e-15
    q = {'w': torch.tensor([1.0], dtype=torch.float64)}
    TFRMSProp(['w'], q).apply(q, {'w': torch.tensor([3.0], dtype=torch.float64)})
    ms = f32(0.9) * 1.0 + (1 - f32(0.9)) * 9.0
    assert abs(float(q['w'][0]) - (1.0 - f32(5e-5) * 3.0 / math.sqrt(ms + f32(1e-10)))) < 1e-15
