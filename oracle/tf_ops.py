"""torch-CPU restatement of the TF-1.0 / slim ops the reference composes.

TEST INFRASTRUCTURE (see oracle/__init__.py) - parity unpinned.

All tensors are NHWC ``torch.Tensor`` on CPU, fp64 or fp32.  Conv filters are HWIO
``[kh, kw, Cin, Cout]``; conv2d_transpose filters are ``[kh, kw, Cout, Cin]`` exactly as
slim creates them.  Each function cites the reference line that invokes the TF op it
restates (paths are into /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- bf16 storage emulation
# BASELINE configs 3 and 5 keep activations (and their gradients) in bfloat16 in memory; everything is still computed
# and accumulated in higher precision.  ``with bf16_storage():`` makes the model restatement (oracle/models.py) round
# exactly the tensors the HIP pipeline stores as bf16 - conv operands and outputs, BatchNorm+activation outputs, the
# action-concatenated maps, the discriminator input, and on the way back the gradients of those tensors - while
# filters round in the forward pass only (master weights and weight gradients stay float32 there too).
_BF16 = [False]


class _RoundBoth(torch.autograd.Function):
    """forward: x -> bf16(x) (round to nearest even); backward: g -> bf16(g).  A tensor that is stored as bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    """forward: bf16 copy of a float32 master tensor; backward: identity (its gradient is accumulated in float32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """forward: identity (the tensor is kept in float32); backward: g -> bf16(g) (its gradient is stored as bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def q_act(x):
    return _RoundBoth.apply(x) if _BF16[0] else x


def q_grad(x):
    """A float32 tensor inside a bf16 network whose gradient goes back as bf16 (the conv output of a head layer)."""
    return _RoundBwd.apply(x) if _BF16[0] else x


def q_weight(w):
    return _RoundFwd.apply(w) if _BF16[0] else w


class bf16_storage:
    def __enter__(self):
        self._prev = _BF16[0]
        _BF16[0] = True

    def __exit__(self, *a):
        _BF16[0] = self._prev
        return False


# --------------------------------------------------------------------------- padding
def same_pads(in_size, k, s):
    """TF 'SAME' geometry: (out, pad_before, pad_after).  SURVEY Appendix A.1."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv_geometry(h, w, kh, kw, sh, sw, padding):
    if padding == 'SAME':
        oh, pt, pb = same_pads(h, kh, sh)
        ow, pl, pr = same_pads(w, kw, sw)
    elif padding == 'VALID':
        oh, pt, pb = (h - kh) // sh + 1, 0, 0
        ow, pl, pr = (w - kw) // sw + 1, 0, 0
    else:
        raise ValueError('unexpected padding argument')
    return oh, ow, pt, pb, pl, pr


# --------------------------------------------------------------------------- convs
def conv2d(x, w, stride=1, padding='SAME'):
    """tf.nn.conv2d as used by slim.conv2d (models.py:12-15,34-37,42-51,82-88).

    y[b,p,q,o] = sum_{i,j,c} x[b, p*s - pt + i, q*s - pl + j, c] * w[i,j,c,o]
    """
    kh, kw = w.shape[0], w.shape[1]
    _, _, pt, pb, pl, pr = conv_geometry(x.shape[1], x.shape[2], kh, kw, stride, stride, padding)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn.contiguous(), w.permute(3, 2, 0, 1).contiguous(), stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def conv2d_transpose(x, w, stride=2, padding='SAME'):
    """tf.nn.conv2d_transpose as used by slim.conv2d_transpose (models.py:17-21,39-40,53-59).

    Equals conv2d_backprop_input of the forward conv whose *input* is the deconv output:
    y[b,p,q,o] = sum_{i,j,a,c'} x[b,i,j,c] * w[a,c',o,c]   with p = i*s - pt + a, q = j*s - pl + c'.
    SAME: output spatial = input * stride.  No kernel flip.
    """
    kh, kw, cout, cin = w.shape
    ih, iw = x.shape[1], x.shape[2]
    if padding == 'SAME':
        oh, ow = ih * stride, iw * stride
    elif padding == 'VALID':
        oh, ow = (ih - 1) * stride + kh, (iw - 1) * stride + kw
    else:
        raise ValueError('unexpected padding argument')
    _, _, pt, _, pl, _ = conv_geometry(oh, ow, kh, kw, stride, stride, padding)
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2).contiguous(), w.permute(3, 2, 0, 1).contiguous(), stride=stride)
    # full[p'] with p' = i*s + a ; TF index p = p' - pt.  Zero-extend if the crop overruns.
    need_h, need_w = pt + oh, pl + ow
    full = F.pad(full, (0, max(need_w - full.shape[3], 0), 0, max(need_h - full.shape[2], 0)))
    y = full[:, :, pt:pt + oh, pl:pl + ow]
    return y.permute(0, 2, 3, 1).contiguous()


# --------------------------------------------------------------------------- norm / act
def batch_norm_train(x, beta, eps=1e-3):
    """slim.batch_norm defaults in training mode (implicit via argscope, models.py:10-11,31-32,80-81).

    center=True, scale=False, epsilon=1e-3, batch moments over (B,H,W), biased variance.
    """
    mean = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) + beta


def relu(x):
    return torch.relu(x)


def lrelu(x, leak=0.2):
    """ops.py:22-26  -> f1*x + f2*|x| with f1=.5(1+leak), f2=.5(1-leak)."""
    f1 = 0.5 * (1 + leak)
    f2 = 0.5 * (1 - leak)
    return f1 * x + f2 * x.abs()


# --------------------------------------------------------------------------- DNA tail
def extract_image_patches(img, ksize):
    """tf.extract_image_patches(ksizes=[1,k,k,1], strides 1, rates 1, SAME) (models.py:62-66).

    Returns [B,H,W,k*k,C] with patch depth ordered (row i, col j, channel).
    """
    b, h, w, c = img.shape
    _, pt, pb = same_pads(h, ksize, 1)
    _, pl, pr = same_pads(w, ksize, 1)
    xp = F.pad(img.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    taps = []
    for i in range(ksize):
        for j in range(ksize):
            taps.append(xp[:, :, i:i + h, j:j + w])
    out = torch.stack(taps, dim=1)                      # [B, k*k, C, H, W]
    return out.permute(0, 3, 4, 1, 2).contiguous()      # [B, H, W, k*k, C]


def cdna_transform(params, img, num_masks, ksize=5, relu_shift=1e-12):
    """cdna_transformation after its fully-connected layer (reference ops.py:77-98): params [B, k*k*M] read as
    [B,k,k,1,M]; k = relu(p - shift) + shift, normalised over (k, k, 1); per-sample depthwise SAME correlation
    (tf.nn.depthwise_conv2d: output channel c*M + m); concatenated over the batch and split into M pieces of C
    channels along the channel axis.  Returns the list of M tensors [B,H,W,C]."""
    B, H, W, C = img.shape
    M, k = num_masks, ksize
    kern = params.reshape(B, k, k, 1, M)
    kern = torch.relu(kern - relu_shift) + relu_shift
    kern = kern / kern.sum(dim=(1, 2, 3), keepdim=True)
    pad = (k - 1) // 2
    outs = []
    for b in range(B):
        w = kern[b, :, :, 0, :].permute(2, 0, 1)                                    # [M,k,k]
        w = w.unsqueeze(0).expand(C, M, k, k).reshape(C * M, 1, k, k)               # group c, multiplier m -> channel c*M+m
        x = img[b].permute(2, 0, 1).unsqueeze(0)                                    # [1,C,H,W]
        x = F.pad(x, (pad, k - 1 - pad, pad, k - 1 - pad))
        y = F.conv2d(x, w, groups=C)                                                # [1,C*M,H,W]
        outs.append(y[0].permute(1, 2, 0))                                          # [H,W,C*M]
    t = torch.stack(outs)                                                           # [B,H,W,C*M]
    return [t[..., j * C:(j + 1) * C] for j in range(M)]


def dna_gather(logits, img, ksize):
    """models.py:60-72: softmax over k*k logits, per-pixel weighted sum of the k x k window."""
    m = torch.softmax(logits, dim=-1)
    patches = extract_image_patches(img, ksize)
    return (m.unsqueeze(-1) * patches).sum(dim=3)


# --------------------------------------------------------------------------- losses
def sigmoid_cross_entropy(labels, logits):
    """tf.losses.sigmoid_cross_entropy (ops.py:30-31,39-42): mean over all elements."""
    x, z = logits, labels
    return (torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))).mean()


def g_adv_loss(d_out_gen, arg_loss):
    """ops.py:28-35."""
    if arg_loss == 'bce':
        return sigmoid_cross_entropy(torch.ones_like(d_out_gen), d_out_gen)
    elif arg_loss == 'wass':
        return d_out_gen.mean()
    raise ValueError('unexpected loss argument')


def d_loss(d_out_direct, d_out_gen, arg_loss):
    """ops.py:37-50.  Returns (total, direct, gen)."""
    if arg_loss == 'bce':
        direct = sigmoid_cross_entropy(0.9 * torch.ones_like(d_out_direct), d_out_direct)
        gen = sigmoid_cross_entropy(torch.zeros_like(d_out_gen), d_out_gen)
    elif arg_loss == 'wass':
        direct = d_out_direct.mean()
        gen = -d_out_gen.mean()
    else:
        raise ValueError('unexpected loss argument')
    return direct + gen, direct, gen


def gdl(a, b, alpha=1):
    """ops.py:100-120 gradient-difference loss (symmetric in its two arguments).

    dx[x] = in[x+1]-in[x] (zero beyond the right edge), dy[y] = in[y]-in[y+1] (zero beyond
    the bottom edge); sum (not mean) of | |d gt| - |d gen| |^alpha over both directions.
    """
    def dxy(t):
        tx = F.pad(t, (0, 0, 0, 1))              # pad W by one on the right
        ty = F.pad(t, (0, 0, 0, 0, 0, 1))        # pad H by one at the bottom
        dx = tx[:, :, 1:, :] - tx[:, :, :-1, :]
        dy = ty[:, :-1, :, :] - ty[:, 1:, :, :]
        return dx.abs(), dy.abs()
    adx, ady = dxy(a)
    bdx, bdy = dxy(b)
    return ((adx - bdx).abs() ** alpha + (ady - bdy).abs() ** alpha).sum()


def l1_over_batch(a, b, batch):
    """train.py:73  tf.norm(ord=1, axis=None) / BATCH_SIZE."""
    return (a - b).abs().sum() / batch


def l2norm_over_batch(a, b, batch):
    """train.py:77  tf.norm(ord=2, axis=None) / BATCH_SIZE."""
    return torch.sqrt(((a - b) ** 2).sum()) / batch


def psnr(true, pred):
    """ops.py:19-20."""
    mse = ((true - pred) ** 2).mean()
    return 10.0 * torch.log(1.0 / mse) / math.log(10.0)


# --------------------------------------------------------------------------- host helpers
def build_all_mask(num_frame):
    """util.py:10-16: one-hot rows selecting frame t, t in [0, T-2]."""
    m = np.zeros((num_frame - 1, num_frame), dtype=bool)
    m[np.arange(num_frame - 1), np.arange(num_frame - 1)] = True
    return m


def xavier_uniform_(shape, fan_in, fan_out, gen, dtype=torch.float32):
    """slim.xavier_initializer(uniform=True): U(-l, l), l = sqrt(6 / (fan_in + fan_out))."""
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(limit).to(dtype)
