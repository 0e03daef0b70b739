"""Functional helpers that call the C ABI directly on torch tensors (CPU for the C oracle,
cuda for libacgan_hip.so).  Test-side only: allocation, descriptors and workspace handling so
a parity test reads `y = abi.conv2d_fwd(x, w, 2, 'SAME')`."""
import ctypes

import torch

from action_conditioned_gans_amd import _lib as L

ACT = {None: L.ACT_NONE, 'none': L.ACT_NONE, 'relu': L.ACT_RELU, 'lrelu': L.ACT_LRELU, 'tanh': L.ACT_TANH}


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Abi:
    def __init__(self, lib, device, conv_dtype=L.ACG_F32):
        self.lib, self.device, self.conv_dtype = lib, torch.device(device), conv_dtype
        self.bn_flags = 0            # `flags` of the BatchNorm entries (ACG_BN_NO_GRID_EXCHANGE): tests set it to cover that path

    # ---- plumbing
    def stream(self):
        if self.device.type == 'cuda':
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    _CANARY = 512

    def ws(self, nbytes):
        """A zeroed workspace of `nbytes` with a canary region behind it (`no_timeout` / `canary_intact` check that nothing wrote
        past the size the library asked for)."""
        n = max(int(nbytes), 16)
        buf = torch.zeros(n + self._CANARY, dtype=torch.uint8, device=self.device)
        buf[n:] = 0xA5
        view = buf[:n]
        view._acg_canary = buf
        return view, n

    @classmethod
    def canary_intact(cls, ws):
        buf = getattr(ws, '_acg_canary', None)
        if buf is not None:
            tail = buf[ws.numel():]
            assert bool((tail == 0xA5).all()), 'a kernel wrote past its workspace (%d of %d canary bytes changed)' % (int((tail != 0xA5).sum()), tail.numel())

    def bn_ws(self, rows, c, groups):
        """A BatchNorm workspace (ABI 6: call-site state, zero before first use); `no_timeout` checks its flag word afterwards."""
        return self.ws(self.lib.bn_workspace_bytes(rows, c, groups))

    @staticmethod
    def no_timeout(ws):
        """acgan_hip.h: uint32 word 2 of a BatchNorm workspace is set when a block of a one-launch kernel gave up waiting for its peers."""
        if ws.numel() >= 12:
            assert int(ws[8:12].view(torch.int32)[0]) == 0, 'BatchNorm grid exchange timed out (workspace word 2)'
        Abi.canary_intact(ws)

    def desc(self, batch, h, w, c, kh, kw, cout, stride, padding, pitch=0):
        d = L.ConvDesc()
        self.lib.conv_desc_init(ctypes.byref(d), batch, h, w, c, kh, kw, cout, stride, 1 if padding == 'SAME' else 0)
        d.in_pitch = pitch
        return d

    def sync(self):
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)

    # ---- bf16 storage (conv_dtype == ACG_BF16): activations bf16 at pitch round8(C), filters as prepared copies
    @property
    def half(self):
        return self.conv_dtype == L.ACG_BF16

    def to16(self, x):
        """fp32 [..., C] -> bf16 [..., round8(C)] with zero pad channels."""
        c = x.shape[-1]
        out = torch.zeros(*x.shape[:-1], (c + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
        out[..., :c] = x.to(self.device).to(torch.bfloat16)
        return out

    def from16(self, y, c):
        return y[..., :c].float()

    def prep_weights(self, w):
        """acg_weights_prepare_bf16 on one filter [kh,kw,A,B] -> (rm [taps,A,B8], tr [taps,B,A8])."""
        kh, kw, a, b = w.shape
        w = w.to(self.device).float().contiguous()
        rm = torch.full((kh * kw, a, (b + 7) // 8 * 8), float('nan'), dtype=torch.bfloat16, device=self.device)
        tr = torch.full((kh * kw, b, (a + 7) // 8 * 8), float('nan'), dtype=torch.bfloat16, device=self.device)
        pl = L.PrepList()
        pl.src[0], pl.rm[0], pl.tr[0], pl.taps[0], pl.a[0], pl.b[0] = w.data_ptr(), rm.data_ptr(), tr.data_ptr(), kh * kw, a, b
        self.lib.weights_prepare_bf16(ctypes.byref(pl), 1, self.stream())
        self._keep16 = (w, rm, tr)
        return rm, tr

    # ---- conv family (x NHWC, w HWIO)
    def conv2d_fwd(self, x, w, stride, padding, out_f32=False):
        """``out_f32`` (bf16 only): ACG_DTYPE2(ACG_BF16, ACG_F32) - bf16 operands, float32 result at the pitch round8."""
        b, h, wd, c = x.shape
        pitch = c if c != w.shape[2] else 0            # x carries pad channels beyond the filter's Cin
        d = self.desc(b, h, wd, w.shape[2], w.shape[0], w.shape[1], w.shape[3], stride, padding, pitch)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_FWD, self.conv_dtype))
        if self.half and out_f32:
            d.in_pitch = 0
            x16, (rm, tr) = self.to16(x[..., :d.in_c]), self.prep_weights(w)
            cp = (d.out_c + 7) // 8 * 8
            y = torch.full((b, d.out_h, d.out_w, cp), 7.0, dtype=torch.float32, device=self.device)
            self.lib.conv2d_fwd(_p(x16), _p(tr), _p(y), ctypes.byref(d), L.dtype2(L.ACG_BF16, L.ACG_F32), _p(ws), n, self.stream())
            return y[..., :d.out_c].contiguous()
        if self.half:
            d.in_pitch = 0
            x16, (rm, tr) = self.to16(x[..., :d.in_c]), self.prep_weights(w)
            y = torch.zeros(b, d.out_h, d.out_w, (d.out_c + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
            self.lib.conv2d_fwd(_p(x16), _p(tr), _p(y), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
            return self.from16(y, d.out_c)
        y = self.empty(b, d.out_h, d.out_w, d.out_c)
        self.lib.conv2d_fwd(_p(x), _p(w), _p(y), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return y

    def deconv2d_fwd_bias_act(self, x, w, bias, stride, act, leak=0.2):
        """acg_deconv2d_fwd_bias_act: y = act(conv2d_transpose(x, w) + bias), dense float32; None where the shape is not fused."""
        d = self._adj(x.shape, tuple(w.shape), stride)
        if self.half:
            d.in_pitch = d.out_pitch = 0
        if not self.lib.deconv2d_fwd_bias_act_ok(ctypes.byref(d), self.conv_dtype):
            return None
        y = torch.full((d.batch, d.in_h, d.in_w, d.in_c), 7.0, dtype=torch.float32, device=self.device)
        if self.half:
            x16, (rm, tr) = self.to16(x), self.prep_weights(w)
            self.lib.deconv2d_fwd_bias_act(_p(x16), _p(rm), _p(bias), _p(y), ctypes.byref(d), ACT[act], leak, self.conv_dtype, self.stream())
        else:
            self.lib.deconv2d_fwd_bias_act(_p(x), _p(w), _p(bias), _p(y), ctypes.byref(d), ACT[act], leak, self.conv_dtype, self.stream())
        return y

    def conv2d_dgrad(self, dy, w, x_shape, stride, padding, grad_c=0):
        """``grad_c`` > 0: acg_conv_desc.dgrad_c - only the first grad_c channels of dx are computed."""
        b, h, wd, c = x_shape
        pitch = c if c != w.shape[2] else 0
        d = self.desc(b, h, wd, w.shape[2], w.shape[0], w.shape[1], w.shape[3], stride, padding, pitch)
        d.dgrad_c = grad_c
        dx = torch.zeros(*x_shape, device=self.device)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_DGRAD, self.conv_dtype))
        if self.half:
            d.in_pitch = 0
            dy16, (rm, tr) = self.to16(dy), self.prep_weights(w)
            dx16 = torch.zeros(b, h, wd, (d.in_c + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
            self.lib.conv2d_dgrad(_p(dy16), _p(rm), _p(dx16), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
            return self.from16(dx16, d.in_c)
        self.lib.conv2d_dgrad(_p(dy), _p(w), _p(dx), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return dx

    def conv2d_wgrad(self, x, dy, w_shape, stride, padding, dw=None, accumulate=0.0):
        b, h, wd, c = x.shape
        pitch = c if c != w_shape[2] else 0
        d = self.desc(b, h, wd, w_shape[2], w_shape[0], w_shape[1], w_shape[3], stride, padding, pitch)
        if dw is None:
            dw = self.empty(*w_shape)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype))
        if self.half:
            d.in_pitch = 0
            x, dy = self.to16(x[..., :d.in_c]), self.to16(dy)
        self.lib.conv2d_wgrad(_p(x), _p(dy), _p(dw), accumulate, ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return dw

    # ---- deconv family (x NHWC [B,IH,IW,Cin], w [kh,kw,Cout,Cin]); SAME only
    def _adj(self, x_shape, w_shape, stride):
        b, ih, iw, cphys = x_shape
        kh, kw, cout, cin = w_shape
        d = self.desc(b, ih * stride, iw * stride, cout, kh, kw, cin, stride, 'SAME')
        d.out_pitch = cphys if cphys != cin else 0     # deconv input stored with pad channels
        return d

    def deconv2d_fwd(self, x, w, stride):
        d = self._adj(x.shape, w.shape, stride)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_DGRAD, self.conv_dtype))
        if self.half:
            d.out_pitch = 0
            x16, (rm, tr) = self.to16(x[..., :d.out_c]), self.prep_weights(w)
            y = torch.zeros(d.batch, d.in_h, d.in_w, (d.in_c + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
            self.lib.deconv2d_fwd(_p(x16), _p(rm), _p(y), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
            return self.from16(y, d.in_c)
        y = self.empty(d.batch, d.in_h, d.in_w, d.in_c)
        self.lib.deconv2d_fwd(_p(x), _p(w), _p(y), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return y

    def deconv2d_dgrad(self, dy, w, x_shape, stride, grad_c=0):
        """``grad_c`` > 0: acg_conv_desc.adj_dgrad_c - only the first grad_c channels of dx are computed."""
        d = self._adj(x_shape, w.shape, stride)
        d.adj_dgrad_c = grad_c
        dx = torch.zeros(*x_shape, device=self.device)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_FWD, self.conv_dtype))
        if self.half:
            d.out_pitch = 0
            dy16, (rm, tr) = self.to16(dy), self.prep_weights(w)
            dx16 = torch.zeros(*x_shape[:3], (d.out_c + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
            self.lib.deconv2d_dgrad(_p(dy16), _p(tr), _p(dx16), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
            return self.from16(dx16, d.out_c)
        self.lib.deconv2d_dgrad(_p(dy), _p(w), _p(dx), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return dx

    def deconv2d_wgrad(self, x, dy, w_shape, stride, dw=None, accumulate=0.0):
        d = self._adj(x.shape, w_shape, stride)
        if dw is None:
            dw = self.empty(*w_shape)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype))
        if self.half:
            d.out_pitch = 0
            x, dy = self.to16(x[..., :d.out_c]), self.to16(dy)
        self.lib.deconv2d_wgrad(_p(x), _p(dy), _p(dw), accumulate, ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return dw

    # ---- a layer's input gradient + weight gradient in one launch
    def bwd_pair(self, x, dy, w, stride, padding=None, transposed=False, accumulate=0.0, dw=None, slabs_only=False, grad_c=0):
        """-> (dx, dw or (slab workspace, splits)).  ``grad_c``: acg_conv_desc dgrad_c / adj_dgrad_c."""
        if transposed:
            d = self._adj(x.shape, tuple(w.shape), stride)
            d.adj_dgrad_c = grad_c
            which_d = L.CONV_FWD
        else:
            b, h, wd, c = x.shape
            d = self.desc(b, h, wd, w.shape[2], w.shape[0], w.shape[1], w.shape[3], stride, padding, c if c != w.shape[2] else 0)
            d.dgrad_c = grad_c
            which_d = L.CONV_DGRAD
        dx = torch.zeros_like(x)
        if dw is None and not slabs_only:
            dw = self.empty(*w.shape)
        wsd, nd = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), which_d, self.conv_dtype))
        wsw, nw = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype))
        fn = self.lib.deconv2d_bwd_pair if transposed else self.lib.conv2d_bwd_pair
        if self.half:          # bf16 tensors at pitch round8(C); dgrad of a conv reads the 'rm' copy, of a deconv the 'tr' copy
            d.in_pitch = d.out_pitch = 0
            cx = d.out_c if transposed else d.in_c
            x16, dy16, (rm, tr) = self.to16(x[..., :cx]), self.to16(dy), self.prep_weights(w)
            dx16 = torch.zeros(*x.shape[:3], (cx + 7) // 8 * 8, dtype=torch.bfloat16, device=self.device)
            fn(_p(dy16), _p(tr if transposed else rm), _p(x16), _p(dx16), None if slabs_only else _p(dw), accumulate, ctypes.byref(d),
               self.conv_dtype, _p(wsd), nd, _p(wsw), nw, 1 if slabs_only else 0, self.stream())
            dx = self.from16(dx16, cx)
        else:
            fn(_p(dy), _p(w), _p(x), _p(dx), None if slabs_only else _p(dw), accumulate, ctypes.byref(d), self.conv_dtype, _p(wsd), nd,
               _p(wsw), nw, 1 if slabs_only else 0, self.stream())
        if slabs_only:
            return dx, (wsw, self.lib.conv2d_splits(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype))
        return dx, dw

    # ---- deferred split-K reduction of weight gradients
    def wgrad_slabs(self, x, dy, w_shape, stride, padding=None, transposed=False):
        """-> (slab workspace, splits), or (None, 1) when the planner does not split this shape."""
        if transposed:
            d = self._adj(x.shape, w_shape, stride)
        else:
            b, h, wd, c = x.shape
            d = self.desc(b, h, wd, w_shape[2], w_shape[0], w_shape[1], w_shape[3], stride, padding, c if c != w_shape[2] else 0)
        splits = self.lib.conv2d_splits(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype)
        if splits < 2:
            return None, splits
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype))
        fn = self.lib.deconv2d_wgrad_slabs if transposed else self.lib.conv2d_wgrad_slabs
        fn(_p(x), _p(dy), ctypes.byref(d), self.conv_dtype, _p(ws), n, self.stream())
        return ws, splits

    def splitk_reduce_many(self, entries, step=None):
        """entries: (slab workspace, out tensor, splits, accumulate); ``step``: an int32 tensor the launch increments (ABI 7)."""
        rl = L.ReduceList()
        for i, (ws, out, splits, acc) in enumerate(entries):
            rl.slabs[i], rl.out[i], rl.numel[i], rl.splits[i], rl.accumulate[i] = ws.data_ptr(), out.data_ptr(), out.numel(), splits, acc
        rl.step_inc = step.data_ptr() if step is not None else None
        self.lib.splitk_reduce_many(ctypes.byref(rl), len(entries), self.stream())

    # ---- bn / bias
    def bn_act_fwd(self, x, beta, act, groups=1, eps=1e-3, leak=0.2, y_dtype=None, c=None):
        """Storage types follow the tensors: x float32 or bfloat16, y like x unless ``y_dtype`` says otherwise.
        ``c`` < x.shape[-1]: x rows carry pad channels (pitch x.shape[-1]); y is then dense [.., c]."""
        xp = x.shape[-1]
        c = c or xp
        rows = x.numel() // xp
        y = torch.zeros(*x.shape[:-1], c, dtype=y_dtype or x.dtype, device=self.device)
        mean, rstd = self.empty(groups * c), self.empty(groups * c)
        ws, n = self.ws(self.lib.bn_workspace_bytes(rows, c, groups))
        self.lib.bn_act_fwd(_p(x), _p(beta), _p(y), _p(mean), _p(rstd), rows, c, xp, c, groups, eps, ACT[act], leak,
                            L.dtype2(L.code(x.dtype), L.code(y.dtype)), self.bn_flags, _p(ws), n, self.stream())
        self.no_timeout(ws)
        return y, mean, rstd

    def bn_act_bwd(self, x, dy, beta, mean, rstd, act, groups=1, leak=0.2, dbeta=None, accumulate=0.0, dx_dtype=None):
        """``dx_dtype`` bfloat16 with float32 x / dy: ACG_DTYPE2(ACG_F32, ACG_BF16), the float32 head of a bf16 network."""
        xp, c = x.shape[-1], dy.shape[-1]
        rows = x.numel() // xp
        dx = torch.zeros_like(x, dtype=dx_dtype or x.dtype)
        if dbeta is None:
            dbeta = self.empty(c)
        ws, n = self.ws(self.lib.bn_workspace_bytes(rows, c, groups))
        dt = L.dtype2(L.code(x.dtype), L.code(dy.dtype))
        if dx.dtype != x.dtype:
            assert x.dtype == torch.float32 and dy.dtype == torch.float32 and dx.dtype == torch.bfloat16
            dt = L.dtype2(L.ACG_F32, L.ACG_BF16)
        self.lib.bn_act_bwd(_p(x), _p(dy), _p(beta), _p(mean), _p(rstd), _p(dx), _p(dbeta), accumulate, rows, c, xp, c,
                            groups, ACT[act], leak, dt, self.bn_flags, _p(ws), n, self.stream())
        self.no_timeout(ws)
        return dx, dbeta

    # ---- BatchNorm statistics out of the producing convolution's epilogue
    def conv_bn_fused(self, x, w, beta, stride, padding, act, groups=1, transposed=False, eps=1e-3, leak=0.2):
        """acg_(de)conv2d_fwd_stats + acg_bn_act_fwd_partials -> (conv output, y, mean, rstd) as float32, or None when
        acg_conv2d_stats_blocks says this shape provides no partials."""
        if transposed:
            d = self._adj(x.shape, tuple(w.shape), stride)
            which, c, oshape = L.CONV_DGRAD, d.in_c, (d.batch, d.in_h, d.in_w)
        else:
            b, h, wd, cin = x.shape
            d = self.desc(b, h, wd, w.shape[2], w.shape[0], w.shape[1], w.shape[3], stride, padding)
            which, c, oshape = L.CONV_FWD, d.out_c, (b, d.out_h, d.out_w)
        brows, rrows = ctypes.c_int32(0), ctypes.c_int32(0)
        nblk = self.lib.conv2d_stats_layout(ctypes.byref(d), which, self.conv_dtype, groups, ctypes.byref(brows), ctypes.byref(rrows))
        assert nblk == self.lib.conv2d_stats_blocks(ctypes.byref(d), which, self.conv_dtype, groups)
        if nblk <= 0:
            return None
        brows, rrows = brows.value, rrows.value
        part = torch.full((groups * nblk * 2 * c,), float('nan'), device=self.device)
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), which, self.conv_dtype))
        fn = self.lib.deconv2d_fwd_stats if transposed else self.lib.conv2d_fwd_stats
        rows = oshape[0] * oshape[1] * oshape[2]
        mean, rstd = self.empty(groups * c), self.empty(groups * c)
        if self.half:
            cp = (c + 7) // 8 * 8
            x16, (rm, tr) = self.to16(x), self.prep_weights(w)
            conv = torch.zeros(*oshape, cp, dtype=torch.bfloat16, device=self.device)
            fn(_p(x16), _p(rm if transposed else tr), _p(conv), ctypes.byref(d), self.conv_dtype, _p(ws), n, _p(part), groups, self.stream())
            y = torch.zeros(*oshape, cp, dtype=torch.bfloat16, device=self.device)
            self.lib.bn_act_fwd_partials(_p(conv), _p(beta), _p(part), nblk, brows, rrows, _p(y), _p(mean), _p(rstd), rows, c, cp, cp, groups, eps,
                                         ACT[act], leak, L.dtype2(L.ACG_BF16, L.ACG_BF16), self.stream())
            return self.from16(conv, c), self.from16(y, c), mean, rstd
        conv = self.empty(*oshape, c)
        fn(_p(x), _p(w), _p(conv), ctypes.byref(d), self.conv_dtype, _p(ws), n, _p(part), groups, self.stream())
        y = self.empty(*oshape, c)
        self.lib.bn_act_fwd_partials(_p(conv), _p(beta), _p(part), nblk, brows, rrows, _p(y), _p(mean), _p(rstd), rows, c, c, c, groups, eps, ACT[act],
                                     leak, L.dtype2(L.ACG_F32, L.ACG_F32), self.stream())
        return conv, y, mean, rstd

    # ---- split-K hand-off: the contraction leaves its slabs, the layer's BatchNorm kernel sums them
    def _layer_desc(self, x_shape, w_shape, stride, padding, transposed):
        if transposed:
            d = self._adj(x_shape, tuple(w_shape), stride)
            which, c, oshape = L.CONV_DGRAD, d.in_c, (d.batch, d.in_h, d.in_w)
        else:
            b, h, wd, cin = x_shape
            d = self.desc(b, h, wd, w_shape[2], w_shape[0], w_shape[1], w_shape[3], stride, padding)
            which, c, oshape = L.CONV_FWD, d.out_c, (b, d.out_h, d.out_w)
        if self.half:
            d.in_pitch = d.out_pitch = 0
        return d, which, c, oshape

    def conv_bn_handoff(self, x, w, beta, stride, padding, act, groups=1, transposed=False, layout=None, eps=1e-3, leak=0.2):
        """acg_(de)conv2d_fwd_slabs + acg_bn_act_fwd_slabs -> (conv output as written back, y, mean, rstd, layout), float32;
        None when the planner does not split this layer.  ``layout`` None: what acg_bn_slabs_layout asks for."""
        d, which, c, oshape = self._layer_desc(x.shape, w.shape, stride, padding, transposed)
        splits = self.lib.conv2d_splits(ctypes.byref(d), which, self.conv_dtype)
        if splits < 2:
            return None
        cp = (c + 7) // 8 * 8 if self.half else c
        rows = oshape[0] * oshape[1] * oshape[2]
        if layout is None:
            layout = self.lib.bn_slabs_layout(rows, c, cp, cp, groups, self.conv_dtype, 0, self.bn_flags)
        assert layout >= 0
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), which, self.conv_dtype))
        ws.fill_(0xFF)                                                  # NaN patterns: every slab element read must have been written
        fn = self.lib.deconv2d_fwd_slabs if transposed else self.lib.conv2d_fwd_slabs
        tdt = torch.bfloat16 if self.half else torch.float32
        if self.half:
            x16, (rm, tr) = self.to16(x), self.prep_weights(w)
            fn(_p(x16), _p(rm if transposed else tr), ctypes.byref(d), self.conv_dtype, layout, _p(ws), n, self.stream())
        else:
            fn(_p(x), _p(w), ctypes.byref(d), self.conv_dtype, layout, _p(ws), n, self.stream())
        conv = torch.zeros(*oshape, cp, dtype=tdt, device=self.device)
        y = torch.zeros(*oshape, cp, dtype=tdt, device=self.device)
        mean, rstd = self.empty(groups * c), self.empty(groups * c)
        bws, bn = self.ws(self.lib.bn_workspace_bytes(rows, c, groups))
        self.lib.bn_act_fwd_slabs(_p(ws), splits, _p(conv), _p(beta), _p(y), _p(mean), _p(rstd), rows, c, cp, cp, groups, eps, ACT[act], leak,
                                  self.conv_dtype, layout, self.bn_flags, _p(bws), bn, self.stream())
        self.no_timeout(bws)
        return conv[..., :c].float(), y[..., :c].float(), mean, rstd, layout

    def dgrad_bn_bwd_handoff(self, xb, beta, mean, rstd, act, dy2, w2, stride, padding, groups=1, transposed=False, pair_x=None, layout=None,
                             leak=0.2):
        """BatchNorm backward fed by the split input gradient of the NEXT layer (filter ``w2``, output gradient ``dy2``; its
        input is the BatchNorm's output, shaped like ``xb``): acg_(de)conv2d_dgrad_slabs - or, with ``pair_x`` (that layer's
        input), acg_(de)conv2d_bwd_pair with flag 2 - then acg_bn_act_bwd_slabs.  -> (dx, dbeta, layout) float32, or None
        when that input gradient is not split or the BatchNorm cannot take slabs."""
        if transposed:
            d = self._adj(xb.shape, tuple(w2.shape), stride)
            which = L.CONV_FWD
        else:
            b, h, wd, c = xb.shape
            d = self.desc(b, h, wd, w2.shape[2], w2.shape[0], w2.shape[1], w2.shape[3], stride, padding)
            which = L.CONV_DGRAD
        if self.half:
            d.in_pitch = d.out_pitch = 0
        c = xb.shape[-1]
        cp = (c + 7) // 8 * 8 if self.half else c
        rows = xb.numel() // c
        splits = self.lib.conv2d_splits(ctypes.byref(d), which, self.conv_dtype)
        want = self.lib.bn_slabs_layout(rows, c, cp, cp, groups, self.conv_dtype, 1, self.bn_flags)
        if splits < 2 or want < 0:
            return None
        layout = want if layout is None else layout
        ws, n = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), which, self.conv_dtype))
        ws.fill_(0xFF)
        tdt = torch.bfloat16 if self.half else torch.float32
        if self.half:
            dy16, (rm, tr) = self.to16(dy2), self.prep_weights(w2)
            wd_ = tr if transposed else rm
        else:
            dy16, wd_ = dy2, w2
        if pair_x is None:
            fn = self.lib.deconv2d_dgrad_slabs if transposed else self.lib.conv2d_dgrad_slabs
            fn(_p(dy16), _p(wd_), ctypes.byref(d), self.conv_dtype, layout, _p(ws), n, self.stream())
        else:
            fn = self.lib.deconv2d_bwd_pair if transposed else self.lib.conv2d_bwd_pair
            px = self.to16(pair_x) if self.half else pair_x
            dw = self.empty(*w2.shape)
            wsw, nw = self.ws(self.lib.conv2d_workspace_bytes(ctypes.byref(d), L.CONV_WGRAD, self.conv_dtype))
            fn(_p(dy16), _p(wd_), _p(px), None, _p(dw), 0.0, ctypes.byref(d), self.conv_dtype, _p(ws), n, _p(wsw), nw,
               2 | (4 if layout == L.SLABS_QUADS else 0), self.stream())
        x16 = self.to16(xb) if self.half else xb
        dx = torch.zeros(*xb.shape[:-1], cp, dtype=tdt, device=self.device)
        dbeta = self.empty(c)
        bws, bn = self.ws(self.lib.bn_workspace_bytes(rows, c, groups))
        self.lib.bn_act_bwd_slabs(_p(x16), _p(ws), splits, _p(beta), _p(mean), _p(rstd), _p(dx), _p(dbeta), 0.0, rows, c, cp, cp, groups,
                                  ACT[act], leak, self.conv_dtype, layout, self.bn_flags, _p(bws), bn, self.stream())
        self.no_timeout(bws)
        return dx[..., :c].float(), dbeta, layout

    # ---- synchronised BatchNorm entries (statistics supplied by the caller)
    def bn_moments(self, x, groups=1, c=None):
        """``c`` < x.shape[-1]: rows carry pad channels (pitch x.shape[-1]); x float32 or bfloat16."""
        xp = x.shape[-1]
        c = c or xp
        rows = x.numel() // xp
        mom = self.empty(groups * 2 * c)
        ws, n = self.ws(self.lib.bn_workspace_bytes(rows, c, groups))
        self.lib.bn_moments(_p(x), _p(mom), rows, c, xp, groups, L.code(x.dtype), _p(ws), n, self.stream())
        return mom

    def bn_act_fwd_moments(self, x, beta, moments, act, groups=1, eps=1e-3, leak=0.2, c=None, y_dtype=None):
        xp = x.shape[-1]
        c = c or xp
        rows = x.numel() // xp
        dense = y_dtype is not None and y_dtype != x.dtype
        y = torch.zeros(*x.shape[:-1], c if dense else xp, dtype=y_dtype or x.dtype, device=self.device)
        mean, rstd = self.empty(groups * c), self.empty(groups * c)
        self.lib.bn_act_fwd_moments(_p(x), _p(beta), _p(moments), _p(y), _p(mean), _p(rstd), rows, c, xp, y.shape[-1], groups, eps,
                                    ACT[act], leak, L.dtype2(L.code(x.dtype), L.code(y.dtype)), self.stream())
        return y, mean, rstd

    def bn_bwd_sums(self, x, dy, beta, mean, rstd, act, groups=1, leak=0.2, c=None):
        xp = x.shape[-1]
        c = c or xp
        rows = x.numel() // xp
        sums = self.empty(groups * 2 * c)
        ws, n = self.ws(self.lib.bn_workspace_bytes(rows, c, groups))
        self.lib.bn_bwd_sums(_p(x), _p(dy), _p(beta), _p(mean), _p(rstd), _p(sums), rows, c, xp, dy.shape[-1], groups, ACT[act], leak,
                             L.dtype2(L.code(x.dtype), L.code(dy.dtype)), _p(ws), n, self.stream())
        return sums

    def bn_act_bwd_sums(self, x, dy, beta, mean, rstd, sums, local_sums, total_rows, act, groups=1, leak=0.2, c=None, dx_dtype=None):
        xp = x.shape[-1]
        c = c or xp
        rows = x.numel() // xp
        dx, dbeta = torch.zeros_like(x, dtype=dx_dtype or x.dtype), self.empty(c)
        dt = L.dtype2(L.code(x.dtype), L.code(dy.dtype))
        if dx.dtype != x.dtype:
            dt = L.dtype2(L.ACG_F32, L.ACG_BF16)
        self.lib.bn_act_bwd_sums(_p(x), _p(dy), _p(beta), _p(mean), _p(rstd), _p(sums), _p(local_sums), total_rows, _p(dx),
                                 _p(dbeta), 0.0, rows, c, xp, dy.shape[-1], groups, ACT[act], leak, dt, self.stream())
        return dx, dbeta

    def bias_act_fwd(self, x, bias, act, leak=0.2, c=None, y_dtype=None):
        """``c`` < x.shape[-1]: x rows carry pad channels (pitch x.shape[-1]); y is dense [.., c] of ``y_dtype``."""
        xp = x.shape[-1]
        c = c or xp
        y = torch.empty(*x.shape[:-1], c, dtype=y_dtype or x.dtype, device=self.device)
        self.lib.bias_act_fwd(_p(x), _p(bias), _p(y), x.numel() // xp, c, xp, c, ACT[act], leak,
                              L.dtype2(L.code(x.dtype), L.code(y.dtype)), self.stream())
        return y

    def bias_act_bwd(self, y, dy, act, leak=0.2, want_dx=True, x_pitch=None, x_dtype=None):
        c = y.shape[-1]
        rows = y.numel() // c
        xp = x_pitch or c
        dx = torch.zeros(*y.shape[:-1], xp, dtype=x_dtype or y.dtype, device=self.device) if want_dx else None
        dbias = self.empty(c)
        ws, n = self.ws(self.lib.bias_workspace_bytes(rows, c))
        self.lib.bias_act_bwd(_p(y), _p(dy), _p(dx), _p(dbias), 0.0, rows, c, xp, c, ACT[act], leak,
                              L.dtype2(L.code(x_dtype or y.dtype), L.code(y.dtype)), _p(ws), n, self.stream())
        return dx, dbias

    # ---- cdna
    def cdna_fwd(self, params, img, masks, k, shift=1e-12):
        b, h, w, c = img.shape
        out = torch.empty((masks, b, h, w, c), dtype=torch.float32, device=self.device)
        kn = torch.empty((b, k * k * masks), dtype=torch.float32, device=self.device)
        self.lib.cdna_fwd(_p(params), _p(img), _p(out), _p(kn), b, h, w, c, masks, k, shift, L.ACG_F32, self.stream())
        return out, kn

    def cdna_bwd(self, params, kn, img, dout, masks, k, shift=1e-12, want_dimg=True):
        b, h, w, c = img.shape
        dpar = torch.empty_like(params)
        dimg = torch.empty_like(img) if want_dimg else None
        nb = self.lib.cdna_workspace_bytes(b, h, w, c, masks, k)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=self.device)
        self.lib.cdna_bwd(_p(params), _p(kn), _p(img), _p(dout), _p(dpar), _p(dimg) if want_dimg else None, b, h, w, c,
                          masks, k, shift, L.ACG_F32, _p(ws), nb, self.stream())
        return dpar, dimg

    # ---- dna
    def dna_fwd(self, logits, img, k, bias=None, out2=None, out2_off=0):
        """logits float32 [B,H,W,k*k] or bfloat16 [B,H,W,round8(k*k)].  ``out2`` [B,H,W,pitch] (float32 / bfloat16): the frame is
        also written into its channels [out2_off, out2_off + C)."""
        b, h, w, c = img.shape
        out = torch.empty_like(img)
        self.lib.dna_fwd(_p(logits), _p(bias), _p(img), _p(out), _p(out2), out2.shape[-1] if out2 is not None else 0, out2_off,
                         L.code(out2.dtype) if out2 is not None else 0, b, h, w, c, k, L.code(logits.dtype), self.stream())
        return out

    def dna_bwd(self, logits, img, dout, k, bias=None, want_dbias=False, dout2=None, dout2_off=0):
        """``dout2`` [B,H,W,pitch]: its channels [dout2_off, dout2_off + C) are added to dout."""
        b, h, w, c = img.shape
        dl = torch.zeros_like(logits)
        dbias = self.empty(k * k) if want_dbias else None
        ws, n = self.ws(self.lib.dna_workspace_bytes(b, h, w, k))
        self.lib.dna_bwd(_p(logits), _p(bias), _p(img), _p(dout), _p(dout2), dout2.shape[-1] if dout2 is not None else 0, dout2_off,
                         L.code(dout2.dtype) if dout2 is not None else 0, _p(dl), _p(dbias), 0.0, b, h, w, c, k, L.code(logits.dtype),
                         _p(ws), n, self.stream())
        return (dl, dbias) if want_dbias else dl

    # ---- plumbing ops
    def concat_actions(self, x, actions, pitch=0):
        b, h, w, c = x.shape
        a = actions.shape[1]
        y = torch.zeros(b, h, w, pitch or (c + a), dtype=x.dtype, device=self.device)
        self.lib.concat_actions_fwd(_p(x), _p(actions), _p(y), b, h * w, c, a, pitch, L.code(x.dtype), self.stream())
        return y

    def concat_channels(self, a, b, pitch=0, y_dtype=None):
        ca, cb = a.shape[-1], (b.shape[-1] if b is not None else 0)
        y = torch.zeros(*a.shape[:-1], pitch or (ca + cb), dtype=y_dtype or a.dtype, device=self.device)
        self.lib.concat_channels_fwd(_p(a), _p(b), _p(y), a.numel() // ca, ca, cb, pitch, L.dtype2(L.code(a.dtype), L.code(y.dtype)),
                                     self.stream())
        return y

    def slice_channels(self, src, off, cdst, dst=None, accumulate=0.0, dst_dtype=None):
        cs = src.shape[-1]
        if dst is None:
            dst = self.empty(*src.shape[:-1], cdst, dtype=dst_dtype or src.dtype)
        self.lib.slice_channels(_p(src), _p(dst), accumulate, src.numel() // cs, cs, off, cdst,
                                L.dtype2(L.code(src.dtype), L.code(dst.dtype)), self.stream())
        return dst

    def copy_many(self, pairs):
        """pairs: [(src [rows, cols] dense, dst [rows, pitch])]; copies every src into the first cols channels of dst."""
        cl = L.CopyList()
        for i, (src, dst) in enumerate(pairs):
            cl.src[i], cl.dst[i] = src.data_ptr(), dst.data_ptr()
            cl.rows[i], cl.cols[i], cl.dst_pitch[i], cl.dst_dtype[i] = src.shape[0], src.shape[1], dst.shape[1], L.code(dst.dtype)
        self.lib.copy_many(ctypes.byref(cl), len(pairs), L.ACG_F32, self.stream())

    def add(self, a, b):
        y = torch.empty_like(a)
        self.lib.add(_p(a), _p(b), _p(y), a.numel(), L.code(a.dtype), self.stream())
        return y

    # ---- losses
    def frame_loss(self, gen, gt, w_l1, w_gdl, want_grad=True, want_values=True):
        b, h, w, c = gen.shape
        out = self.empty(2) if want_values else None
        dgen = torch.empty_like(gen) if want_grad else None
        ws, n = self.ws(self.lib.frame_loss_workspace_bytes(gen.numel()))
        self.lib.frame_loss(_p(gen), _p(gt), _p(out), _p(dgen), b, h, w, c, w_l1, w_gdl, L.ACG_F32, _p(ws), n,
                            self.stream())
        return out, dgen

    def l2norm_loss(self, pred, gt, scale):
        out, d = self.empty(1), torch.empty_like(pred)
        self.lib.l2norm_loss(_p(pred), _p(gt), _p(out), _p(d), pred.numel(), scale, self.stream())
        return out, d

    def sigmoid_ce_loss(self, logits, label, scale):
        out, d = self.empty(1), torch.empty_like(logits)
        self.lib.sigmoid_ce_loss(_p(logits), label, _p(out), _p(d), logits.numel(), scale, self.stream())
        return out, d

    def mean_loss(self, x, scale):
        out, d = self.empty(1), torch.empty_like(x)
        self.lib.mean_loss(_p(x), _p(out), _p(d), x.numel(), scale, self.stream())
        return out, d

    def psnr(self, a, b):
        out = self.empty(1)
        ws, n = self.ws(self.lib.frame_loss_workspace_bytes(a.numel()))
        self.lib.psnr(_p(a), _p(b), _p(out), a.numel(), L.ACG_F32, _p(ws), n, self.stream())
        return out

    def scalar_combine(self, terms):
        out = self.empty(1)
        args = []
        for i in range(4):
            t, w = terms[i] if i < len(terms) else (None, 0.0)
            args += [_p(t), w]
        self.lib.scalar_combine(_p(out), *args, self.stream())
        return out

    # ---- optimizers (in place)
    def adam_step(self, p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, gs=1.0, clip=None):
        self.lib.step_inc(_p(step), self.stream())
        lo, hi = clip if clip else (0.0, 0.0)
        self.lib.adam_step(_p(p), _p(g), _p(m), _p(v), _p(step), p.numel(), lr, b1, b2, eps, gs, 1 if clip else 0,
                           lo, hi, self.stream())

    def rmsprop_step(self, p, g, ms, lr=5e-5, decay=0.9, eps=1e-10, gs=1.0, clip=None):
        lo, hi = clip if clip else (0.0, 0.0)
        self.lib.rmsprop_step(_p(p), _p(g), _p(ms), p.numel(), lr, decay, eps, gs, 1 if clip else 0, lo, hi,
                              self.stream())

    def clip(self, p, lo, hi):
        self.lib.clip(_p(p), p.numel(), lo, hi, self.stream())
