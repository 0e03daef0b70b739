"""ctypes binding of libacgan_hip.so (include/acgan_hip.h).

There is no fallback: if the HIP library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C action_conditioned_gans_amd/csrc``) every
operator raises ``RuntimeError``.  Nothing here knows about the CPU oracle.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

# torch FIRST: it carries its own copy of the HIP runtime, and the process must end up with ONE.  Loaded after torch,
# libacgan_hip.so binds to the runtime torch has already mapped (streams, device memory and graphs are then shared);
# loaded before it, the library would initialise the system runtime and torch a second one, and the first kernel
# launch from here fails with "no ROCm-capable device is detected" (seen in build() -> smoke() in one process).
import torch  # noqa: F401  (import order is the point)

ACG_F32, ACG_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
CONV_FWD, CONV_DGRAD, CONV_WGRAD = 0, 1, 2
SLABS_ROWS, SLABS_QUADS = 0, 1       # acgan_hip.h ACG_SLABS_*
BN_NO_GRID_EXCHANGE = 1               # acgan_hip.h ACG_BN_NO_GRID_EXCHANGE
ABI_VERSION = 8       # include/acgan_hip.h ACG_ABI_VERSION: bumped with every signature / layout / flag-meaning change

LIB_NAME = 'libacgan_hip.so'
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', LIB_NAME)


class ConvDesc(ctypes.Structure):
    """struct acg_conv_desc."""
    _fields_ = [(n, c_int32) for n in (
        'batch', 'in_h', 'in_w', 'in_c', 'out_h', 'out_w', 'out_c', 'kh', 'kw',
        'stride_h', 'stride_w', 'pad_top', 'pad_left', 'in_pitch', 'out_pitch', 'dgrad_c', 'adj_dgrad_c')]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class ReduceList(ctypes.Structure):
    """struct acg_reduce_list (ACG_REDUCE_MAX = 32 entries)."""
    _fields_ = [('slabs', c_void_p * 32), ('out', c_void_p * 32), ('numel', c_int64 * 32), ('splits', c_int32 * 32),
                ('accumulate', c_float * 32), ('step_inc', c_void_p)]


class PrepList(ctypes.Structure):
    """struct acg_prep_list (ACG_PREP_MAX = 32 filters)."""
    _fields_ = [('src', c_void_p * 32), ('rm', c_void_p * 32), ('tr', c_void_p * 32), ('taps', c_int32 * 32), ('a', c_int32 * 32),
                ('b', c_int32 * 32)]


class OptArgs(ctypes.Structure):
    """struct acg_opt_args (acg_opt_step_prepare_bf16)."""
    _fields_ = [('kind', c_int32), ('lr', c_float), ('beta1_or_decay', c_float), ('beta2', c_float), ('eps', c_float),
                ('grad_scale', c_float), ('use_clip', c_int32), ('clip_lo', c_float), ('clip_hi', c_float)]


class CopyList(ctypes.Structure):
    """struct acg_copy_list (ACG_COPY_MAX = 8 segments)."""
    _fields_ = [('src', c_void_p * 8), ('dst', c_void_p * 8), ('rows', c_int64 * 8), ('cols', c_int32 * 8),
                ('dst_pitch', c_int32 * 8), ('dst_dtype', c_int32 * 8), ('src_div', c_int32 * 8), ('src_mod', c_int32 * 8)]


_P = c_void_p
_D = ctypes.POINTER(ConvDesc)
_conv = [_P, _P, _P, _D, c_int32, _P, c_size_t, _P]
_wgrad = [_P, _P, _P, c_float, _D, c_int32, _P, c_size_t, _P]

# name -> (restype, argtypes); mirrors include/acgan_hip.h one to one.
SIGNATURES = {
    'acg_version': (c_int32, []),
    'acg_build_info': (c_char_p, []),
    'acg_last_error': (c_char_p, []),
    'acg_conv_desc_init': (c_int32, [_D] + [c_int32] * 9),
    'acg_conv2d_workspace_bytes': (c_size_t, [_D, c_int32, c_int32]),
    'acg_conv2d_fwd': (c_int32, _conv),
    'acg_conv2d_dgrad': (c_int32, _conv),
    'acg_conv2d_wgrad': (c_int32, _wgrad),
    'acg_conv2d_splits': (c_int32, [_D, c_int32, c_int32]),
    'acg_conv2d_tile': (c_int32, [_D, c_int32, c_int32, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]),
    'acg_conv2d_wgrad_slabs': (c_int32, [_P, _P, _D, c_int32, _P, c_size_t, _P]),
    'acg_deconv2d_wgrad_slabs': (c_int32, [_P, _P, _D, c_int32, _P, c_size_t, _P]),
    'acg_conv2d_stats_blocks': (c_int32, [_D, c_int32, c_int32, c_int32]),
    'acg_conv2d_stats_layout': (c_int32, [_D, c_int32, c_int32, c_int32, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]),
    'acg_conv2d_fwd_stats': (c_int32, [_P, _P, _P, _D, c_int32, _P, c_size_t, _P, c_int32, _P]),
    'acg_deconv2d_fwd_stats': (c_int32, [_P, _P, _P, _D, c_int32, _P, c_size_t, _P, c_int32, _P]),
    'acg_deconv2d_fwd_bias_act_ok': (c_int32, [_D, c_int32]),
    'acg_deconv2d_fwd_bias_act': (c_int32, [_P, _P, _P, _P, _D, c_int32, c_float, c_int32, _P]),
    'acg_conv2d_slab_layouts': (c_int32, [_D, c_int32, c_int32]),
    'acg_conv2d_fwd_slabs': (c_int32, [_P, _P, _D, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_conv2d_dgrad_slabs': (c_int32, [_P, _P, _D, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_deconv2d_fwd_slabs': (c_int32, [_P, _P, _D, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_deconv2d_dgrad_slabs': (c_int32, [_P, _P, _D, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_conv2d_bwd_pair': (c_int32, [_P, _P, _P, _P, _P, c_float, _D, c_int32, _P, c_size_t, _P, c_size_t, c_int32, _P]),
    'acg_deconv2d_bwd_pair': (c_int32, [_P, _P, _P, _P, _P, c_float, _D, c_int32, _P, c_size_t, _P, c_size_t, c_int32, _P]),
    'acg_splitk_reduce_many': (c_int32, [ctypes.POINTER(ReduceList), c_int32, _P]),
    'acg_weights_prepare_bf16': (c_int32, [ctypes.POINTER(PrepList), c_int32, _P]),
    'acg_deconv2d_fwd': (c_int32, _conv),
    'acg_deconv2d_dgrad': (c_int32, _conv),
    'acg_deconv2d_wgrad': (c_int32, _wgrad),
    'acg_bn_workspace_bytes': (c_size_t, [c_int64, c_int32, c_int32]),
    'acg_bn_moments': (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_bn_act_fwd_moments': (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_float,
                                         c_int32, _P]),
    'acg_bn_bwd_sums': (c_int32, [_P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, _P,
                                  c_size_t, _P]),
    'acg_bn_act_bwd_sums': (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, c_float, c_int64, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, c_float, c_int32, _P]),
    'acg_bn_exchange_selftest': (c_int32, [_P, c_size_t, _P, c_int32, c_int32, c_int32, ctypes.c_uint32, _P]),
    'acg_bn_act_fwd': (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_float,
                                 c_int32, c_int32, _P, c_size_t, _P]),
    'acg_bn_act_bwd': (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_float, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_float, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_bn_act_fwd_partials': (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_float,
                                          c_int32, _P]),
    'acg_bn_slabs_layout': (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    'acg_bn_act_fwd_slabs': (c_int32, [_P, c_int32, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_float,
                                       c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_bn_bwd_slabs_ok': (c_int32, [c_int64, c_int32]),
    'acg_bn_act_bwd_slabs': (c_int32, [_P, _P, c_int32, _P, _P, _P, _P, _P, c_float, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                       c_float, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    'acg_bias_workspace_bytes': (c_size_t, [c_int64, c_int32]),
    'acg_bias_act_fwd': (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, _P]),
    'acg_bias_act_bwd': (c_int32, [_P, _P, _P, _P, c_float, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_int32,
                                   _P, c_size_t, _P]),
    'acg_dna_workspace_bytes': (c_size_t, [c_int32] * 4),
    'acg_dna_fwd': (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'acg_dna_bwd': (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_float, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P,
                              c_size_t, _P]),
    'acg_cdna_workspace_bytes': (c_size_t, [c_int32] * 6),
    'acg_cdna_fwd': (c_int32, [_P, _P, _P, _P] + [c_int32] * 6 + [c_float, c_int32, _P]),
    'acg_cdna_bwd': (c_int32, [_P, _P, _P, _P, _P, _P] + [c_int32] * 6 + [c_float, c_int32, _P, c_size_t, _P]),
    'acg_concat_actions_fwd': (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'acg_concat_channels_fwd': (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, _P]),
    'acg_slice_channels': (c_int32, [_P, _P, c_float, c_int64, c_int32, c_int32, c_int32, c_int32, _P]),
    'acg_copy_many': (c_int32, [ctypes.POINTER(CopyList), c_int32, c_int32, _P]),
    'acg_stream_edge_create': (c_int32, [ctypes.POINTER(c_void_p)]),
    'acg_stream_edge_destroy': (c_int32, [_P]),
    'acg_stream_edge': (c_int32, [_P, _P, _P]),
    'acg_add': (c_int32, [_P, _P, _P, c_int64, c_int32, _P]),
    'acg_frame_loss_workspace_bytes': (c_size_t, [c_int64]),
    'acg_frame_loss': (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_int32,
                                 _P, c_size_t, _P]),
    'acg_l2norm_loss': (c_int32, [_P, _P, _P, _P, c_int64, c_float, _P]),
    'acg_sumsq_diff': (c_int32, [_P, _P, _P, c_int64, _P]),
    'acg_l2norm_loss_global': (c_int32, [_P, _P, _P, _P, _P, c_int64, c_float, _P]),
    'acg_sigmoid_ce_loss': (c_int32, [_P, c_float, _P, _P, c_int64, c_float, _P]),
    'acg_mean_loss': (c_int32, [_P, _P, _P, c_int64, c_float, _P]),
    'acg_psnr': (c_int32, [_P, _P, _P, c_int64, c_int32, _P, c_size_t, _P]),
    'acg_scalar_combine': (c_int32, [_P, _P, c_float, _P, c_float, _P, c_float, _P, c_float, _P]),
    'acg_opt_step_prepare_bf16': (c_int32, [_P, _P, _P, _P, _P, c_int64, ctypes.POINTER(OptArgs), ctypes.POINTER(PrepList), c_int32, _P]),
    'acg_adam_step': (c_int32, [_P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float,
                                c_int32, c_float, c_float, _P]),
    'acg_rmsprop_step': (c_int32, [_P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int32, c_float,
                                   c_float, _P]),
    'acg_clip': (c_int32, [_P, c_int64, c_float, c_float, _P]),
    'acg_step_inc': (c_int32, [_P, _P]),
}


COPY_MAX = 8
REDUCE_MAX = 32
PREP_MAX = 32


def dtype2(first, second):
    """ACG_DTYPE2: two storage types in one dtype argument (the plain code when they agree)."""
    return first if first == second else first | (second << 4) | 0x100


def code(torch_dtype):
    """ACG_F32 / ACG_BF16 of a torch dtype."""
    if torch_dtype == torch.float32:
        return ACG_F32
    if torch_dtype == torch.bfloat16:
        return ACG_BF16
    raise TypeError('no storage code for %s' % torch_dtype)
VALUE_RETURNING = ('acg_version', 'acg_conv2d_splits', 'acg_conv2d_tile', 'acg_bn_bwd_slabs_ok', 'acg_bn_slabs_layout', 'acg_conv2d_slab_layouts', 'acg_conv2d_stats_blocks', 'acg_conv2d_stats_layout', 'acg_deconv2d_fwd_bias_act_ok')     # int32 results that are not status codes


class AcgError(RuntimeError):
    pass


class Library:
    """A loaded C-ABI library; every int-returning entry point is checked and raises AcgError."""

    def __init__(self, path, extra=None):
        self.path = path
        self._cdll = ctypes.CDLL(path)
        sigs = dict(SIGNATURES, **(extra or {}))
        missing = [n for n in sigs if not hasattr(self._cdll, n)]
        if missing:
            raise AcgError('%s does not export %s' % (path, ', '.join(missing)))
        for name, (res, args) in sigs.items():
            fn = getattr(self._cdll, name)
            fn.restype, fn.argtypes = res, args
            if res is c_int32 and name not in VALUE_RETURNING:
                fn = self._checked(name, fn)
            setattr(self, name[4:], fn)
        if self.version() != ABI_VERSION:
            raise AcgError('%s: ABI version %d, expected %d' % (path, self.version(), ABI_VERSION))

    def _checked(self, name, fn):
        last_error = self._cdll.acg_last_error

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise AcgError('%s failed (code %d): %s' % (name, rc, last_error().decode()))
        call.__name__ = name
        return call


_LIB = None


def get():
    """The process-wide HIP library; raises if it was not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                '%s not found: the HIP kernels are not built and there is no fallback path. '
                'Run `python -c "import __graft_entry__ as g; g.build()"` first.' % LIB_PATH)
        _LIB = Library(LIB_PATH)
    return _LIB


TUNING_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), 'libacgan_hip_tuning.so')


def load_tuning():
    """tools/ only: the -DACG_TUNING build (`make -C action_conditioned_gans_amd/csrc tuning`) with acg_debug_conv_plan
    and the ACG_* environment knobs, installed as the process's library.  The package itself never loads it."""
    global _LIB
    if not os.path.exists(TUNING_LIB_PATH):
        # not shipped to the GPU box (.gpurunignore) and never built from inside a library loader: a hidden 16-way hipcc
        # build at call time would also run under whatever preload (rocprofv3) the calling tool was started with
        raise RuntimeError('%s not found: build it first, before any profiler or GPU process starts: '
                           '`make -s -j16 -C %s tuning`' % (TUNING_LIB_PATH, os.path.dirname(LIB_PATH)))
    _LIB = Library(TUNING_LIB_PATH, extra={'acg_debug_conv_plan': (c_int32, [c_int32, c_int32])})
    return _LIB
