"""GPU parity: every C-ABI entry point of libacgan_hip.so against the torch-fp64 restatement
(oracle/tf_ops.py) on seeded inputs, through the C ABI (tests/abi_call.py).
Tolerance: north_star asks 1e-3 rel in fp32; the per-op bars here are tighter."""
import pytest

import op_cases as C

pytestmark = pytest.mark.gpu

TOL_CONV = 1e-4    # fp32 MFMA fma-chains over K <= 13050 products
TOL_BF16 = 4e-3    # bf16 storage: outputs are rounded to 8 significant bits (2^-9 of the element, <= 2^-8 of the tensor scale)
TOL = 2e-5


@pytest.mark.parametrize('shape', C.CONV_SHAPES, ids=str)
def test_conv(hip_abi, shape):
    C.case_conv(hip_abi, shape, TOL_CONV)


@pytest.fixture(scope='module')
def hip_abi_bf16(hip_abi):
    from abi_call import Abi
    from action_conditioned_gans_amd import _lib
    return Abi(hip_abi.lib, 'cuda:0', conv_dtype=_lib.ACG_BF16)


@pytest.mark.parametrize('shape', C.CONV_SHAPES + [(32, 16, 16, 128, 128, 5, 2, 'SAME'), (32, 32, 32, 64, 128, 5, 2, 'SAME'),
                                   (4, 64, 64, 128, 128, 5, 2, 'SAME')], ids=str)
def test_conv_bf16(hip_abi_bf16, shape):
    """dtype=ACG_BF16 (BASELINE configs 3 and 5): bf16 tensors in memory (pitch round8), bf16 matrix cores, fp32
    accumulation; fp32 weight gradients.  Reference: the fp64 conv of the bf16-rounded operands."""
    C.case_conv_bf16(hip_abi_bf16, shape, TOL_BF16, TOL_CONV)


@pytest.mark.parametrize('shape', C.DECONV_SHAPES + [(8, 32, 32, 128, 25, 5, 2), (32, 16, 16, 128, 128, 5, 2), (4, 64, 64, 128, 121, 5, 2)], ids=str)
def test_deconv_bf16(hip_abi_bf16, shape):
    C.case_conv_bf16(hip_abi_bf16, shape, TOL_BF16, TOL_CONV, transposed=True)


def test_conv_bn_stats(hip_abi):
    """BatchNorm statistics out of the conv epilogue (acg_(de)conv2d_fwd_stats -> acg_bn_act_fwd_partials)."""
    C.case_conv_bn_stats(hip_abi, TOL_CONV, 1e-4, min_fused=4)


def test_conv_adjoint_identities_at_baseline_sizes(hip_abi):
    """<fwd(x,w),dy> = <x,dgrad(dy,w)> = <w,wgrad(x,dy)> for every layer of config 2 at batch 32 (+ config 5's largest two)."""
    C.case_conv_adjoint_identities(hip_abi, 2e-5)


def test_conv_adjoint_identities_at_baseline_sizes_bf16(hip_abi_bf16):
    C.case_conv_adjoint_identities(hip_abi_bf16, 3e-3)


def test_full_size_properties_of_dna_and_batchnorm(hip_abi):
    """Constant-image, linearity and sum-to-zero properties of the DNA stencil; moment and orthogonality properties of BatchNorm -
    at config 2 / 3 / 5's per-GPU tensor sizes."""
    C.case_full_size_properties(hip_abi)


def test_conv_bn_stats_bf16(hip_abi_bf16):
    C.case_conv_bn_stats(hip_abi_bf16, TOL_BF16, 2e-3, min_fused=4)


def test_merged_input_gradient(hip_abi):
    C.case_merged_dgrad(hip_abi, TOL_CONV)


def test_merged_input_gradient_bf16(hip_abi_bf16):
    C.case_merged_dgrad(hip_abi_bf16, TOL_BF16)


def test_dgrad_channel_limit(hip_abi):
    C.case_dgrad_channel_limit(hip_abi, TOL_CONV)


def test_dgrad_channel_limit_bf16(hip_abi_bf16):
    C.case_dgrad_channel_limit(hip_abi_bf16, TOL_BF16)


def test_slab_handoff_layouts(hip_abi):
    C.case_slab_handoff(hip_abi, 2e-5)


def test_slab_handoff_layouts_bf16(hip_abi_bf16):
    C.case_slab_handoff(hip_abi_bf16, 2e-5, min_rows=1)


def test_bn_large_tensor(hip_abi):
    """Config-5-sized BatchNorm (8.4 M float32 elements, 2048+ partial blocks): the finalize launch in front of the apply."""
    C.case_bn_large_tensor(hip_abi, 1e-4)


def test_bn_large_tensor_bf16(hip_abi_bf16):
    C.case_bn_large_tensor(hip_abi_bf16, 2e-3)


def test_conv_bn_stats_large_mean(hip_abi):
    """Tile statistics are centred before they are squared and merged Chan-style (mean / std ~ 1e3)."""
    C.case_conv_bn_stats_large_mean(hip_abi, 2e-3)


def test_conv_bn_stats_large_mean_bf16(hip_abi_bf16):
    C.case_conv_bn_stats_large_mean(hip_abi_bf16, 2e-3)


def test_dna_second(hip_abi):
    """The frame's second home: acg_dna_fwd out2 / acg_dna_bwd dout2, float32 and bf16 tensors, both kernel families."""
    C.case_dna_second(hip_abi, 1e-5)


def test_bwd_pair_bf16(hip_abi_bf16):
    C.case_bwd_pair_bf16(hip_abi_bf16, TOL_BF16, TOL_CONV)


@pytest.mark.parametrize('shape', [(32, 64, 64, 64, 128, 5, 2, 'SAME')], ids=str)
def test_conv_bf16_big_tiles(hip_abi_bf16, shape):
    """Large enough for the planner's 128x128 tiles in all three contractions (256 output tiles; wgrad: long K)."""
    C.case_conv_bf16(hip_abi_bf16, shape, TOL_BF16, TOL_CONV)


def test_deconv_bf16_big_tiles(hip_abi_bf16):
    C.case_conv_bf16(hip_abi_bf16, (16, 32, 32, 128, 128, 5, 2), TOL_BF16, TOL_CONV, transposed=True)


# Shapes whose forward or input gradient the planner hands to the 256 x 128 LDS-DMA kernel (conv_bf16_glds.h: more than 64
# output columns, at least 192 tiles, at least 20 K-steps): full tiles; a partial last row tile with a channel count that makes
# K-steps straddle taps (56 channels: 64 k = 1 tap + 8) and ragged columns (96 of 128); an input gradient (N = 128 input
# channels, 9 taps x 160 channels in its largest stride class) with classes of unequal size (odd extents); 3 x 3 / stride 1
@pytest.mark.parametrize('shape,which', [((48, 64, 64, 64, 128, 5, 2, 'SAME'), 0), ((52, 62, 62, 56, 96, 5, 2, 'SAME'), 0),
                                         ((12, 64, 64, 128, 160, 5, 2, 'SAME'), 1), ((13, 63, 61, 96, 160, 5, 2, 'SAME'), 1),
                                         ((48, 32, 32, 160, 128, 3, 1, 'SAME'), 0)], ids=str)
def test_conv_bf16_wide_tiles(hip_abi_bf16, shape, which):
    b, h, w, cin, cout, k, s, pad = shape
    assert C.tile_rows(hip_abi_bf16, which, b, h, w, cin, k, cout, s, pad) == 256, 'test premise: the planner does not pick the wide kernel'
    C.case_conv_bf16(hip_abi_bf16, shape, TOL_BF16, TOL_CONV)


@pytest.mark.parametrize('shape', [(12, 32, 32, 160, 128, 5, 2), (12, 32, 32, 160, 121, 5, 2)], ids=str)
def test_deconv_bf16_wide_tiles(hip_abi_bf16, shape):
    """A transposed layer's forward as four stride classes of the wide kernel; 121 output channels = g/tconv4 at config 5."""
    b, ih, iw, cin, cout, k, s = shape
    assert C.tile_rows(hip_abi_bf16, 1, b, ih * s, iw * s, cout, k, cin, s, 'SAME') == 256, 'test premise: the planner does not pick the wide kernel'
    C.case_conv_bf16(hip_abi_bf16, shape, TOL_BF16, TOL_CONV, transposed=True)


def test_conv_bn_stats_wide_bf16(hip_abi_bf16):
    """BatchNorm partials out of the 256-row tiles of the wide kernel (one group, two groups, a transposed layer)."""
    C.case_conv_bn_stats(hip_abi_bf16, TOL_BF16, 2e-3, min_fused=3, layers=C.STATS_LAYERS_WIDE)


def test_hand_synchronised_kernels_are_repeatable(hip_abi_bf16):
    """60 launches each of the LDS-DMA convolution and of the one-launch BatchNorm kernels: bit-identical results (race screen)."""
    C.case_repeatable_launches(hip_abi_bf16)


@pytest.mark.parametrize('shape', C.BN_SHAPES[:2] + C.BN_SHAPES[3:7] + [((32, 16, 16), 128, 2, 'lrelu'), ((32, 32, 32), 64, 1, 'relu')], ids=str)
def test_bn_bf16(hip_abi, shape):
    C.case_bn_bf16(hip_abi, shape, 6e-3)


def test_bn_head_bf16(hip_abi):
    C.case_bn_head_bf16(hip_abi, 6e-3)


def test_head_f32_in_bf16_network(hip_abi_bf16):
    """d/conv6 of a bf16 network stays float32: conv result and BatchNorm input float32, gradient back as bf16."""
    C.case_head_f32_in_bf16_network(hip_abi_bf16, TOL_CONV, 6e-3)


def test_deconv_bias_act_epilogue(hip_abi):
    C.case_deconv_bias_act(hip_abi, TOL_CONV)


def test_deconv_bias_act_epilogue_bf16(hip_abi_bf16):
    C.case_deconv_bias_act(hip_abi_bf16, TOL_CONV)          # float32 result of bf16 operands: accumulation level


def test_bias_bf16(hip_abi):
    C.case_bias_bf16(hip_abi, 6e-3)


@pytest.mark.parametrize('shape', C.DNA_SHAPES, ids=str)
def test_dna_bias(hip_abi, shape):
    C.case_dna_bias(hip_abi, shape, TOL)


@pytest.mark.parametrize('shape', C.DNA_SHAPES + [(2, 128, 128, 3, 11)], ids=str)
def test_dna_bf16(hip_abi, shape):
    C.case_dna_bf16(hip_abi, shape, TOL, 6e-3)


def test_plumbing_bf16(hip_abi):
    C.case_plumbing_bf16(hip_abi)


def test_weights_prepare(hip_abi):
    C.case_weights_prepare(hip_abi)


def test_opt_step_prepared(hip_abi):
    """The optimizer update and the refresh of the bf16 filter copies in one launch == the two launches, bit for bit."""
    C.case_opt_step_prepared(hip_abi)


def test_deconv_pitched(hip_abi):
    C.case_deconv_pitched(hip_abi, TOL_CONV)


def test_conv_pitched(hip_abi):
    C.case_conv_pitched(hip_abi, TOL_CONV)


@pytest.mark.parametrize('shape', C.DECONV_SHAPES, ids=str)
def test_deconv(hip_abi, shape):
    C.case_deconv(hip_abi, shape, TOL_CONV)


@pytest.mark.parametrize('shape', [(32, 16, 16, 128, 128, 5, 2, 'SAME'), (32, 4, 4, 256, 512, 5, 2, 'SAME'),
                                   (16, 32, 32, 64, 128, 5, 2, 'SAME')], ids=str)
def test_conv_batch32(hip_abi, shape):
    """BASELINE config-2 batch: exercises the 128x128 tile and split-K paths."""
    C.case_conv(hip_abi, shape, TOL_CONV)


@pytest.mark.parametrize('shape', [(32, 16, 16, 128, 128, 5, 2), (8, 32, 32, 128, 25, 5, 2)], ids=str)
def test_deconv_batch32(hip_abi, shape):
    C.case_deconv(hip_abi, shape, TOL_CONV)


@pytest.mark.parametrize('shape', [(64, 4, 4, 256, 512, 5, 2, 'SAME'),     # d/conv5 at the D step's batch: one output pixel per 64-row tile
                                   (12, 4, 4, 24, 40, 5, 2, 'SAME'),       # two or three pixels per tile, ragged columns
                                   (9, 3, 4, 16, 8, 3, 2, 'SAME'),         # odd map, 3 x 3 filter
                                   (16, 4, 4, 6, 32, 5, 2, 'SAME'),        # 6 gathered channels: the non-LIN forward path either way
                                   (8, 2, 2, 64, 16, 3, 1, 'SAME')], ids=str)
def test_conv_small_maps_walk_live_taps_only(hip_abi, shape):
    """Maps of at most four output pixels (per stride class for the input gradient) at batch >= 8: pixel-major rows and the
    compacted tap list (ConvArgs::compact) - forward and input gradient against the float64 conv, weight gradient alongside."""
    C.case_conv(hip_abi, shape, TOL_CONV)


@pytest.mark.parametrize('shape', [(32, 2, 2, 64, 32, 5, 2), (8, 2, 1, 16, 12, 5, 2)], ids=str)
def test_deconv_small_maps_walk_live_taps_only(hip_abi, shape):
    C.case_deconv(hip_abi, shape, TOL_CONV)


@pytest.mark.parametrize('shape', C.BN_SHAPES + [((32, 32, 32), 128, 1, 'relu'), ((64, 4, 4), 512, 2, 'lrelu'),
                                            # register-resident kernels (<= 4096 rows per group): exact fit, one row past it, ragged rows, 3 groups
                                            ((4, 32, 32), 16, 1, 'relu'), ((1, 17, 241), 8, 1, 'lrelu'), ((3, 9, 19), 12, 3, None),
                                            ((2, 45, 45), 4, 2, 'relu'), ((2, 30, 30), 3, 1, 'lrelu')], ids=str)
def test_bn(hip_abi, shape):
    C.case_bn(hip_abi, shape, TOL)


@pytest.mark.parametrize('shape', [((32, 64, 64), 16, 1, None), ((16, 64, 64), 8, 1, None), ((32, 64, 64), 4, 2, None), ((8, 128, 128), 28, 1, None)], ids=str)
def test_bn_few_channels_many_rows_stay_inside_the_workspace(hip_abi, shape):
    """a chunk of <= 32 channels costs the one-launch kernels 512 exchange bytes per row block WHATEVER the channel
    count, and round 4 sized the workspace by the channels alone: 131072 x 16 wrote 131 KB of granules into a 66 KB workspace.
    Now acg_bn_workspace_bytes covers the exchange area of every admissible grid (<= 512 blocks); every workspace the test
    helpers hand out carries a canary behind it (abi_call.Abi.ws) that no_timeout checks.  (No activation: with millions of
    elements one pre-activation lands within float32 rounding of the ReLU kink, where the float64 reference and the kernel
    legitimately disagree on the derivative of that one element - seen as a single-element dx error of |dy| * rstd.)"""
    C.case_bn(hip_abi, shape, TOL)


def test_bn_without_grid_exchange_flag(hip_abi):
    """ACG_BN_NO_GRID_EXCHANGE (a caller whose BatchNorm launches can overlap: Session(side_branches=True)): the same results
    from the register-resident / two-launch kernels - the epoch word of the workspace, which only the grid kernels count in,
    stays zero - and acg_bn_slabs_layout no longer offers the rows layout that only the grid kernels read."""
    import torch
    from action_conditioned_gans_amd import _lib as L
    rows, c = 32 * 32 * 32, 128
    assert hip_abi.lib.bn_slabs_layout(rows, c, c, c, 1, L.ACG_F32, 0, 0) == L.SLABS_ROWS
    assert hip_abi.lib.bn_slabs_layout(rows, c, c, c, 1, L.ACG_F32, 0, L.BN_NO_GRID_EXCHANGE) == L.SLABS_ROWS     # forward: the two-launch kernels sum rows slabs too
    assert hip_abi.lib.bn_slabs_layout(rows, c, c, c, 1, L.ACG_F32, 1, L.BN_NO_GRID_EXCHANGE) == -1               # backward: nobody does
    hip_abi.bn_flags = L.BN_NO_GRID_EXCHANGE
    try:
        for shape in [((32, 32, 32), 128, 1, 'relu'), ((64, 16, 16), 128, 2, 'lrelu'), ((32, 8, 8), 128, 1, 'relu')]:
            C.case_bn(hip_abi, shape, TOL)
        x = torch.randn(rows, c, device='cuda')
        ws, n = hip_abi.bn_ws(rows, c, 1)
        y, mean, rstd = torch.empty_like(x), hip_abi.empty(c), hip_abi.empty(c)
        hip_abi.lib.bn_act_fwd(C._ptr(x), C._ptr(torch.zeros(c, device='cuda')), C._ptr(y), C._ptr(mean), C._ptr(rstd), rows, c, 0, 0, 1, 1e-3, L.ACT_RELU, 0.2,
                               L.ACG_F32, L.BN_NO_GRID_EXCHANGE, C._ptr(ws), n, hip_abi.stream())
        torch.cuda.synchronize()
        assert int(ws[0:4].view(torch.int32)[0]) == 0, 'a grid-exchange kernel ran although the flag forbids it'
    finally:
        hip_abi.bn_flags = 0


def test_bn_exchange_timeout_is_flagged_and_raised(hip_abi):
    """the FAILURE path of the in-launch exchange, provoked once.  acg_bn_exchange_selftest runs the very
    FusedExchange code of bn_fwd_fused / bn_bwd_fused: healthy, every block reads n (n + 1) / 2 - also when the same workspace is
    reused (the epoch advances); with one block withheld every block runs into the (lowered) spin bound, word 2 of the workspace
    is set, the sums come out short, and graph.Runtime.check_exchange_flags - what train() and bench.py call - raises AcgError."""
    import torch
    from action_conditioned_gans_amd import _lib as L, graph as G
    lib = hip_abi.lib
    rt = G.Runtime(lib, 'cuda:0')
    for blocks, threads in ((256, 1024), (512, 256), (7, 256)):
        ws, n = rt.state_workspace(16 + 512 * blocks)
        out = torch.zeros(blocks, device='cuda')
        for rep in range(3):
            lib.bn_exchange_selftest(C._ptr(ws), n, C._ptr(out), blocks, threads, -1, 1 << 22, hip_abi.stream())
            torch.cuda.synchronize()
            assert torch.equal(out.cpu(), torch.full((blocks,), blocks * (blocks + 1) / 2.0)), (blocks, threads, rep, out[:4].tolist())
            assert int(ws[0:4].view(torch.int32)[0]) == rep + 1 and int(ws[8:12].view(torch.int32)[0]) == 0
    rt.check_exchange_flags()                       # all clear so far
    blocks, threads = 64, 1024
    ws, n = rt.state_workspace(16 + 512 * blocks)
    out = torch.zeros(blocks, device='cuda')
    lib.bn_exchange_selftest(C._ptr(ws), n, C._ptr(out), blocks, threads, 5, 2000, hip_abi.stream())      # block 5 publishes nothing
    torch.cuda.synchronize()                        # (bounded: 2000 polls, not a hang)
    assert int(ws[8:12].view(torch.int32)[0]) == 1, 'the timeout word was not set'
    want = blocks * (blocks + 1) / 2.0 - 6.0
    assert all(v == want or v == -1.0 for v in out.cpu().tolist()), out[:8].tolist()      # short sums: a wrong result, as documented
    with pytest.raises(L.AcgError, match='grid exchange timed out in 1 of 4 call sites'):
        rt.check_exchange_flags()
    with pytest.raises(L.AcgError):                 # argument checks of the entry
        lib.bn_exchange_selftest(C._ptr(ws), n, C._ptr(out), 513, 256, -1, 1, hip_abi.stream())


def test_bn_large_mean(hip_abi):
    C.case_bn_large_mean(hip_abi, 2e-2)


def test_bias(hip_abi):
    C.case_bias(hip_abi, TOL)


@pytest.mark.parametrize('shape', C.DNA_SHAPES + [(32, 64, 64, 3, 5), (2, 128, 128, 3, 11)], ids=str)
def test_dna(hip_abi, shape):
    C.case_dna(hip_abi, shape, TOL)


@pytest.mark.parametrize('shape', [(2, 9, 7, 3, 4, 5), (1, 16, 16, 1, 1, 3), (2, 20, 17, 3, 10, 5), (1, 8, 8, 4, 3, 7),
                                   (4, 64, 64, 3, 10, 5)])
def test_cdna(hip_abi, shape):
    C.case_cdna(hip_abi, shape, TOL)


@pytest.mark.parametrize('shape,act,groups', [((4, 6, 5, 8), 'relu', 1), ((6, 3, 3, 12), 'lrelu', 2), ((2, 5, 7, 3), None, 1),
                                              ((32, 16, 16, 128), 'relu', 1), ((64, 8, 8, 256), 'lrelu', 2)])
def test_sync_bn_entries(hip_abi, shape, act, groups):
    C.case_sync_bn_entries(hip_abi, shape, act, groups, TOL)


def test_sync_bn_entries_bf16(hip_abi):
    C.case_sync_bn_entries_bf16(hip_abi, TOL)


def test_bwd_pair(hip_abi):
    C.case_bwd_pair(hip_abi, TOL_CONV, exact=True)


def test_wgrad_deferred_reduction(hip_abi):
    C.case_wgrad_deferred(hip_abi, TOL_CONV, exact=True)


def test_copy_many(hip_abi):
    C.case_copy_many(hip_abi)


def test_dna_extreme(hip_abi):
    C.case_dna_extreme_logits(hip_abi, TOL)


def test_plumbing(hip_abi):
    C.case_plumbing(hip_abi, TOL)


def test_losses(hip_abi):
    C.case_losses(hip_abi, TOL)


def test_optimizers(hip_abi):
    C.case_optimizers(hip_abi, TOL)


def test_error_reporting(hip_abi):
    """Bad arguments come back as a non-zero code + message, never a launch."""
    import torch
    from action_conditioned_gans_amd._lib import AcgError
    x = torch.zeros(2, 4, 4, 3, device='cuda')
    with pytest.raises(AcgError):
        hip_abi.dna_fwd(torch.zeros(2, 4, 4, 144, device='cuda'), x, 12)        # ksize > 11
    with pytest.raises(AcgError):
        hip_abi.bn_act_fwd(torch.zeros(3, 4, 4, 8, device='cuda'), torch.zeros(8, device='cuda'), 'relu', groups=5)
