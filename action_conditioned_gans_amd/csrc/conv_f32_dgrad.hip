// Instantiates the DGRAD contraction of the fp32 MFMA convolution (see conv_f32.hip).
#include "conv_f32_kernel.h"

namespace acgconv {
ACG_DEFINE_CONV_LAUNCH(1)

// A transposed layer's forward with bias + activation in the epilogue (acg_deconv2d_fwd_bias_act): the 128x32 tile,
// 16-byte gathers - the plain generator's last layer, tanh(deconv(x) + b) with 3 output channels (models.py:20-21).
int launch_deconv_fwd_epi(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, pl.bm) * acg::ceil_div(pl.N, pl.bn)), (unsigned)pl.classes, 1u);
  ACG_LAUNCH((conv_mfma_f32<MODE_DGRAD, 128, 32, 4, 1, false, true, true>), grid, dim3(256), 0, st, a);
  return acg::check_launch("conv_mfma_f32 (bias + activation epilogue)");
}
}  // namespace acgconv
