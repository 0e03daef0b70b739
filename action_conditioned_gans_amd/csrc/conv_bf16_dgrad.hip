// Instantiates the DGRAD contraction of the bf16 MFMA convolution (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace acgconv {
ACG_DEFINE_CONV16_LAUNCH(1)

// the bf16 counterpart of launch_deconv_fwd_epi (conv_f32_dgrad.hip): bf16 operands, dense float32 result
int launch_deconv_fwd_epi16(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, pl.bm) * acg::ceil_div(pl.N, pl.bn)), (unsigned)pl.classes, 1u);
  ACG_LAUNCH((conv_mfma_bf16<MODE_DGRAD, 128, 32, true>), grid, dim3(256), 0, st, a);
  return acg::check_launch("conv_mfma_bf16 (bias + activation epilogue)");
}
}  // namespace acgconv
