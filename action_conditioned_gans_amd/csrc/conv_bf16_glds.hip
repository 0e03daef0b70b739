// Instantiates the LDS-DMA staged bf16 convolution (FWD / DGRAD, 256 x 128 tile; see conv_bf16_glds.h).
#include "conv_bf16_glds.h"

namespace acgconv {

int launch_glds16(int mode, const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, GBM) * acg::ceil_div(pl.N, GBN)), (unsigned)pl.classes, 1u);
  if (mode == MODE_FWD) ACG_LAUNCH((conv_glds_bf16<MODE_FWD>), grid, dim3(GNT), 0, st, a);
  else ACG_LAUNCH((conv_glds_bf16<MODE_DGRAD>), grid, dim3(GNT), 0, st, a);
  return acg::check_launch("conv_glds_bf16");
}

}  // namespace acgconv
