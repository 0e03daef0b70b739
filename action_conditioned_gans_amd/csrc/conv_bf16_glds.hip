// Instantiates the LDS-DMA staged bf16 convolution (FWD / DGRAD, 256 x 128 tile; see conv_bf16_glds.h).
#include "conv_bf16_glds.h"

#include <stdlib.h>

namespace acgconv {

#ifdef ACG_TUNING
static int glds_env(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
#endif

int launch_glds16(int mode, const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, GBM) * acg::ceil_div(pl.N, GBN)), (unsigned)pl.classes, 1u);
  // K-loop variants (conv_bf16_glds.h): 2 = two wave groups one phase apart, DMA pieces issued between the MFMAs (shipped);
  // 0 = all waves in step, 1 = two groups, DMA in the load phase - tuning builds only (profiles/r4/h_bf16_kloop_load_path.txt)
#ifdef ACG_TUNING
  static const int var = glds_env("ACG_GLDS_VAR", 2);
  if (var == 0) {
    if (mode == MODE_FWD) ACG_LAUNCH((conv_glds_bf16<MODE_FWD, 0>), grid, dim3(GNT), 0, st, a);
    else ACG_LAUNCH((conv_glds_bf16<MODE_DGRAD, 0>), grid, dim3(GNT), 0, st, a);
    return acg::check_launch("conv_glds_bf16");
  }
  if (var == 1) {
    if (mode == MODE_FWD) ACG_LAUNCH((conv_glds_bf16<MODE_FWD, 1>), grid, dim3(GNT), 0, st, a);
    else ACG_LAUNCH((conv_glds_bf16<MODE_DGRAD, 1>), grid, dim3(GNT), 0, st, a);
    return acg::check_launch("conv_glds_bf16");
  }
#endif
  if (mode == MODE_FWD) ACG_LAUNCH((conv_glds_bf16<MODE_FWD, 2>), grid, dim3(GNT), 0, st, a);
  else ACG_LAUNCH((conv_glds_bf16<MODE_DGRAD, 2>), grid, dim3(GNT), 0, st, a);
  return acg::check_launch("conv_glds_bf16");
}

}  // namespace acgconv
