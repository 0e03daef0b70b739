// Instantiates the paired bf16 launches (input gradient + weight gradient of one layer in one grid), see conv_bf16_kernel.h.
#include "conv_bf16_kernel.h"

namespace acgconv {

namespace {
template <int MODE_A, int BMA, int BNA>
void launch_b(const Plan& pb, const ConvArgs& a, const ConvArgs& b, const PairGeom& g, unsigned blocks, hipStream_t st) {
  if (pb.bm == 128) ACG_LAUNCH((conv_pair_bf16<MODE_A, BMA, BNA, 128, 128>), dim3(blocks), dim3(256), 0, st, a, b, g);
  else ACG_LAUNCH((conv_pair_bf16<MODE_A, BMA, BNA, 64, 64>), dim3(blocks), dim3(256), 0, st, a, b, g);
}
template <int MODE_A>
void launch_a(const Plan& pa, const Plan& pb, const ConvArgs& a, const ConvArgs& b, const PairGeom& g, unsigned blocks, hipStream_t st) {
  if (pa.bn == 32) launch_b<MODE_A, 128, 32>(pb, a, b, g, blocks, st);          // narrow input gradient (g/conv2: 32 channels)
  else if (pa.bm == 128) launch_b<MODE_A, 128, 128>(pb, a, b, g, blocks, st);
  else launch_b<MODE_A, 64, 64>(pb, a, b, g, blocks, st);
}
}  // namespace

int launch_pair16(int modeA, const Plan& pa, const ConvArgs& a, const Plan& pb, const ConvArgs& b, hipStream_t st) {
  PairGeom g;
  g.gxA = (int)(acg::ceil_div(pa.M, pa.bm) * acg::ceil_div(pa.N, pa.bn));
  g.gyA = pa.classes;
  g.nA = g.gxA * g.gyA * pa.splits;
  g.gxB = (int)(acg::ceil_div(pb.M, pb.bm) * acg::ceil_div(pb.N, pb.bn));
  const unsigned blocks = (unsigned)(g.nA + g.gxB * pb.splits);
  if (modeA == MODE_FWD) launch_a<MODE_FWD>(pa, pb, a, b, g, blocks, st);
  else launch_a<MODE_DGRAD>(pa, pb, a, b, g, blocks, st);
  return acg::check_launch("conv_pair_bf16");
}

}  // namespace acgconv
