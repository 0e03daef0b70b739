// Instantiates the paired launches (input gradient + weight gradient of one layer in one grid), see conv_f32_kernel.h.
#include "conv_f32_kernel.h"

namespace acgconv {

bool pair_supported(int modeA, const Plan& pa, const Plan& pb) {
  return (modeA == MODE_FWD || modeA == MODE_DGRAD) && !pa.bf16 && !pb.bf16 && pa.ragged == pb.ragged && pa.nvec && pb.nvec &&
         (long long)pa.tiles * pa.splits + (long long)pb.tiles * pb.splits < (1ll << 30);
}

namespace {
template <int MODE_A, int BMA, int BNA, int WMA, int WNA, bool RAGGED, bool LIN_A>
void launch_b(const Plan& pb, const ConvArgs& a, const ConvArgs& b, const PairGeom& g, unsigned blocks, hipStream_t st) {
  if (pb.cfg == 2) ACG_LAUNCH((conv_pair_f32<MODE_A, BMA, BNA, WMA, WNA, 128, 32, 4, 1, RAGGED, LIN_A>), dim3(blocks), dim3(256), 0, st, a, b, g);
  else ACG_LAUNCH((conv_pair_f32<MODE_A, BMA, BNA, WMA, WNA, 64, 64, 2, 2, RAGGED, LIN_A>), dim3(blocks), dim3(256), 0, st, a, b, g);
}
template <int MODE_A, bool RAGGED, bool LIN_A = false>
void launch_a(const Plan& pa, const Plan& pb, const ConvArgs& a, const ConvArgs& b, const PairGeom& g, unsigned blocks, hipStream_t st) {
  if (pa.cfg == 2) launch_b<MODE_A, 128, 32, 4, 1, RAGGED, LIN_A>(pb, a, b, g, blocks, st);
  else launch_b<MODE_A, 64, 64, 2, 2, RAGGED, LIN_A>(pb, a, b, g, blocks, st);
}
}  // namespace

int launch_pair(int modeA, const Plan& pa, const ConvArgs& a, const Plan& pb, const ConvArgs& b, hipStream_t st) {
  PairGeom g;
  g.gxA = (int)(acg::ceil_div(pa.M, pa.bm) * acg::ceil_div(pa.N, pa.bn));
  g.gyA = pa.classes;
  g.nA = g.gxA * g.gyA * pa.splits;
  g.gxB = (int)(acg::ceil_div(pb.M, pb.bm) * acg::ceil_div(pb.N, pb.bn));
  const unsigned blocks = (unsigned)(g.nA + g.gxB * pb.splits);
  if (modeA == MODE_FWD) {
    // (pair_supported: the dense operands are float4-able; gathered channels a multiple of 4 -> the LIN variant of A)
    if (pa.ragged) launch_a<MODE_FWD, true>(pa, pb, a, b, g, blocks, st);
    else if ((a.C & 3) == 0 && !a.compact) launch_a<MODE_FWD, false, true>(pa, pb, a, b, g, blocks, st);
    else launch_a<MODE_FWD, false>(pa, pb, a, b, g, blocks, st);
  } else {
    if (pa.ragged) launch_a<MODE_DGRAD, true>(pa, pb, a, b, g, blocks, st);
    else launch_a<MODE_DGRAD, false>(pa, pb, a, b, g, blocks, st);
  }
  return acg::check_launch("conv_pair_f32");
}

}  // namespace acgconv
