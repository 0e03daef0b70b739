// Instantiates the FWD contraction of the fp32 MFMA convolution (see conv_f32.hip).
#include "conv_f32_kernel.h"

namespace acgconv {
ACG_DEFINE_CONV_LAUNCH(0)
}  // namespace acgconv
