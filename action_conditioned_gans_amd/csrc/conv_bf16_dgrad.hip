// Instantiates the DGRAD contraction of the bf16 MFMA convolution (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

namespace acgconv {
ACG_DEFINE_CONV16_LAUNCH(1)
}  // namespace acgconv
