// bf16-operand MFMA variants of the implicit-GEMM convolution (HOIG_PREC_BF16X3 / HOIG_PREC_BF16).
// Until a shape is covered here the dispatcher falls back to the exact-fp32 MFMA kernels of conv_igemm.hip.
#include "common.h"

int hoig_conv_bf16_fwd_like(const hoig_conv_desc *, const float *, const float *, const float *, float *, bool,
                            hipStream_t) {
    return HOIG_EUNSUPPORTED;
}
int hoig_conv_bf16_wgrad(const hoig_conv_desc *, const float *, const float *, float *, hipStream_t) {
    return HOIG_EUNSUPPORTED;
}
