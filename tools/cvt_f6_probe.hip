// Hardware probe of the fp16 -> e2m3 (fp6) conversion used for the block-scaled cross terms:
// __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(32 x f16, scale) -> 6 dwords.  Questions: element order in the 192 bits, meaning
// of `scale` (value / scale ?), rounding and saturation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));

__global__ void k(const float *in, float scale, unsigned *out) {
    h32 x;
    for (int j = 0; j < 32; ++j) x[j] = (_Float16)in[threadIdx.x * 32 + j];
    u6 r = __builtin_bit_cast(u6, __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, scale));
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}

static float dec_e2m3(unsigned c) {
    const float s = (c & 32) ? -1.f : 1.f;
    const int e = (c >> 3) & 3, m = c & 7;
    return s * (e == 0 ? m / 8.f : (1.f + m / 8.f) * (float)(1 << (e - 1)));
}

int main() {
    float h[64 * 32];
    // lane 0: element j = (j % 8) with alternating sign; lane 1: multiples of 0.125 and halves; lane 2: values beyond 7.5; lane 3: ties
    for (int j = 0; j < 32; ++j) {
        h[j] = (float)(j % 8) * ((j & 8) ? -1.f : 1.f);
        h[32 + j] = 0.0625f * j;
        h[64 + j] = 6.f + 0.25f * j;
        h[96 + j] = 1.0f + 0.0625f * j;      // ties between 1/8 steps in [1,2): round to nearest even?
    }
    for (int i = 128; i < 64 * 32; ++i) h[i] = 0.f;
    float *din; unsigned *dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 64 * 6 * 4);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (int t = 0; t < 2; ++t) {
        const float scale = t ? 2.f : 1.f;
        k<<<1, 64>>>(din, scale, dout);
        unsigned ho[64 * 6];
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int l = 0; l < 4; ++l) {
            printf(" lane %d in :", l);
            for (int j = 0; j < 32; ++j) printf(" %g", h[l * 32 + j]);
            printf("\n lane %d out:", l);
            for (int j = 0; j < 32; ++j) {
                const int bit = 6 * j;
                unsigned c = ho[l * 6 + (bit >> 5)] >> (bit & 31);
                if ((bit & 31) > 26) c |= ho[l * 6 + (bit >> 5) + 1] << (32 - (bit & 31));
                printf(" %g", dec_e2m3(c & 63));
            }
            printf("\n");
        }
    }
    return 0;
}
