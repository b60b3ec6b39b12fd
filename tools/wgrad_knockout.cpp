// Times the weight gradient of the step's dominant layer (16 x 32x32 x 512 -> 512, 3x3: wgrad_halo_bf16_kernel<NSX, 3, 2>) with parts
// of the kernel compiled out (HOIG_WG_KO bits, hoig_amd/csrc/wgrad_igemm_bf16.hip); second argument 0 = all-zero operands (the same
// instruction stream without data toggling).  Built per variant by tools/wgrad_knockout.sh.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hoig_kernels.h"

int hoig_conv_bf16_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, float *dbias, hipStream_t st);

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, H = 32, W = 32;
    const int C = argc > 4 ? atoi(argv[4]) : 512, N = argc > 5 ? atoi(argv[5]) : C;      // (channel counts other than 512: strides that
                                                                                        // are not a multiple of 2 KB)
    const bool zero = argc > 2 && atoi(argv[2]) == 0;
    const int prec = argc > 3 ? atoi(argv[3]) : HOIG_PREC_F16X2;
    const size_t nx = (size_t)B * H * W * C, ny = (size_t)B * H * W * N, nw = (size_t)N * 9 * C;
    std::vector<float> hx(nx), hy(ny);
    srand(3);
    for (auto &v : hx) v = zero ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto &v : hy) v = zero ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *x, *dy, *dw;
    hipMalloc(&x, nx * 4); hipMalloc(&dy, ny * 4); hipMalloc(&dw, nw * 4);
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, hy.data(), ny * 4, hipMemcpyHostToDevice);
    hipMemset(dw, 0, nw * 4);
    hoig_conv_desc d = {};
    d.B = B; d.Hi = H; d.Wi = W; d.Ci = C; d.Ho = H; d.Wo = W; d.Co = N; d.R = 3; d.S = 3; d.stride = 1; d.pad = 1;
    d.act = HOIG_ACT_NONE; d.precision = prec;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        if (hoig_conv_bf16_wgrad(&d, x, dy, dw, nullptr, nullptr) != HOIG_OK) { printf("launch failed\n"); return 2; }
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) hoig_conv_bf16_wgrad(&d, x, dy, dw, nullptr, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, flop = 2.0 * B * H * W * N * 9.0 * C;
    printf("WG_KO=%d%s prec=%d  B=%d C=%d N=%d  %.1f us  %.1f TFLOP/s\n", HOIG_WG_KO_VALUE, zero ? " zero" : "", prec, B, C, N, us, flop / us * 1e-6);
    return 0;
}
