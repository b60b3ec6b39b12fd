// Times hoig_conv2d_fwd_f6 on the step's dominant shape (16 x 32x32 x 512 -> 512) with parts of the kernel compiled out
// (HOIG_F6_KO bits, hoig_amd/csrc/conv_f6.hip): where the time of the launch goes.  Second argument 0 = all-zero operands: the
// same instruction stream without data toggling, i.e. at the clock the chip holds when the MFMA datapath draws little power.  Built per variant by tools/f6_knockout.sh:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHOIG_F6_KO=<bits> -Iinclude -Ihoig_amd/csrc tools/f6_knockout.cpp
//         hoig_amd/csrc/conv_f6.hip -o tools/_build/f6_ko_<bits>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hoig_kernels.h"

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, H = 32, W = 32, C = 512, N = 512;
    const bool zero = argc > 2 && atoi(argv[2]) == 0;      // all-zero operands: the same instruction stream without data toggling
    const size_t nx = (size_t)B * H * W * C, ny = (size_t)B * H * W * N, nw = (size_t)N * 9 * C;
    std::vector<float> hx(nx), hw(nw);
    srand(3);
    for (auto &v : hx) v = zero ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto &v : hw) v = zero ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    std::vector<uint16_t> hwh(nw);
    for (size_t i = 0; i < nw; ++i) hwh[i] = zero ? 0 : (uint16_t)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15));   // fp16 in [0.125, 0.25)
    float *x, *w, *y;
    uint16_t *wh;
    uint8_t *qh, *ql;
    const int64_t qb = hoig_f6_plane_bytes(N, 9, C);
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4); hipMalloc(&wh, nw * 2); hipMalloc(&qh, qb); hipMalloc(&ql, qb);
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(wh, hwh.data(), nw * 2, hipMemcpyHostToDevice);
    if (hoig_pack_conv_weight_f6(w, N, 9, C, qh, ql, nullptr) != HOIG_OK) return 1;
    hoig_conv_desc d = {};
    d.B = B; d.Hi = H; d.Wi = W; d.Ci = C; d.Ho = H; d.Wo = W; d.Co = N; d.R = 3; d.S = 3; d.stride = 1; d.pad = 1;
    d.act = HOIG_ACT_NONE; d.precision = HOIG_PREC_F16F6;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        if (hoig_conv2d_fwd_f6(&d, x, wh, qh, ql, nullptr, y, nullptr) != HOIG_OK) { printf("launch failed\n"); return 2; }
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) hoig_conv2d_fwd_f6(&d, x, wh, qh, ql, nullptr, y, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, flop = 2.0 * B * H * W * N * 9.0 * C;
    printf("KO=%d%s  B=%d  %.1f us  %.1f TFLOP/s\n", HOIG_F6_KO_VALUE, zero ? " zero" : "", B, us, flop / us * 1e-6);
    return 0;
}
