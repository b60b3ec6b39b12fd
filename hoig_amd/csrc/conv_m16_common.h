// Shared by the kernels on v_mfma_f32_16x16x32 (conv_halo16.hip, conv_s2_16.hip): the MFMA wrapper, the LDS image rounding and the
// instance-norm statistics epilogue.
#pragma once
#include "conv_bf16_common.h"

namespace hoig_detail {
namespace {

template <bool F16>
__device__ __forceinline__ f32x4 mfma_m16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int round128(int v) { return (v + 127) / 128 * 128; }

// Instance-norm statistics of what a workgroup stored (halo_stats_epilogue of the 32x32 kernels): the 16 lanes of a channel quad
// hold different pixels (four xor steps), the WM waves with the same wn different rows (LDS, dead by now), then ONE atomic per
// (workgroup, channel, moment) into the accumulators of the image.
template <int NTW, int WM, int BN, int NT>
__device__ __forceinline__ void m16_stats_epilogue(float (&st1)[NTW][4], float (&st2)[NTW][4], unsigned char *lds, float *stats_img, int N,
                                                   int n0, int wm, int wn, int lane, int tid) {
    const int l15 = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                st1[j][q] += __shfl_xor(st1[j][q], o);
                st2[j][q] += __shfl_xor(st2[j][q], o);
            }
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);      // [WM][2][BN]
    if (l15 == 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = wn * (NTW * 16) + j * 16 + lg * 4 + q;
                red[(wm * 2 + 0) * BN + cl] = st1[j][q];
                red[(wm * 2 + 1) * BN + cl] = st2[j][q];
            }
    }
    __syncthreads();
    for (int e = tid; e < 2 * BN; e += NT) {
        const int mom = e / BN, cl = e - mom * BN;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < WM; ++k) v += red[(k * 2 + mom) * BN + cl];
        if (n0 + cl < N) atomicAdd(&stats_img[(size_t)mom * N + n0 + cl], v);
    }
}

}  // namespace
}  // namespace hoig_detail
