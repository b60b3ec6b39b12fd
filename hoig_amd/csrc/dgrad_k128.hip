// The attention MLP's first layer in the backward (split off conv_igemm_bf16.hip in round 6).
#include "conv_bf16_common.h"

namespace hoig_detail {
namespace {
// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of a 1x1 convolution with 128 output channels and a very wide input (the attention MLP's first layer over
// the 25*C sampled channels): dX[m][n] = sum_k dY[m][k] W[n][k] with K = 128 and N = 25*C up to 12800 -- an outer-product
// shaped GEMM whose only real cost is WRITING dX (419 MB per 32x32 layer).  The tile kernels pay a prologue of two operand
// tiles for four k-steps of work per 64-KB output tile and reached 1.7 TB/s.  Here a wave keeps its 32 rows of dY as split
// fragments in REGISTERS for the whole launch (K = 128: 64 registers) and the workgroup walks a range of 64-column steps:
// per step only the weight tile streams (double-buffered LDS image, 32 KB) and 32 KB of dX leave.
struct ThinArgs {
    const float *A;
    const unsigned short *Wh, *Wl;
    float *C;
    int M, N, nsteps, steps_per_wg;
};

template <int NSX>
__global__ __launch_bounds__(256, 2) void dgrad_thin_k128_kernel(const ThinArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int K = 128, PLANE = 8 * 2048, STAGE = NB * PLANE;     // a stage: NB planes x [4 k-blocks][2 n-blocks] x 2 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int s_begin = blockIdx.y * p.steps_per_wg, s_end = min(p.nsteps, s_begin + p.steps_per_wg);
    if (s_begin >= s_end) return;

    bf16x8 ah[8], al[8];
    {
        const float *arow = p.A + (size_t)(m0 + l31) * K + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float4 v0 = *reinterpret_cast<const float4 *>(arow + ks * 16), v1 = *reinterpret_cast<const float4 *>(arow + ks * 16 + 4);
            uint2 h0, l0, h1, l1;
            split4(v0, h0, l0);
            split4(v1, h1, l1);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 hv = {h0.x, h0.y, h1.x, h1.y}, lv = {l0.x, l0.y, l1.x, l1.y};
            ah[ks] = __builtin_bit_cast(bf16x8, hv);
            al[ks] = __builtin_bit_cast(bf16x8, lv);
        }
    }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t rbh[4], rbl[4];
    auto load_b = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, blk = idx >> 7, within = idx & 127;      // blk = kb * 2 + nb
            const size_t src = ((size_t)(step * 2 + (blk & 1)) * (K / 32) + (blk >> 1)) * 1024 + within * 8;
            rbh[i] = *reinterpret_cast<const u32x4_t *>(p.Wh + src);
            if (NB == 2) rbl[i] = *reinterpret_cast<const u32x4_t *>(p.Wl + src);
        }
    };
    auto store_b = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<u32x4_t *>(smem + buf * STAGE + idx * 16) = rbh[i];
            if (NB == 2) *reinterpret_cast<u32x4_t *>(smem + buf * STAGE + PLANE + idx * 16) = rbl[i];
        }
    };
    load_b(s_begin);
    store_b(0);
    load_b(min(s_begin + 1, s_end - 1));
    __syncthreads();
#pragma unroll 1
    for (int s_ = s_begin; s_ < s_end; ++s_) {
        const int buf = (s_ - s_begin) & 1;
        store_b(buf ^ 1);                         // (unconditional: past the last step the tile is simply not used)
        load_b(min(s_ + 2, s_end - 1));
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const unsigned char *Bh = smem + buf * STAGE, *Bl = Bh + PLANE;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = ((ks >> 1) * 2 + j) * 2048 + l31 * 64 + ((((ks & 1) * 2 + lh) ^ ((l31 >> 2) & 3)) << 4);
                const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(Bh + off);
                if (NS == 2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc[j], 0, 0, 0);
                    if (NB == 2) {
                    const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(Bl + off);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc[j], 0, 0, 0);
            }
        }
        const int n0 = s_ * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float *row = p.C + (size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * p.N + n0 + l31;
            row[0] = acc[0][r];
            row[32] = acc[1][r];
        }
        __syncthreads();
    }
}

}  // namespace

int launch_dgrad_thin(const float *dy, const unsigned short *wh, const unsigned short *wl, float *dx, int M, int N, int ns,
                      hipStream_t st) {
    ThinArgs a{dy, wh, wl, dx, M, N, N / 64, 0};
    const int mtiles = M / 128;
    int split = (int)hoig_cdiv(512, mtiles);
    if (split > a.nsteps / 4) split = a.nsteps / 4;
    if (split < 1) split = 1;
    a.steps_per_wg = (int)hoig_cdiv(a.nsteps, split);
    split = (int)hoig_cdiv(a.nsteps, a.steps_per_wg);
    const size_t shm = (size_t)2 * ns_b(ns) * 8 * 2048;
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&dgrad_thin_k128_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&dgrad_thin_k128_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&dgrad_thin_k128_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    HOIG_NS_SWITCH(ns, dgrad_thin_k128_kernel<NSX><<<dim3(mtiles, split), 256, shm, st>>>(a));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace hoig_detail
