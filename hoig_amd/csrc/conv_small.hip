// Direct (non-MFMA) kernels for the image / mask heads of the generator: 7x7 stride-1 convolutions with 1 or 3 output
// channels over 64 or 128 input channels at full resolution (generator.py:124,219-235: img_reg, attetion_reg_hand,
// attetion_reg_bg, bg_model.model.27).  On the MFMA implicit GEMM their N = Co <= 3 is padded to a 32-wide tile, i.e.
// >10x wasted matrix work and 17 % of the step; here they run on the fp32 VALU with LDS-tiled inputs:
//   forward : one 16x16 output tile per workgroup, input halo tile staged per 16-channel chunk (pixel stride 20 floats:
//             conflict-free ds_read_b128), weights through the scalar path, bias + tanh/sigmoid fused.
//   wgrad   : dW[n][r][s][c] = sum_pix dy[pix][n] * x[pix + (r,s) - pad][c]; a thread owns one (c, r) pair and slides a
//             register window over s, so one LDS read feeds 3*S FMAs; partial sums of 16 tiles per atomic.
// Exact fp32 arithmetic (k-ordered fmaf chains) in every precision mode.
#include "common.h"

namespace {

constexpr int TILE = 16, CC = 16, PSTR = 20;   // output tile side, channel chunk, LDS pixel stride (floats)
constexpr int KMAX = 7, KS = 7;                 // the heads are all 7x7 (R = S = KS)

template <int CO>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, float *__restrict__ y,
                                                             int B, int H, int W, int Ci, int R, int S, int pad, int act,
                                                             float slope, int CoReal) {
    // x: [B][H][W][Ci], w: [CoReal][R][S][Ci], y: [B][H][W][CoReal]; stride 1, "same" size output
    __shared__ __attribute__((aligned(16))) float xs[(TILE + KMAX - 1) * (TILE + KMAX - 1) * PSTR];
    const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
    int t = blockIdx.x;
    const int bx = t % tiles_x;
    t /= tiles_x;
    const int by = t % tiles_y, b = t / tiles_y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int TW = TILE + S - 1, TH = TILE + R - 1;
    const int y0 = by * TILE - pad, x0 = bx * TILE - pad;
    float acc[CO];
#pragma unroll
    for (int n = 0; n < CO; ++n) acc[n] = 0.f;
    for (int c0 = 0; c0 < Ci; c0 += CC) {
        __syncthreads();
        // stage the halo tile: TH*TW pixels x 16 channels, one float4 per thread-iteration
        for (int i = threadIdx.x; i < TH * TW * (CC / 4); i += 256) {
            const int c4 = i & 3, pix = i >> 2;
            const int py = pix / TW, px = pix - py * TW;
            const int gy = y0 + py, gx = x0 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4 *>(x + (((size_t)b * H + gy) * W + gx) * Ci + c0 + c4 * 4);
            *reinterpret_cast<float4 *>(&xs[pix * PSTR + c4 * 4]) = v;
        }
        __syncthreads();
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) {
                const float *px_ = &xs[((ty + r) * TW + tx + s) * PSTR];   // R = S = KS
                const float *wp = w + ((size_t)r * S + s) * Ci + c0;      // + n*R*S*Ci ; wave-uniform -> scalar loads
#pragma unroll
                for (int c4 = 0; c4 < CC / 4; ++c4) {
                    const float4 v = *reinterpret_cast<const float4 *>(px_ + c4 * 4);
#pragma unroll
                    for (int n = 0; n < CO; ++n) {
                        const float4 ww = *reinterpret_cast<const float4 *>(wp + (size_t)n * R * S * Ci + c4 * 4);
                        acc[n] = fmaf(v.x, ww.x, acc[n]);
                        acc[n] = fmaf(v.y, ww.y, acc[n]);
                        acc[n] = fmaf(v.z, ww.z, acc[n]);
                        acc[n] = fmaf(v.w, ww.w, acc[n]);
                    }
                }
            }
    }
    const int oy = by * TILE + ty, ox = bx * TILE + tx;
    if (oy < H && ox < W) {
        float *o = y + (((size_t)b * H + oy) * W + ox) * CoReal;
#pragma unroll
        for (int n = 0; n < CO; ++n) o[n] = hoig_act(acc[n] + (bias ? bias[n] : 0.f), act, slope);
    }
}

// grid: (tile groups, Ci/16).  Threads: c = tid & 15, r = (tid >> 4) % R, half = (tid >> 4) / R  (needs 2*16*R <= 256)
template <int CO>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                               float *__restrict__ dw, int B, int H, int W, int Ci, int R,
                                                               int S, int pad, int CoReal, int tiles_per_block) {
    __shared__ __attribute__((aligned(16))) float xs[(TILE + KMAX - 1) * (TILE + KMAX - 1) * PSTR];
    __shared__ __attribute__((aligned(16))) float ds[TILE * TILE * 4];
    const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
    const int ntiles = B * tiles_y * tiles_x;
    const int c0 = blockIdx.y * CC;
    const int c = threadIdx.x & 15, rr = (threadIdx.x >> 4) % R, half = (threadIdx.x >> 4) / R;
    const bool worker = half < 2;
    const int TW = TILE + S - 1, TH = TILE + R - 1;
    float acc[KMAX][CO];
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[s][n] = 0.f;

    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(ntiles, t_begin + tiles_per_block);
    for (int t = t_begin; t < t_end; ++t) {
        int q = t;
        const int bx = q % tiles_x;
        q /= tiles_x;
        const int by = q % tiles_y, b = q / tiles_y;
        const int y0 = by * TILE - pad, x0 = bx * TILE - pad;
        __syncthreads();
        for (int i = threadIdx.x; i < TH * TW * (CC / 4); i += 256) {
            const int c4 = i & 3, pix = i >> 2;
            const int py = pix / TW, px = pix - py * TW;
            const int gy = y0 + py, gx = x0 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4 *>(x + (((size_t)b * H + gy) * W + gx) * Ci + c0 + c4 * 4);
            *reinterpret_cast<float4 *>(&xs[pix * PSTR + c4 * 4]) = v;
        }
        {
            const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
            const int gy = by * TILE + py, gx = bx * TILE + px;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W) {
                const float *s_ = dy + (((size_t)b * H + gy) * W + gx) * CoReal;
                g.x = s_[0];
                if (CO > 1) g.y = s_[1];
                if (CO > 2) g.z = s_[2];
                if (CO > 3) g.w = s_[3];
            }
            *reinterpret_cast<float4 *>(&ds[threadIdx.x * 4]) = g;
        }
        __syncthreads();
        if (worker) {
            for (int py = half * (TILE / 2); py < (half + 1) * (TILE / 2); ++py) {
                const float *xrow = &xs[((py + rr) * TW) * PSTR + c];
                float win[KS];                        // x[py+r][px + s][c], s = 0..KS-1 (sliding register window)
#pragma unroll
                for (int s = 0; s < KS - 1; ++s) win[s + 1] = xrow[s * PSTR];
#pragma unroll 4
                for (int px = 0; px < TILE; ++px) {
#pragma unroll
                    for (int s = 0; s < KS - 1; ++s) win[s] = win[s + 1];
                    win[KS - 1] = xrow[(px + KS - 1) * PSTR];
                    const float4 g = *reinterpret_cast<const float4 *>(&ds[(py * TILE + px) * 4]);
                    const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int n = 0; n < CO; ++n) acc[s][n] = fmaf(win[s], gv[n], acc[s][n]);
                }
            }
        }
    }
    if (worker) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int n = 0; n < CO; ++n) atomicAdd(&dw[(((size_t)n * R + rr) * S + s) * Ci + c0 + c], acc[s][n]);
    }
}

}  // namespace

// returns HOIG_EUNSUPPORTED when the problem is not a small-Co stride-1 "same" convolution
int hoig_conv_small_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                        hipStream_t st) {
    if (d->transposed || d->stride != 1 || d->Co > 4 || (d->Ci % CC) || d->R != KS || d->S != KS) return HOIG_EUNSUPPORTED;
    if (d->Ho != d->Hi || d->Wo != d->Wi || 2 * d->pad != d->R - 1 || d->R != d->S) return HOIG_EUNSUPPORTED;
    const int tiles = d->B * (int)hoig_cdiv(d->Hi, TILE) * (int)hoig_cdiv(d->Wi, TILE);
#define HOIG_SMALL_FWD(N)                                                                                              \
    conv_small_fwd_kernel<N><<<tiles, 256, 0, st>>>(x, w, bias, y, d->B, d->Hi, d->Wi, d->Ci, d->R, d->S, d->pad, d->act, \
                                                    d->slope, d->Co)
    switch (d->Co) {
        case 1: HOIG_SMALL_FWD(1); break;
        case 2: HOIG_SMALL_FWD(2); break;
        case 3: HOIG_SMALL_FWD(3); break;
        default: HOIG_SMALL_FWD(4); break;
    }
#undef HOIG_SMALL_FWD
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

int hoig_conv_small_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, hipStream_t st) {
    if (d->transposed || d->stride != 1 || d->Co > 4 || (d->Ci % CC) || d->R != KS || d->S != KS) return HOIG_EUNSUPPORTED;
    if (d->Ho != d->Hi || d->Wo != d->Wi || 2 * d->pad != d->R - 1 || d->R != d->S) return HOIG_EUNSUPPORTED;
    if (2 * 16 * d->R > 256) return HOIG_EUNSUPPORTED;
    const int tiles = d->B * (int)hoig_cdiv(d->Hi, TILE) * (int)hoig_cdiv(d->Wi, TILE);
    int tpb = 16;
    while (tpb > 1 && hoig_cdiv(tiles, tpb) * (d->Ci / CC) < 512) tpb >>= 1;
    dim3 grid((unsigned)hoig_cdiv(tiles, tpb), d->Ci / CC);
#define HOIG_SMALL_WG(N) \
    conv_small_wgrad_kernel<N><<<grid, 256, 0, st>>>(x, dy, dw, d->B, d->Hi, d->Wi, d->Ci, d->R, d->S, d->pad, d->Co, tpb)
    switch (d->Co) {
        case 1: HOIG_SMALL_WG(1); break;
        case 2: HOIG_SMALL_WG(2); break;
        case 3: HOIG_SMALL_WG(3); break;
        default: HOIG_SMALL_WG(4); break;
    }
#undef HOIG_SMALL_WG
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
