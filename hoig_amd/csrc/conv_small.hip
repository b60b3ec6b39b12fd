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

// Forward, register-blocked: a thread owns FOUR adjacent output pixels of one row, so the 10 input pixels a tap row needs
// are read from LDS once and feed 7 taps x 4 pixels (0.36 LDS reads per tap-pixel instead of 1), and the channel pairs of
// a float4 go through v_pk_fma_f32 (two FMAs per lane per instruction: the 157 TF form of the fp32 VALU).
// Tile: 16 rows x 64 columns per workgroup; the halo tile is staged 8 channels at a time as two float4 planes
// [plane][row][slot], slot = px + (px >> 4) (one pad slot per 16 pixels) with rows 80 slots apart: the 16 lanes a
// ds_read_b128 services together (pixels 4*tg + j of up to two rows) then fall on 16 distinct 16-B bank slots.
typedef float f2_t __attribute__((ext_vector_type(2)));
constexpr int F4_TW = 64, F4_TH = 16, F4_ROW = 80, F4_CC = 8;

template <int CO>
__global__ __launch_bounds__(256) void conv_small_fwd4_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ y,
                                                              int B, int H, int W, int Ci, int pad, unsigned long long acts,
                                                              float slope, int CoReal) {
    constexpr int HT = F4_TH + KS - 1, WT = F4_TW + KS - 1;          // 22 x 70 halo pixels
    constexpr int PLANE = HT * F4_ROW;                               // float4 slots per plane
    __shared__ float4 xs[2 * PLANE];                                 // 56 KB
    __shared__ float4 wsm[KS * KS * CO * 2];                         // this chunk's weights [tap][n][plane] (broadcast reads)
    const int tiles_x = (W + F4_TW - 1) / F4_TW, tiles_y = (H + F4_TH - 1) / F4_TH;
    int t = blockIdx.x;
    const int bx = t % tiles_x;
    t /= tiles_x;
    const int by = t % tiles_y, b = t / tiles_y;
    const int tg = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int y0 = by * F4_TH - pad, x0 = bx * F4_TW - pad;
    f2_t acc[4][CO];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[j][n] = (f2_t){0.f, 0.f};
    const size_t wn = (size_t)KS * KS * Ci;                          // weight stride between output channels
    // the next 8-channel chunk of the halo is fetched into registers while the current one is multiplied (the fetch is an
    // HBM-latency wait of about the length of a chunk's arithmetic; two workgroups per CU do not hide it alone)
    constexpr int NSL = (HT * WT * 2 + 255) / 256;
    float4 stage[NSL];
    constexpr int NW = (KS * KS * CO * 2 + 255) / 256;
    float4 wstage[NW];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = threadIdx.x + 256 * k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < KS * KS * CO * 2) {
                const int pl = i & 1, n = (i >> 1) % CO, tap = (i >> 1) / CO;
                v = *reinterpret_cast<const float4 *>(w + (size_t)n * wn + (size_t)tap * Ci + c0 + pl * 4);
            }
            wstage[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int i = threadIdx.x + 256 * k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HT * WT * 2) {
                const int pl = i & 1, pix = i >> 1;
                const int py = pix / WT, px = pix - py * WT;
                const int gy = y0 + py, gx = x0 + px;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v = *reinterpret_cast<const float4 *>(x + (((size_t)b * H + gy) * W + gx) * Ci + c0 + pl * 4);
            }
            stage[k] = v;
        }
    };
    auto publish = [&]() {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < KS * KS * CO * 2) wsm[i] = wstage[k];
        }
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < HT * WT * 2) {
                const int pl = i & 1, pix = i >> 1;
                const int py = pix / WT, px = pix - py * WT;
                xs[pl * PLANE + py * F4_ROW + px + (px >> 4)] = stage[k];
            }
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Ci; c0 += F4_CC) {
        __syncthreads();                                             // every wave is done with the previous chunk
        publish();
        __syncthreads();
        if (c0 + F4_CC < Ci) fetch(c0 + F4_CC);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll 1
            for (int r = 0; r < KS; ++r) {
                const float4 *row = xs + pl * PLANE + (ty + r) * F4_ROW;
                float4 in[4 + KS - 1];
#pragma unroll
                for (int q = 0; q < 4 + KS - 1; ++q) {
                    const int px = 4 * tg + q;
                    in[q] = row[px + (px >> 4)];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
#pragma unroll
                    for (int n = 0; n < CO; ++n) {
                        const float4 ww = wsm[((r * KS + s) * CO + n) * 2 + pl];
                        const f2_t w01 = {ww.x, ww.y}, w23 = {ww.z, ww.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f2_t v01 = {in[j + s].x, in[j + s].y}, v23 = {in[j + s].z, in[j + s].w};
                            acc[j][n] = __builtin_elementwise_fma(v01, w01, acc[j][n]);
                            acc[j][n] = __builtin_elementwise_fma(v23, w23, acc[j][n]);
                        }
                    }
                }
            }
        }
    }
    const int oy = by * F4_TH + ty;
    if (oy < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = bx * F4_TW + 4 * tg + j;
            if (ox < W) {
                float *o = y + (((size_t)b * H + oy) * W + ox) * CoReal;
#pragma unroll
                for (int n = 0; n < CO; ++n)
                    o[n] = hoig_act(acc[j][n].x + acc[j][n].y + (bias ? bias[n] : 0.f), (int)((acts >> (4 * n)) & 15), slope);
            }
        }
    }
}

// Data gradient of the same heads: dx[p][c] = sum_{r,s,n} dy[p + pad - (r,s)][n] * w[n][r][s][c].  On the implicit GEMM this
// is N = Ci columns over a K = 49*Co <= 147 reduction -- mostly padding.  Here: the same 16x64 pixel tile and the same four
// adjacent pixels per thread; the dy halo tile (Co <= 4 values per pixel = one float4) and the 16-channel weight slice
// [tap][n][4 quads] sit in LDS; a thread accumulates 4 pixels x 16 channels (32 packed accumulators); blockIdx.y walks the
// 16-channel groups.
// FWD = true turns the same kernel into the FORWARD of the first-layer convs (3 / 8 input channels -> 64): "dy" is then the
// input image (channels s_off .. s_off+CO-1 of its CoReal), "dx" the output, taps are walked unflipped, the weight slice is
// gathered from w[co][tap][ci] and, for the second 4-channel group of an 8-channel input, the result accumulates into y.
template <int CO, bool FWD>
__global__ __launch_bounds__(256) void conv_small_dgrad4_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                                float *__restrict__ dx, int B, int H, int W, int Ci,
                                                                int pad, int CoReal, int s_off, int accumulate,
                                                                const float *__restrict__ bias, int act, float slope) {
    constexpr int HT = F4_TH + KS - 1, WT = F4_TW + KS - 1;
    constexpr int PLANE = HT * F4_ROW;
    __shared__ float4 gs[PLANE];                                     // dy halo, channels n in .x .y .z .w
    __shared__ float4 wsm[KS * KS * CO * 4];                         // [tap][n][quad] : w[n][tap][c0 + 4*quad ..]
    const int tiles_x = (W + F4_TW - 1) / F4_TW, tiles_y = (H + F4_TH - 1) / F4_TH;
    int t = blockIdx.x;
    const int bx = t % tiles_x;
    t /= tiles_x;
    const int by = t % tiles_y, b = t / tiles_y;
    const int c0 = blockIdx.y * 16;
    const int tg = threadIdx.x & 15, ty = threadIdx.x >> 4;
    // dx[p] gathers dy at p + pad - tap: halo origin = tile origin + pad - (KS-1);  forward: x at p - pad + tap
    const int y0 = by * F4_TH + (FWD ? -pad : pad - (KS - 1)), x0 = bx * F4_TW + (FWD ? -pad : pad - (KS - 1));
    for (int i = threadIdx.x; i < HT * WT; i += 256) {
        const int py = i / WT, px = i - py * WT;
        const int gy = y0 + py, gx = x0 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const float *g = dy + (((size_t)b * H + gy) * W + gx) * CoReal + s_off;
            v.x = g[0];
            if (CO > 1) v.y = g[1];
            if (CO > 2) v.z = g[2];
            if (CO > 3) v.w = g[3];
        }
        gs[py * F4_ROW + px + (px >> 4)] = v;
    }
    const size_t wn = (size_t)KS * KS * Ci;
    for (int i = threadIdx.x; i < KS * KS * CO * 4; i += 256) {
        const int q = i & 3, n = (i >> 2) % CO, tap = (i >> 2) / CO;
        if (FWD) {      // w[co][tap][ci]: Ci = output channels here, CoReal = input channels
            const float *wp = w + ((size_t)(c0 + q * 4) * KS * KS + tap) * CoReal + s_off + n;
            const size_t st_ = (size_t)KS * KS * CoReal;
            wsm[i] = make_float4(wp[0], wp[st_], wp[2 * st_], wp[3 * st_]);
        } else {
            wsm[i] = *reinterpret_cast<const float4 *>(w + (size_t)n * wn + (size_t)tap * Ci + c0 + q * 4);
        }
    }
    __syncthreads();
    f2_t acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[j][k] = (f2_t){0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < KS; ++r) {
        // output pixel (ty, 4tg+j) reads halo row ty + (KS-1-r), column 4tg + j + (KS-1-s)   (forward: ty + r, 4tg + j + s)
        const float4 *row = gs + (ty + (FWD ? r : KS - 1 - r)) * F4_ROW;
        float4 in[4 + KS - 1];
#pragma unroll
        for (int q = 0; q < 4 + KS - 1; ++q) {
            const int px = 4 * tg + q;
            in[q] = row[px + (px >> 4)];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int n = 0; n < CO; ++n) {
                float4 wq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wq[q] = wsm[((r * KS + s) * CO + n) * 4 + q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 g4 = in[j + (FWD ? s : KS - 1 - s)];
                    const float g = n == 0 ? g4.x : (n == 1 ? g4.y : (n == 2 ? g4.z : g4.w));
                    const f2_t gg = {g, g};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[j][2 * q] = __builtin_elementwise_fma(gg, (f2_t){wq[q].x, wq[q].y}, acc[j][2 * q]);
                        acc[j][2 * q + 1] = __builtin_elementwise_fma(gg, (f2_t){wq[q].z, wq[q].w}, acc[j][2 * q + 1]);
                    }
                }
            }
        }
    }
    const int oy = by * F4_TH + ty;
    if (oy < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = bx * F4_TW + 4 * tg + j;
            if (ox < W) {
                float *o = dx + (((size_t)b * H + oy) * W + ox) * Ci + c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[j][2 * q].x, acc[j][2 * q].y, acc[j][2 * q + 1].x, acc[j][2 * q + 1].y);
                    if (FWD) {
                        if (accumulate) {
                            const float4 old_ = *reinterpret_cast<const float4 *>(o + 4 * q);
                            v.x += old_.x; v.y += old_.y; v.z += old_.z; v.w += old_.w;
                        }
                        if (bias) {
                            const float4 bb = *reinterpret_cast<const float4 *>(bias + c0 + 4 * q);
                            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                        }
                        v.x = hoig_act(v.x, act, slope); v.y = hoig_act(v.y, act, slope);
                        v.z = hoig_act(v.z, act, slope); v.w = hoig_act(v.w, act, slope);
                    }
                    *reinterpret_cast<float4 *>(o + 4 * q) = v;
                }
            }
        }
    }
}

// grid: (tile groups, Ci/16).  Threads: c = tid & 15, r = (tid >> 4) % R, half = (tid >> 4) / R  (needs 2*16*R <= 256)
// (running the stems' weight gradient -- few INPUT channels -- on this kernel with the tensors exchanged measured 1.3-1.7x
// SLOWER than the fp32 MFMA wgrad kernel; they stay there)
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
#pragma unroll
                for (int px = 0; px < TILE; ++px) {      // fully unrolled: the sliding window becomes register renaming
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
// `acts`: the HOIG_ACT_* code of output channel n in bits [4n, 4n+4) (one activation per head of a fused launch)
int hoig_conv_small_fwd_acts(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                             unsigned long long acts, hipStream_t st) {
    if (d->transposed || d->stride != 1 || d->Co > 5 || (d->Ci % CC) || d->R != KS || d->S != KS) return HOIG_EUNSUPPORTED;
    if (d->Ho != d->Hi || d->Wo != d->Wi || 2 * d->pad != d->R - 1 || d->R != d->S) return HOIG_EUNSUPPORTED;
    const int tiles = d->B * (int)hoig_cdiv(d->Hi, F4_TH) * (int)hoig_cdiv(d->Wi, F4_TW);
#define HOIG_SMALL_FWD(N)                                                                                          \
    conv_small_fwd4_kernel<N><<<tiles, 256, 0, st>>>(x, w, bias, y, d->B, d->Hi, d->Wi, d->Ci, d->pad, acts, d->slope, d->Co)
    switch (d->Co) {
        case 1: HOIG_SMALL_FWD(1); break;
        case 2: HOIG_SMALL_FWD(2); break;
        case 3: HOIG_SMALL_FWD(3); break;
        case 4: HOIG_SMALL_FWD(4); break;
        default: HOIG_SMALL_FWD(5); break;
    }
#undef HOIG_SMALL_FWD
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
int hoig_conv_small_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                        hipStream_t st) {
    if (d->Co > 4) return HOIG_EUNSUPPORTED;
    unsigned long long acts = 0;
    for (int n = 0; n < 16; ++n) acts |= (unsigned long long)(d->act & 15) << (4 * n);
    return hoig_conv_small_fwd_acts(d, x, w, bias, y, acts, st);
}

// forward of 7x7 stride-1 "same" convs with few INPUT channels (3, 8: the stems) and Co % 16 == 0: groups of <= 4 input
// channels, the activation (and bias) applied by the last group's launch
int hoig_conv_small_ci_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                           hipStream_t st) {
    if (d->transposed || d->stride != 1 || d->Ci > 8 || (d->Co % 16) || d->R != KS || d->S != KS) return HOIG_EUNSUPPORTED;
    if (d->Ho != d->Hi || d->Wo != d->Wi || 2 * d->pad != d->R - 1) return HOIG_EUNSUPPORTED;
    dim3 grid(d->B * (unsigned)hoig_cdiv(d->Hi, F4_TH) * (unsigned)hoig_cdiv(d->Wi, F4_TW), d->Co / 16);
    for (int off = 0; off < d->Ci; off += 4) {
        const int n = d->Ci - off < 4 ? d->Ci - off : 4;
        const bool last = off + 4 >= d->Ci;
        const float *bb = last ? bias : nullptr;
        const int act = last ? d->act : HOIG_ACT_NONE;
#define HOIG_SMALL_CF(N) \
    conv_small_dgrad4_kernel<N, true><<<grid, 256, 0, st>>>(x, w, y, d->B, d->Hi, d->Wi, d->Co, d->pad, d->Ci, off,       \
                                                            off > 0 ? 1 : 0, bb, act, d->slope)
        switch (n) {
            case 1: HOIG_SMALL_CF(1); break;
            case 2: HOIG_SMALL_CF(2); break;
            case 3: HOIG_SMALL_CF(3); break;
            default: HOIG_SMALL_CF(4); break;
        }
#undef HOIG_SMALL_CF
        HOIG_LAUNCH_CHECK();
    }
    return HOIG_OK;
}

int hoig_conv_small_dgrad(const hoig_conv_desc *d, const float *dy, const float *w, float *dx, hipStream_t st) {
    if (d->transposed || d->stride != 1 || d->Co > 4 || (d->Ci % 16) || d->R != KS || d->S != KS) return HOIG_EUNSUPPORTED;
    if (d->Ho != d->Hi || d->Wo != d->Wi || 2 * d->pad != d->R - 1) return HOIG_EUNSUPPORTED;
    dim3 grid(d->B * (unsigned)hoig_cdiv(d->Hi, F4_TH) * (unsigned)hoig_cdiv(d->Wi, F4_TW), d->Ci / 16);
#define HOIG_SMALL_DG(N) \
    conv_small_dgrad4_kernel<N, false><<<grid, 256, 0, st>>>(dy, w, dx, d->B, d->Hi, d->Wi, d->Ci, d->pad, d->Co, 0, 0, \
                                                             nullptr, 0, 0.f)
    switch (d->Co) {
        case 1: HOIG_SMALL_DG(1); break;
        case 2: HOIG_SMALL_DG(2); break;
        case 3: HOIG_SMALL_DG(3); break;
        default: HOIG_SMALL_DG(4); break;
    }
#undef HOIG_SMALL_DG
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


// ---------------------------------------------------------------------------------------------------------------------
// Convolutions with <= 4 output channels over MANY input channels (the PatchGAN head, discriminator.py:46: 4x4, 512 -> 1):
// every output is a dot product over K = R*S*Ci >= 1024 values.  On the GEMM kernels the single output channel sits in a
// 128-wide tile and a handful of workgroups walk K serially (385 us for 1568 outputs).  Here: one wave per output pixel,
// lanes stride over (tap, 4 channels) with coalesced float4 reads, one reduction per output channel.
namespace {
template <int CO>
__global__ __launch_bounds__(256) void conv_dot_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ y, int B, int Hi,
                                                          int Wi, int Ci, int Ho, int Wo, int R, int S, int stride, int pad,
                                                          int act, float slope) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (int64_t)B * Ho * Wo) return;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
    const int CV = Ci >> 2, K4 = R * S * CV;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int i = lane; i < K4; i += 64) {
        const int tap = i / CV, cv = i - tap * CV;
        const int r = tap / S, t = tap - r * S;
        const int iy = oy * stride - pad + r, ix = ox * stride - pad + t;
        if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
        const float4 xv = *reinterpret_cast<const float4 *>(x + (((size_t)b * Hi + iy) * Wi + ix) * Ci + cv * 4);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float4 wv = *reinterpret_cast<const float4 *>(w + ((size_t)c * R * S + tap) * Ci + cv * 4);
            acc[c] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
    const float nslope = act == HOIG_ACT_NONE ? 1.f : (act == HOIG_ACT_RELU ? 0.f : slope);
    const bool special = act == HOIG_ACT_TANH || act == HOIG_ACT_SIGMOID;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        float v = hoig_wave_sum(acc[c]);
        if (lane == 0) {
            v += bias ? bias[c] : 0.f;
            y[o * CO + c] = fast_act(v, nslope, special, act, slope);
        }
    }
}
}  // namespace

int hoig_conv_dot_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, hipStream_t st) {
    if (d->transposed || d->Co > 4 || (d->Ci & 3) || d->R * d->S * d->Ci < 1024) return HOIG_EUNSUPPORTED;
    const int64_t outs = (int64_t)d->B * d->Ho * d->Wo;
    const unsigned grid = (unsigned)hoig_cdiv(outs, 4);
#define HOIG_DOT_FWD(N)                                                                                                    \
    conv_dot_fwd_kernel<N><<<grid, 256, 0, st>>>(x, w, bias, y, d->B, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->R, d->S, d->stride, \
                                                 d->pad, d->act, d->slope)
    switch (d->Co) {
        case 1: HOIG_DOT_FWD(1); break;
        case 2: HOIG_DOT_FWD(2); break;
        case 3: HOIG_DOT_FWD(3); break;
        default: HOIG_DOT_FWD(4); break;
    }
#undef HOIG_DOT_FWD
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
