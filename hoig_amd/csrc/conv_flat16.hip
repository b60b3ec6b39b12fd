// KS x KS stride-1 convolutions over narrow maps and their data gradients, on v_mfma_f32_16x16x32, with an LDS-resident input halo
// on a FLATTENED pixel axis: the attention's two VALID 5x5 convolutions over replicate-padded 36- and 40-pixel-wide maps
// (models/networks/extract_attn.py:18; 32-pixel tile columns would waste 44 % of a 36-wide output), and the 3x3 "same" layers whose
// launch has too few tiles for conv_halo3_m16_kernel (the data gradient of SPADE's 128 -> 1024 convolutions: N = 128 at 32 x 32),
// as a valid convolution over the zero-padded canvas, split over the channel blocks.
//
// The batch is one sequence of B * Hc * Wc "canvas" positions q = (b, y, x); out[q] = sum_{r,s} in[q + r * Wc + s] * w[r][s] is
// the valid convolution wherever (y, x) is a valid output position, and every other q (the last KS-1 columns and rows: 19 % of a
// 40-wide canvas) is computed and dropped.  A workgroup owns 256 consecutive positions x 128 output channels; per 32-channel block
// it stages the 256 + (KS-1) * (Wc + 1) positions its taps touch ONCE (split to 16-bit hi / lo), every tap reads its fragments
// out of that image at a lane-uniform offset (r * Wc + s) * 32 B, and only the weight tiles stream (three taps per step,
// double-buffered) -- the structure of conv_halo3_m16_kernel (conv_halo16.hip) on a 1-D image.  The generic implicit GEMM
// re-gathers its A tile from L2 for each of the 25 taps and ran these layers at 160-300 TFLOP/s.
//
//   forward      : canvas = the input grid (Hc = Hi, Wc = Wi), source = x, outputs kept for y < Ho, x < Wo;
//   data gradient: dx[y][x] = sum dy[y - r][x - s] w[r][s] is the same sum over the canvas of the INPUT grid with the source dy placed
//                  at offset (KS-1, KS-1) (zero elsewhere) and the taps flipped: every canvas position is a valid output.
// Few tiles and a long K (N = 128: 50 tiles at batch 8): the channel blocks are split over blockIdx.y and the partial results
// added with fp32 atomics (pixels on the result's rows, v_permlane16_swap -> two 128-B runs per atomic instruction: conv_igemm16.hip).
#include "conv_bf16_common.h"
#include "tuning.h"

namespace hoig_detail {
namespace {

template <bool F16>
__device__ __forceinline__ f32x4 mfma_m16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
constexpr int round128(int v) { return (v + 127) / 128 * 128; }

constexpr int PT = 256, BN = 128, TPS = 3, HSL = 8;      // positions / channels per workgroup, taps per step, halo slices per thread
constexpr int W23 = BN * 32 + 64, PLANE_W = round128(W23 + BN * 32);

// WDMA: the weight tiles of a step by LDS-DMA, as in conv_halo3_m16_kernel (conv_halo16.hip: same plane layout, same LDS images)
template <int NSX, int KS, bool F16, bool SPLITK, bool WDMA = false>
__global__ __launch_bounds__(512) void conv_flat_m16_kernel(const FlatArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;
    constexpr int NT = 512, WN = 2, MT = 4, NTW = 4, KK = KS * KS;
    constexpr int NGRP = (KK + TPS - 1) / TPS;            // steps per channel block
    constexpr int BBUF = TPS * NB * PLANE_W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HP = p.HP;
    const int P23 = round128(HP * 32) + 64, PLANE_P = round128(P23 + HP * 32);
    unsigned char *Ph = smem, *Pl = smem + PLANE_P;
    unsigned char *Wbase = smem + NS * PLANE_P;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int n_mt = p.nblk / p.nblk_n;
    const int q0 = (tile % n_mt) * PT, n0 = (tile / n_mt) * BN;      // channel-tile major: an XCD streams few weight tiles
    // split over K: this workgroup multiplies steps [s_begin, s_end) of the (channel block, tap group) walk
    const int s_begin = blockIdx.y * p.steps_per_split, s_end = min((p.Cg >> 5) * ((KS * KS + TPS - 1) / TPS), s_begin + p.steps_per_split);

    // halo: thread -> (position h, 4-channel group c4); the source offset of a position does not depend on the channel block
    int src_off[HSL];
#pragma unroll
    for (int sl = 0; sl < HSL; ++sl) {
        const int i = tid + NT * sl, h = i >> 3;
        int off = -1;
        if (h < HP) {
            const int q = q0 + h;
            if (q < p.Q) {
                const int b = q / p.HWc, rem = q - b * p.HWc;
                const int y = rem / p.Wc, x = rem - y * p.Wc;
                const int sy = y - p.oy, sx = x - p.ox;
                if (sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws) off = ((b * p.Hs + sy) * p.Ws + sx) * p.Cg + (i & 7) * 4;
            }
        }
        src_off[sl] = off;
    }
    const int brow = tid >> 2, bpos = tid & 3;            // weight staging: 128 rows x 4 chunk positions = 512 threads
    const unsigned short *wrow_h, *wrow_l;
    int woff;
    {
        const int n = n0 + brow;
        const size_t o = ((size_t)(n >> 5) * (p.K >> 5)) * 1024 + (n & 31) * 32 + bpos * 8;
        wrow_h = n < p.N ? p.Wh + o : nullptr;
        wrow_l = (NB == 2 && n < p.N) ? p.Wl + o : nullptr;
        const int c = bpos ^ ((brow >> 2) & 3);
        woff = (c >> 1) * W23 + brow * 32 + (c & 1) * 16;
    }
    int wread[NTW], pread[MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wread[j] = (lg >> 1) * W23 + (wn * (NTW * 16) + j * 16 + l15) * 32 + (lg & 1) * 16;
#pragma unroll
    for (int m = 0; m < MT; ++m) pread[m] = (lg >> 1) * P23 + (wm * 64 + m * 16 + l15) * 32 + (lg & 1) * 16;

    f32x4 acc[NTW][MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 rbh[TPS], rbl[TPS];
    const int T = s_end - s_begin;                        // step = (channel block, group of TPS taps)
    auto load_b = [&](int step) {
        const int cb = (s_begin + step) / NGRP, g = (s_begin + step) % NGRP;
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            const int tap = min(g * TPS + t, KK - 1);
            const int wtap = p.flip ? (KK - 1 - tap) : tap;
            const size_t koff = (size_t)(wtap * p.Cg + cb * 32) * 32;
            rbh[t] = wrow_h ? *reinterpret_cast<const uint4 *>(wrow_h + koff) : make_uint4(0, 0, 0, 0);
            if (NB == 2) rbl[t] = wrow_l ? *reinterpret_cast<const uint4 *>(wrow_l + koff) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            unsigned char *Wh = Wbase + buf * BBUF + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
            *reinterpret_cast<uint4 *>(Wh + woff) = rbh[t];
            if (NB == 2) *reinterpret_cast<uint4 *>(Wl + woff) = rbl[t];
        }
    };
    // ---- WDMA: piece q of a step = (tap t of the group, plane, 32-row block, half image); wave w issues pieces w, w + 8, ..
    constexpr int NPIECE_STEP = TPS * NB * (BN / 32) * 2, NPW = (NPIECE_STEP + 7) / 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_w0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)Wbase);
    const int drow = lane >> 1;
    const unsigned dlane0 = drow * 32 + (((0 + (lane & 1)) ^ ((drow >> 2) & 3)) << 3);
    const unsigned dlane1 = drow * 32 + (((2 + (lane & 1)) ^ ((drow >> 2) & 3)) << 3);
    auto dma_piece = [&](int step, int buf, int i) {
        const int q = wave_u + 8 * i;
        if (q >= NPIECE_STEP) return;
        const int h = q & 1, blk = (q >> 1) % (BN / 32), tp = (q >> 1) / (BN / 32);
        const int t = tp / NB, pl = tp - t * NB;
        const int cb = (s_begin + step) / NGRP, g = (s_begin + step) % NGRP;
        const int tap = min(g * TPS + t, KK - 1);
        const int wtap = p.flip ? (KK - 1 - tap) : tap;
        const size_t koff = (size_t)(wtap * p.Cg + cb * 32) * 32;
        const unsigned short *src = (pl ? p.Wl : p.Wh) + ((size_t)((n0 >> 5) + blk) * (p.K >> 5)) * 1024 + koff + (h ? dlane1 : dlane0);
        const unsigned to = __builtin_amdgcn_readfirstlane(lds_w0 + buf * BBUF + tp * PLANE_W + h * W23 + blk * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(to) : "memory");
    };
    float4 hreg[HSL];
    auto halo_load = [&](int cb) {
#pragma unroll
        for (int sl = 0; sl < HSL; ++sl)
            hreg[sl] = src_off[sl] >= 0 ? *reinterpret_cast<const float4 *>(p.A + (size_t)src_off[sl] + cb * 32)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int sl = 0; sl < HSL; ++sl) {
            const int i = tid + NT * sl, h = i >> 3, c4 = i & 7;
            if (h < HP) {
                uint2 hi, lo;
                split4t<F16>(hreg[sl], hi, lo);
                const int off = (c4 >> 2) * P23 + h * 32 + (c4 & 3) * 8;
                *reinterpret_cast<uint2 *>(Ph + off) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
            }
        }
    };
    struct PF {
        bf16x8 h[MT], l[MT];
    };
    struct WF {
        bf16x8 h, l;
    };
    // a group = (tap, channel tile): its weight fragments against the tap's eight pixel fragments (read once per tap); the
    // fragments of the next group, and a quarter of the next tap's pixel fragments, are read before this group's MFMAs issue
    auto compute = [&](int g, int bbuf, int dma_step) {
        const unsigned char *Wst = Wbase + bbuf * BBUF;
        const int ntap = min(TPS, KK - g * TPS);
        int tapoff[TPS];
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            const int tap = min(g * TPS + t, KK - 1), r = tap / KS, s_ = tap - r * KS;
            tapoff[t] = (r * p.Wc + s_) * 32;
        }
        auto read_p = [&](PF &f, int t, int m) {
            f.h[m] = *reinterpret_cast<const bf16x8 *>(Ph + pread[m] + tapoff[t]);
            if (NS == 2) f.l[m] = *reinterpret_cast<const bf16x8 *>(Pl + pread[m] + tapoff[t]);
        };
        auto read_w = [&](WF &f, int t, int j) {
            const unsigned char *Wh = Wst + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
            f.h = *reinterpret_cast<const bf16x8 *>(Wh + wread[j]);
            if (NB == 2) f.l = *reinterpret_cast<const bf16x8 *>(Wl + wread[j]);
        };
        PF pf[2];
        WF wf[2];
#pragma unroll
        for (int m = 0; m < MT; ++m) read_p(pf[0], 0, m);
        read_w(wf[0], 0, 0);
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            if (t >= ntap) break;                              // (the last group of a channel block holds KK % TPS taps)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int gi = t * NTW + j;
                if (gi + 1 < TPS * NTW) read_w(wf[(gi + 1) & 1], (gi + 1) / NTW, (gi + 1) % NTW);
                if (t + 1 < TPS) read_p(pf[(t + 1) & 1], t + 1, j);
                if constexpr (WDMA) {
                    if (dma_step >= 0 && gi < NPW) dma_piece(dma_step, bbuf ^ 1, gi);
                }
                __builtin_amdgcn_sched_barrier(0);
                const PF &pc = pf[t & 1];
                const WF &wc = wf[gi & 1];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (SPLITK) {
                        if (NS == 2) acc[j][m] = mfma_m16<F16>(pc.l[m], wc.h, acc[j][m]);
                        if (NB == 2) acc[j][m] = mfma_m16<F16>(pc.h[m], wc.l, acc[j][m]);
                        acc[j][m] = mfma_m16<F16>(pc.h[m], wc.h, acc[j][m]);
                    } else {
                        if (NS == 2) acc[j][m] = mfma_m16<F16>(wc.h, pc.l[m], acc[j][m]);
                        if (NB == 2) acc[j][m] = mfma_m16<F16>(wc.l, pc.h[m], acc[j][m]);
                        acc[j][m] = mfma_m16<F16>(wc.h, pc.h[m], acc[j][m]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (WDMA) {                                  // (a channel block's last group has KK % TPS taps: fewer slots than pieces)
            if (dma_step >= 0) {
#pragma unroll
                for (int i = 0; i < NPW; ++i)
                    if (i >= ntap * NTW) dma_piece(dma_step, bbuf ^ 1, i);
            }
        }
    };

    if (T > 0) {
        halo_load(s_begin / NGRP);
        halo_store();
        if constexpr (WDMA) {
#pragma unroll
            for (int i = 0; i < NPW; ++i) dma_piece(0, 0, i);
        } else {
            load_b(0);
            store_b(0);
            if (T > 1) load_b(1);
        }
    }
    if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int bbuf = 0;
#pragma unroll 1
    for (int step = 0; step < T; ++step) {
        const int cb = (s_begin + step) / NGRP, g = (s_begin + step) - cb * NGRP;
        const bool more = step + 1 < T;
        const bool boundary = more && g == NGRP - 1;
        if constexpr (!WDMA) {
            if (more) store_b(bbuf ^ 1);              // weights of step+1 (registers loaded during the previous step)
            if (step + 2 < T) load_b(step + 2);
        }
        if (boundary) halo_load(cb + 1);
        compute(g, bbuf, (WDMA && more) ? step + 1 : -1);
        if (boundary) {
            __syncthreads();                          // every wave is done with the halo
            halo_store();
        }
        if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        bbuf ^= 1;
    }

    // destination of canvas position q: kept if (y, x) lies inside the Hd x Wd output grid
    auto dest = [&](int q) -> long {
        if (q >= p.Q) return -1;
        const int b = q / p.HWc, rem = q - b * p.HWc;
        const int y = rem / p.Wc, x = rem - y * p.Wc;
        if (y >= p.Hd || x >= p.Wd) return -1;
        return ((long)(b * p.Hd + y) * p.Wd + x) * p.N;
    };
    if (!SPLITK) {
        // lane -> position (lane & 15) of tile m, channels 4 * (lane >> 4) .. + 3 of channel tile j
        const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
        const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const long d = dest(q0 + wm * 64 + m * 16 + l15);
            if (d < 0) continue;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = n0 + wn * 64 + j * 16 + lg * 4;
                if (n < p.N) {
                    float4 o = make_float4(acc[j][m][0] * p.oscale, acc[j][m][1] * p.oscale, acc[j][m][2] * p.oscale,
                                           acc[j][m][3] * p.oscale);
                    if (p.bias) {
                        const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
                        o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                    }
                    o.x = fast_act(o.x, nslope, special, p.act, p.slope); o.y = fast_act(o.y, nslope, special, p.act, p.slope);
                    o.z = fast_act(o.z, nslope, special, p.act, p.slope); o.w = fast_act(o.w, nslope, special, p.act, p.slope);
                    if (p.addend) {
                        const float4 ad = *reinterpret_cast<const float4 *>(p.addend + d + n);
                        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                    }
                    *reinterpret_cast<float4 *>(p.C + d + n) = o;
                }
            }
        }
    } else {
        // register r of acc[j][m]: position 4 * (lane >> 4) + r of tile m, channel (lane & 15) of tile j; after the swap of the
        // registers of tiles j, j + 1 lanes 0-31 / 32-63 hold channels 0..31 of the pair at positions r / 8 + r (first) and
        // 4 + r / 12 + r (second)
        const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                long d[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) d[h] = dest(q0 + wm * 64 + m * 16 + r + 8 * lh + 4 * h);
#pragma unroll
                for (int j = 0; j < NTW; j += 2) {
                    const int n = n0 + wn * 64 + j * 16 + l31;
                    const u2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[j][m][r]), __float_as_uint(acc[j + 1][m][r]),
                                                                     false, false);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (d[h] >= 0 && n < p.N) {
                            float v = __uint_as_float(sw[h]) * p.oscale;
                            if (blockIdx.y == 0 && p.bias) v += p.bias[n];
                            atomicAdd(p.C + d[h] + n, v);
                        }
                }
            }
    }
}

template <int NSX, int KS, bool F16, bool SPLITK>
int launch_one(const FlatArgs &a, dim3 grid, size_t shm, hipStream_t st) {
    if (hoig_tuning(HOIG_TUNE_WDMA16) >= 2) {              // (2: the flattened-axis kernel too)
        static hoig_once once_d;
        if (!once_d.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_flat_m16_kernel<NSX, KS, F16, SPLITK, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return HOIG_ELAUNCH;
            once_d.set();
        }
        conv_flat_m16_kernel<NSX, KS, F16, SPLITK, true><<<grid, 512, shm, st>>>(a);
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_flat_m16_kernel<NSX, KS, F16, SPLITK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    conv_flat_m16_kernel<NSX, KS, F16, SPLITK><<<grid, 512, shm, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

template <int KS>
int launch_ks(const FlatArgs &a, int ns, dim3 grid, size_t shm, bool split, hipStream_t st) {
    if (split) {
        if (a.f16) HOIG_NS_SWITCH(ns, return launch_one<NSX, KS, true, true>(a, grid, shm, st));
        else HOIG_NS_SWITCH(ns, return launch_one<NSX, KS, false, true>(a, grid, shm, st));
    }
    if (a.f16) HOIG_NS_SWITCH(ns, return launch_one<NSX, KS, true, false>(a, grid, shm, st));
    else HOIG_NS_SWITCH(ns, return launch_one<NSX, KS, false, false>(a, grid, shm, st));
    return HOIG_EINVAL;
}

}  // namespace

// `a`: A / Wh / Wl / bias / C / addend, Bn, KS (3 or 5), the canvas (Hc, Wc), the source grid (Hs, Ws) and its offset (oy, ox) on the
// canvas, the destination grid (Hd, Wd), Cg, N, K, flip, act / slope, f16, oscale filled in by the caller
int launch_flat_m16(FlatArgs a, int ns, hipStream_t st) {
    const int KS = a.KS;
    if ((KS != 3 && KS != 5) || a.N % 128 || a.Cg % 32) return HOIG_EUNSUPPORTED;
    a.HWc = a.Hc * a.Wc;
    a.Q = a.Bn * a.HWc;
    a.HP = PT + (KS - 1) * (a.Wc + 1);
    if (a.HP * 8 > HSL * 512) return HOIG_EUNSUPPORTED;
    const int p23 = round128(a.HP * 32) + 64, plane_p = round128(p23 + a.HP * 32);
    const size_t shm = (size_t)ns_a(ns) * plane_p + 2 * TPS * ns_b(ns) * PLANE_W;
    if (shm > 160 * 1024) return HOIG_EUNSUPPORTED;
    const int n_mt = (int)hoig_cdiv(a.Q, PT);
    a.nblk_n = a.N / BN;
    a.nblk = n_mt * a.nblk_n;
    const int ncb = a.Cg >> 5;
    const int steps = ncb * ((KS * KS + TPS - 1) / TPS);
    int split = 1;
    if (a.nblk <= 128 && !a.addend && a.act == HOIG_ACT_NONE) {      // few tiles, long K: split the step walk, add with atomics
        split = 256 / a.nblk;                     // ONE round of workgroups on the 256 CUs: 300 of them would take two
        if (split > steps / 8) split = steps / 8;
        if (split < 1) split = 1;
    }
    a.steps_per_split = (int)hoig_cdiv(steps, split);
    split = (int)hoig_cdiv(steps, a.steps_per_split);
    dim3 grid(a.nblk, split);
    if (split > 1 && hipMemsetAsync(a.C, 0, (size_t)a.Bn * a.Hd * a.Wd * a.N * sizeof(float), st) != hipSuccess) return HOIG_ELAUNCH;
    return KS == 3 ? launch_ks<3>(a, ns, grid, shm, split > 1, st) : launch_ks<5>(a, ns, grid, shm, split > 1, st);
}

}  // namespace hoig_detail
