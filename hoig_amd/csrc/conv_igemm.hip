// Implicit-GEMM convolution family for gfx950 (CDNA4): forward, data gradient, weight gradient of
// Conv2d / ConvTranspose2d over NHWC fp32 activations with weights packed [Co][R][S][Ci].
//
// One GEMM view serves every case (C[m][n] = sum_k A[m][k] * Bw[n][k]):
//   conv  fwd   : m=(b,ho,wo) n=co k=(r,s,ci)  A = gather_F(x)   Bw = W[n][k]            (K-contiguous)
//   convT fwd   : m=(b,ho,wo) n=co k=(r,s,ci)  A = gather_T(x)   Bw = W[n][k]
//   conv  dgrad : m=(b,hi,wi) n=ci k=(r,s,co)  A = gather_T(dy)  Bw = W[co][r][s][n]     (N-contiguous)
//   convT dgrad : m=(b,hi,wi) n=ci k=(r,s,co)  A = gather_F(dy)  Bw = W[co][r][s][n]
//   wgrad       : dW[co][(r,s,ci)] += sum_m dy[m][co] * gather(x)[m][(r,s,ci)]  (reduction over pixels, split-K
//                 over blocks, fp32 atomics into the caller's flat gradient buffer)
// gather_F: hg = hp*stride - pad + r.   gather_T: hg = (hp + pad - r)/stride when divisible.
// For stride-2 gather_T the pixel index m is enumerated PHASE-MAJOR ((hp&1, wp&1) outermost) so that a tile has one
// parity class and the 3/4 of the taps that are structurally zero are skipped instead of multiplied.
//
// Arithmetic: HOIG_PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate, 64 FLOP/clk/SIMD);
// HOIG_PREC_BF16X3 / HOIG_PREC_BF16 use v_mfma_f32_32x32x16_bf16 on operands converted while staging into LDS
// (hi+lo split, 3 MFMAs per k-step, or a single one).
//
// Tiling: 256 threads = 4 waves per workgroup, BMxBN output tile, BK=32; the A tile is gathered global->registers
// (one 16-B load per lane along the channel axis, coalesced 128-B runs per pixel) while the previous tile is being
// multiplied, then written to LDS.  Block ids are remapped so that the n-tiles of one m-tile share an XCD (L2 reuse of
// the gathered rows).
#include "common.h"
#include <cstdlib>

namespace {

struct Geom {
    int Bn, Hg, Wg, Cg;  // gathered tensor [Bn][Hg][Wg][Cg]
    int Hp, Wp;          // pixel grid enumerated by m (per image)
    int R, S, stride, pad;
    int gatherT;      // 0: gather_F, 1: gather_T
    int phase_major;  // m enumerates (ph,pw,b,hp/2,wp/2)
    int tile_skip;    // tiles have a uniform phase and a k-block lies inside one tap -> skip dead taps
};

struct IgemmArgs {
    const float *A;
    const float *W;
    const float *bias;
    float *C;
    Geom g;
    int M, N, K;
    int act;
    float slope;
    int nblk_n, nblk;
};

struct WgradArgs {
    const float *X;   // gathered tensor
    const float *DY;  // [M][Co] over the pixel grid
    float *DW;        // [Co][K]
    Geom g;
    int M, Co, K;
    int nblk_n, nblk_mn, m_per_split;
};

__device__ __forceinline__ void decode_m(const Geom &g, int m, int &b, int &hp, int &wp) {
    if (!g.phase_major) {
        const int hw = g.Hp * g.Wp;
        b = m / hw;
        const int rem = m - b * hw;
        hp = rem / g.Wp;
        wp = rem - hp * g.Wp;
    } else {
        const int W2 = g.Wp >> 1, q = (g.Hp >> 1) * W2, bq = g.Bn * q;
        const int ph = m / bq;
        const int rem = m - ph * bq;
        b = rem / q;
        const int r2 = rem - b * q;
        const int h2 = r2 / W2;
        hp = 2 * h2 + (ph >> 1);
        wp = 2 * (r2 - h2 * W2) + (ph & 1);
    }
}

__device__ __forceinline__ int row_base(const Geom &g, int p) { return g.gatherT ? p + g.pad : p * g.stride - g.pad; }

// gathered coordinate for tap r, or -1 when the tap falls on padding / a structural zero
__device__ __forceinline__ int gcoord(const Geom &g, int base, int r, int lim) {
    if (!g.gatherT) {
        const int c = base + r;
        return (c >= 0 && c < lim) ? c : -1;
    }
    int t = base - r;
    if (t < 0) return -1;
    if (g.stride == 2) {
        if (t & 1) return -1;
        t >>= 1;
    } else if (g.stride != 1) {
        if (t % g.stride) return -1;
        t /= g.stride;
    }
    return t < lim ? t : -1;
}

__device__ __forceinline__ bool tap_alive(const Geom &g, int hp, int wp, int rs) {
    // stride-2 gather_T: tap (r,s) contributes to pixel parity class (hp&1, wp&1) iff both differences are even
    const int r = rs / g.S, s = rs - r * g.S;
    return (((hp + g.pad - r) & 1) == 0) && (((wp + g.pad - s) & 1) == 0);
}

template <int VW>
__device__ __forceinline__ float4 gather_vec(const Geom &g, const float *__restrict__ A, int pb, int bh, int bw, int k,
                                             int K) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VW == 4) {
        if (k < K && pb >= 0) {
            const int rs = k / g.Cg, c = k - rs * g.Cg;
            const int r = rs / g.S, s = rs - r * g.S;
            const int hg = gcoord(g, bh, r, g.Hg), wg = gcoord(g, bw, s, g.Wg);
            if (hg >= 0 && wg >= 0)
                v = *reinterpret_cast<const float4 *>(A + ((size_t)(pb + hg) * g.Wg + wg) * g.Cg + c);
        }
    } else {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (pb >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kj = k + j;
                if (kj < K) {
                    const int rs = kj / g.Cg, c = kj - rs * g.Cg;
                    const int r = rs / g.S, s = rs - r * g.S;
                    const int hg = gcoord(g, bh, r, g.Hg), wg = gcoord(g, bw, s, g.Wg);
                    if (hg >= 0 && wg >= 0) e[j] = A[((size_t)(pb + hg) * g.Wg + wg) * g.Cg + c];
                }
            }
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
    }
    return v;
}

// ------------------------------------------------------------------------------------------------- fwd / dgrad
template <int BM, int BN, int WM, int WN, int VW, bool B_NCONTIG>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IgemmArgs p) {
    constexpr int BK = 32;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int LDA = BM + 1;
    constexpr int LDB = B_NCONTIG ? BN + 4 : BN + 1;
    constexpr int RA = BM / 32, RB = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[BK * LDA + BK * LDB];
    float *As = smem, *Bs = smem + BK * LDA;

    const Geom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (tile / p.nblk_n) * BM, n0 = (tile % p.nblk_n) * BN;

    // per-thread gather rows
    const int kc = (tid & 7) * 4, lrow = tid >> 3;
    int pb[RA], bh[RA], bw[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            int b, hp, wp;
            decode_m(g, m, b, hp, wp);
            pb[i] = b * g.Hg;
            bh[i] = row_base(g, hp);
            bw[i] = row_base(g, wp);
        } else {
            pb[i] = -1;
            bh[i] = bw[i] = 0;
        }
    }
    int t_hp = 0, t_wp = 0;
    if (g.tile_skip) {
        int b;
        decode_m(g, m0, b, t_hp, t_wp);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[RA], rb[RB];
    const int nkb = (p.K + BK - 1) / BK;

    auto next_kb = [&](int kb) {
        if (g.tile_skip)
            while (kb < nkb && !tap_alive(g, t_hp, t_wp, (kb * BK) / g.Cg)) ++kb;
        return kb;
    };

    auto load_tiles = [&](int kb) {
        const int k = kb * BK + kc;
#pragma unroll
        for (int i = 0; i < RA; ++i) ra[i] = gather_vec<VW>(g, p.A, pb[i], bh[i], bw[i], k, p.K);
        if (!B_NCONTIG) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int n = n0 + lrow + 32 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N) {
                    const float *src = p.W + (size_t)n * p.K + k;
                    if (VW == 4) {
                        if (k < p.K) v = *reinterpret_cast<const float4 *>(src);
                    } else {
                        if (k < p.K) v.x = src[0];
                        if (k + 1 < p.K) v.y = src[1];
                        if (k + 2 < p.K) v.z = src[2];
                        if (k + 3 < p.K) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
        } else {
            // Bw[n][k=(rs,co)] = W[co][rs][n]
            constexpr int CPR = BN / 4;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int idx = tid + 256 * i;
                const int krow = idx / CPR, n = n0 + (idx % CPR) * 4;
                const int kk = kb * BK + krow;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < p.K && n < p.N) {
                    const int rs = kk / g.Cg, co = kk - rs * g.Cg;
                    const float *src = p.W + ((size_t)co * (g.R * g.S) + rs) * p.N + n;
                    if ((p.N & 3) == 0) {
                        v = *reinterpret_cast<const float4 *>(src);
                    } else {
                        v.x = src[0];
                        if (n + 1 < p.N) v.y = src[1];
                        if (n + 2 < p.N) v.z = src[2];
                        if (n + 3 < p.N) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int r = lrow + 32 * i;
            As[(kc + 0) * LDA + r] = ra[i].x;
            As[(kc + 1) * LDA + r] = ra[i].y;
            As[(kc + 2) * LDA + r] = ra[i].z;
            As[(kc + 3) * LDA + r] = ra[i].w;
        }
        if (!B_NCONTIG) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = lrow + 32 * i;
                Bs[(kc + 0) * LDB + r] = rb[i].x;
                Bs[(kc + 1) * LDB + r] = rb[i].y;
                Bs[(kc + 2) * LDB + r] = rb[i].z;
                Bs[(kc + 3) * LDB + r] = rb[i].w;
            }
        } else {
            constexpr int CPR = BN / 4;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int idx = tid + 256 * i;
                *reinterpret_cast<float4 *>(&Bs[(idx / CPR) * LDB + (idx % CPR) * 4]) = rb[i];
            }
        }
    };

    int kb = next_kb(0);
    if (kb < nkb) {
        load_tiles(kb);
        store_tiles();
    }
    __syncthreads();
    while (kb < nkb) {
        const int kn = next_kb(kb + 1);
        if (kn < nkb) load_tiles(kn);
        const float *a_base = As + wm * (TM * 32) + l31;
        const float *b_base = Bs + wn * (TN * 32) + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_base[(2 * ks + lh) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = b_base[(2 * ks + lh) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kn < nkb) store_tiles();
        __syncthreads();
        kb = kn;
    }

    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    // bias values of this lane's TN output columns, loaded once (a load inside the store loop is re-issued and waited for
    // per element: the stores may alias it)
    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    float bias_r[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + l31;
        bias_r[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= p.M) continue;
            size_t pix = m;
            if (g.phase_major) {
                int b, hp, wp;
                decode_m(g, m, b, hp, wp);
                pix = ((size_t)b * g.Hp + hp) * g.Wp + wp;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n < p.N) {
                    float v = acc[i][j][r];
                    v += bias_r[j];
                    p.C[pix * p.N + n] = fast_act(v, nslope, special, p.act, p.slope);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------- wgrad
// dW[co][j] += sum_m DY[pix(m)][co] * gather(X)[m][j],  j = (r,s,ci).  GEMM rows = co, cols = j, reduction = m.
template <int BM, int BN, int WM, int WN, int VW>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradArgs p) {
    constexpr int BK = 32;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int CPA = BM / 4, CPB = BN / 4;
    __shared__ __attribute__((aligned(16))) float smem[BK * LDA + BK * LDB];
    float *As = smem, *Bs = smem + BK * LDA;

    const Geom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk_mn);
    const int c0 = (tile / p.nblk_n) * BM, j0 = (tile % p.nblk_n) * BN;
    const int m_begin = blockIdx.y * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);

    // this thread's fixed column of the gathered operand
    const int jcol = j0 + (tid % CPB) * 4;
    int tr[4], ts[4], tc[4];  // tap decode per element (VW==1) or element 0 only (VW==4)
#pragma unroll
    for (int e = 0; e < (VW == 4 ? 1 : 4); ++e) {
        const int j = jcol + e;
        const int rs = j / g.Cg;
        tc[e] = j - rs * g.Cg;
        tr[e] = rs / g.S;
        ts[e] = rs - tr[e] * g.S;
    }
    const bool uniform_tap = g.tile_skip && (g.Cg % BN == 0);
    const int tile_rs = j0 / g.Cg;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[RA], rb[RB];

    auto alive = [&](int mb) {
        if (!uniform_tap) return true;
        int b, hp, wp;
        decode_m(g, mb, b, hp, wp);
        return tap_alive(g, hp, wp, tile_rs);
    };
    auto next_mb = [&](int mb) {
        while (mb < m_end && !alive(mb)) mb += BK;
        return mb;
    };

    auto load_tiles = [&](int mb) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int idx = tid + 256 * i;
            const int m = mb + idx / CPA, co = c0 + (idx % CPA) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && co < p.Co) {
                size_t pix = m;
                if (g.phase_major) {
                    int b, hp, wp;
                    decode_m(g, m, b, hp, wp);
                    pix = ((size_t)b * g.Hp + hp) * g.Wp + wp;
                }
                const float *src = p.DY + pix * p.Co + co;
                if ((p.Co & 3) == 0) {
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    v.x = src[0];
                    if (co + 1 < p.Co) v.y = src[1];
                    if (co + 2 < p.Co) v.z = src[2];
                    if (co + 3 < p.Co) v.w = src[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int idx = tid + 256 * i;
            const int m = mb + idx / CPB;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                int b, hp, wp;
                decode_m(g, m, b, hp, wp);
                const int pbase = b * g.Hg, bh = row_base(g, hp), bw = row_base(g, wp);
                if (VW == 4) {
                    if (jcol < p.K) {
                        const int hg = gcoord(g, bh, tr[0], g.Hg), wg = gcoord(g, bw, ts[0], g.Wg);
                        if (hg >= 0 && wg >= 0)
                            v = *reinterpret_cast<const float4 *>(p.X + ((size_t)(pbase + hg) * g.Wg + wg) * g.Cg + tc[0]);
                    }
                } else {
                    float e4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (jcol + e < p.K) {
                            const int hg = gcoord(g, bh, tr[e], g.Hg), wg = gcoord(g, bw, ts[e], g.Wg);
                            if (hg >= 0 && wg >= 0) e4[e] = p.X[((size_t)(pbase + hg) * g.Wg + wg) * g.Cg + tc[e]];
                        }
                    }
                    v = make_float4(e4[0], e4[1], e4[2], e4[3]);
                }
            }
            rb[i] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4 *>(&As[(idx / CPA) * LDA + (idx % CPA) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4 *>(&Bs[(idx / CPB) * LDB + (idx % CPB) * 4]) = rb[i];
        }
    };

    int mb = next_mb(m_begin);
    if (mb < m_end) {
        load_tiles(mb);
        store_tiles();
    }
    __syncthreads();
    while (mb < m_end) {
        const int mn = next_mb(mb + BK);
        if (mn < m_end) load_tiles(mn);
        const float *a_base = As + wm * (TM * 32) + l31;
        const float *b_base = Bs + wn * (TN * 32) + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_base[(2 * ks + lh) * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = b_base[(2 * ks + lh) * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (mn < m_end) store_tiles();
        __syncthreads();
        mb = mn;
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = c0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co >= p.Co) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int jj = j0 + wn * (TN * 32) + j * 32 + l31;
                if (jj < p.K) atomicAdd(&p.DW[(size_t)co * p.K + jj], acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------------- host side
Geom make_geom(const hoig_conv_desc *d, bool on_output_grid, bool gather_is_T) {
    // on_output_grid: m enumerates the conv OUTPUT pixels (fwd, wgrad) and the gathered tensor is x;
    // otherwise m enumerates the conv INPUT pixels (dgrad) and the gathered tensor is dy.
    Geom g;
    g.Bn = d->B;
    if (on_output_grid) {
        g.Hg = d->Hi; g.Wg = d->Wi; g.Cg = d->Ci; g.Hp = d->Ho; g.Wp = d->Wo;
    } else {
        g.Hg = d->Ho; g.Wg = d->Wo; g.Cg = d->Co; g.Hp = d->Hi; g.Wp = d->Wi;
    }
    g.R = d->R; g.S = d->S; g.stride = d->stride; g.pad = d->pad;
    g.gatherT = gather_is_T ? 1 : 0;
    g.phase_major = 0;
    g.tile_skip = 0;
    return g;
}

void enable_phase_major(Geom &g, int BMrows, bool kblock_in_one_tap) {
    if (!(g.gatherT && g.stride == 2 && (g.Hp % 2 == 0) && (g.Wp % 2 == 0))) return;
    g.phase_major = 1;
    const long per_phase = (long)g.Bn * (g.Hp / 2) * (g.Wp / 2);
    g.tile_skip = (per_phase % BMrows == 0 && kblock_in_one_tap) ? 1 : 0;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmArgs a, bool ncontig, hipStream_t st) {
    const int nbm = (int)hoig_cdiv(a.M, BM), nbn = (int)hoig_cdiv(a.N, BN);
    a.nblk_n = nbn;
    a.nblk = nbm * nbn;
    enable_phase_major(a.g, BM, a.g.Cg % 32 == 0);
    const bool vec = (a.g.Cg % 4 == 0);
    dim3 grid(a.nblk), block(256);
    if (vec) {
        if (ncontig) igemm_f32_kernel<BM, BN, WM, WN, 4, true><<<grid, block, 0, st>>>(a);
        else igemm_f32_kernel<BM, BN, WM, WN, 4, false><<<grid, block, 0, st>>>(a);
    } else {
        if (ncontig) igemm_f32_kernel<BM, BN, WM, WN, 1, true><<<grid, block, 0, st>>>(a);
        else igemm_f32_kernel<BM, BN, WM, WN, 1, false><<<grid, block, 0, st>>>(a);
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

int dispatch_igemm(IgemmArgs a, bool ncontig, hipStream_t st) {
    // pick the tile so the launch has >= ~2 workgroups per CU where the problem allows it
    const long t128 = hoig_cdiv(a.M, 128);
    if (a.N <= 32) return launch_igemm<128, 32, 4, 1>(a, ncontig, st);
    if (a.N <= 64) {
        if (t128 >= 512) return launch_igemm<128, 64, 2, 2>(a, ncontig, st);
        return launch_igemm<64, 64, 2, 2>(a, ncontig, st);
    }
    const long n128 = hoig_cdiv(a.N, 128);
    if (t128 * n128 >= 512) return launch_igemm<128, 128, 2, 2>(a, ncontig, st);
    return launch_igemm<64, 128, 2, 2>(a, ncontig, st);
}

int check_desc(const hoig_conv_desc *d) {
    if (!d || d->B <= 0 || d->Ci <= 0 || d->Co <= 0 || d->R <= 0 || d->S <= 0 || d->stride <= 0) return HOIG_EINVAL;
    if (d->precision != HOIG_PREC_F32 && d->precision != HOIG_PREC_BF16X3 && d->precision != HOIG_PREC_BF16 &&
        d->precision != HOIG_PREC_F16X2 && d->precision != HOIG_PREC_F16F6)
        return HOIG_EINVAL;
    if (!d->transposed) {
        if ((d->Hi + 2 * d->pad - d->R) / d->stride + 1 != d->Ho) return HOIG_EINVAL;
        if ((d->Wi + 2 * d->pad - d->S) / d->stride + 1 != d->Wo) return HOIG_EINVAL;
    } else {
        const int ho_min = (d->Hi - 1) * d->stride - 2 * d->pad + d->R;
        if (d->Ho < ho_min || d->Ho >= ho_min + d->stride) return HOIG_EINVAL;
        const int wo_min = (d->Wi - 1) * d->stride - 2 * d->pad + d->S;
        if (d->Wo < wo_min || d->Wo >= wo_min + d->stride) return HOIG_EINVAL;
    }
    return HOIG_OK;
}

}  // namespace

int hoig_conv_bf16_fwd_like(const hoig_conv_desc *d, const float *a, const float *w, const float *bias, float *c,
                            bool dgrad, hipStream_t st);  // conv_igemm_bf16.hip
int hoig_conv_small_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                        hipStream_t st);                  // conv_small.hip
int hoig_conv_small_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, hipStream_t st);
int hoig_conv_small_dgrad(const hoig_conv_desc *d, const float *dy, const float *w, float *dx, hipStream_t st);
int hoig_conv_small_ci_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                           hipStream_t st);
int hoig_conv_dot_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, hipStream_t st);
int hoig_conv_small_fwd_acts(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                             unsigned long long acts, hipStream_t st);
int hoig_conv_head7_m16(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, unsigned long long acts,
                        hipStream_t st);                  // conv_head16.hip
// conv_thin.hip: stride-1 'same' convolutions with <= 8 (3x3: 16) channels on one side, taps in place of the missing channels
int hoig_conv_thin_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, hipStream_t st, float *stats = nullptr);
int hoig_conv_thin_dgrad(const hoig_conv_desc *d, const float *dy, const float *w, float *dx, int accumulate, hipStream_t st);
int hoig_conv_thin_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, hipStream_t st);
int hoig_conv_thin_out(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                       unsigned long long acts, int dgrad, hipStream_t st);
static unsigned long long uniform_acts(int act) {
    unsigned long long a = 0;
    for (int f = 0; f < 16; ++f) a |= (unsigned long long)(act & 15) << (4 * f);
    return a;
}
static bool thin_enabled() {
    constexpr bool on = true;
    return on;
}

extern "C" int hoig_conv2d_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                               hoig_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return HOIG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    rc = hoig_conv_head7_m16(d, x, w, bias, y, uniform_acts(d->act), st);      // 7x7 heads, three-term forward: MFMA, taps as columns
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_small_fwd(d, x, w, bias, y, st);    // ... exact fp32 / other shapes: direct fp32 kernel
    if (rc == HOIG_EUNSUPPORTED && thin_enabled()) rc = hoig_conv_thin_out(d, x, w, bias, y, uniform_acts(d->act), 0, st);   // 3x3, <= 16 outputs
    if (rc == HOIG_EUNSUPPORTED && thin_enabled()) rc = hoig_conv_thin_fwd(d, x, w, bias, y, st);      // thin-input convs on MFMA
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_small_ci_fwd(d, x, w, bias, y, st);    // 7x7 stems with <= 8 input channels
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_dot_fwd(d, x, w, bias, y, st);         // <= 4 outputs over >= 1024 products
    if (rc != HOIG_EUNSUPPORTED) return rc;
    if (d->precision != HOIG_PREC_F32) {
        rc = hoig_conv_bf16_fwd_like(d, x, w, bias, y, false, st);
        if (rc != HOIG_EUNSUPPORTED) return rc;
    }
    IgemmArgs a;
    a.A = x; a.W = w; a.bias = bias; a.C = y;
    a.g = make_geom(d, true, d->transposed != 0);
    a.M = d->B * d->Ho * d->Wo; a.N = d->Co; a.K = d->R * d->S * d->Ci;
    a.act = d->act; a.slope = d->slope;
    return dispatch_igemm(a, false, st);
}

// hoig_conv2d_fwd + the per-image channel sums of y for the instance norm that follows (include/hoig_kernels.h), for the layers
// hoig_conv2d_fwd_packed_stats does not reach: the thin-INPUT convolutions (the 7x7 stems: 3 / 8 -> 64 channels at full resolution,
// where the statistics pass re-reads the largest tensor of the network).  HOIG_EUNSUPPORTED otherwise.
extern "C" int hoig_conv2d_fwd_stats(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, float *stats,
                                     hoig_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y || !stats) return HOIG_EINVAL;
    if (!thin_enabled()) return HOIG_EUNSUPPORTED;
    return hoig_conv_thin_fwd(d, x, w, bias, y, (hipStream_t)stream, stats);
}

// forward of a convolution with <= 16 output channels and a different activation per output channel (the generator's fused
// image / mask heads: tanh | sigmoid | none): `acts` holds the HOIG_ACT_* code of channel f in bits [4f, 4f+4)
extern "C" int hoig_conv2d_fwd_heads(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                                     uint64_t acts, hoig_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return HOIG_EINVAL;
    rc = hoig_conv_head7_m16(d, x, w, bias, y, acts, (hipStream_t)stream);               // 7x7, 3-5 outputs, three-term forward: MFMA
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_small_fwd_acts(d, x, w, bias, y, acts, (hipStream_t)stream);     // 7x7, <= 5 outputs: fp32 VALU
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_thin_out(d, x, w, bias, y, acts, 0, (hipStream_t)stream);
    return rc;
}

extern "C" int hoig_conv2d_bwd_data(const hoig_conv_desc *d, const float *dy, const float *w, float *dx,
                                    hoig_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w || !dx) return HOIG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    rc = thin_enabled() ? hoig_conv_thin_dgrad(d, dy, w, dx, 0, st) : HOIG_EUNSUPPORTED;     // thin-output convs (heads) on MFMA
    if (rc == HOIG_EUNSUPPORTED && thin_enabled()) rc = hoig_conv_thin_out(d, dy, w, nullptr, dx, 0, 1, st);   // thin-INPUT convs (VGG conv1_1)
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_small_dgrad(d, dy, w, dx, st);     // ... or the direct fp32 kernel (HOIG_PREC_F32)
    if (rc != HOIG_EUNSUPPORTED) return rc;
    if (d->precision != HOIG_PREC_F32) {
        rc = hoig_conv_bf16_fwd_like(d, dy, w, nullptr, dx, true, st);
        if (rc != HOIG_EUNSUPPORTED) return rc;
    }
    IgemmArgs a;
    a.A = dy; a.W = w; a.bias = nullptr; a.C = dx;
    a.g = make_geom(d, false, d->transposed == 0);
    a.M = d->B * d->Hi * d->Wi; a.N = d->Ci; a.K = d->R * d->S * d->Co;
    a.act = HOIG_ACT_NONE; a.slope = 0.f;
    return dispatch_igemm(a, true, st);
}

template <int BM, int BN, int WM, int WN>
static int launch_wgrad(WgradArgs a, hipStream_t st) {
    const int nbm = (int)hoig_cdiv(a.Co, BM), nbn = (int)hoig_cdiv(a.K, BN);
    a.nblk_n = nbn;
    a.nblk_mn = nbm * nbn;
    enable_phase_major(a.g, 32, true);
    // split the pixel reduction so the launch has ~1024 workgroups; chunks are multiples of 32 rows and, in
    // phase-major order, never straddle a parity class boundary badly (a straddling 32-row block is still correct:
    // dead taps are then filtered per element).
    int splits = (int)hoig_cdiv(1024, a.nblk_mn);
    // (tiny-K layers -- SPADE's 3->128 3x3 over 32x32 maps -- are launch-shaped: 64 pixels per split give them 128 workgroups)
    const int max_splits = (int)hoig_cdiv(a.M, a.K <= 128 ? 64 : 256);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mps = (int)hoig_cdiv(hoig_cdiv(a.M, splits), 32) * 32;
    a.m_per_split = mps;
    splits = (int)hoig_cdiv(a.M, mps);
    dim3 grid(a.nblk_mn, splits), block(256);
    if (a.g.Cg % 4 == 0) wgrad_f32_kernel<BM, BN, WM, WN, 4><<<grid, block, 0, st>>>(a);
    else wgrad_f32_kernel<BM, BN, WM, WN, 1><<<grid, block, 0, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

int hoig_conv_bf16_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, float *dbias, hipStream_t st);
bool hoig_conv_bf16_wgrad_fuses_bias(const hoig_conv_desc *d);

bool hoig_conv_thin_wgrad_applies(const hoig_conv_desc *d);       // conv_thin.hip
// bytes of per-stream scratch (hoig_stream_scratch_set) hoig_conv2d_bwd_weight wants for this layer: 0 for all but the thin-channel ones
extern "C" int64_t hoig_conv2d_bwd_weight_scratch_bytes(const hoig_conv_desc *d) {
    if (!d || check_desc(d)) return 0;
    return (thin_enabled() && hoig_conv_thin_wgrad_applies(d)) ? hoig_stream_scratch_bytes() : 0;
}

extern "C" int hoig_conv2d_bwd_weight(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, float *dbias,
                                      hoig_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !dy || !dw) return HOIG_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool fused_bias = dbias && hoig_conv_bf16_wgrad_fuses_bias(d);   // the halo wgrad kernel sums dy as it stages it
    if (dbias && !fused_bias) {
        rc = hoig_colsum_accum(dy, dbias, (int64_t)d->B * d->Ho * d->Wo, d->Co, stream);
        if (rc) return rc;
    }
    rc = thin_enabled() ? hoig_conv_thin_wgrad(d, x, dy, dw, st) : HOIG_EUNSUPPORTED;      // thin-input / thin-output convs on MFMA
    if (rc == HOIG_EUNSUPPORTED) rc = hoig_conv_small_wgrad(d, x, dy, dw, st);
    if (rc != HOIG_EUNSUPPORTED) return rc;
    if (d->precision != HOIG_PREC_F32) {
        rc = hoig_conv_bf16_wgrad(d, x, dy, dw, fused_bias ? dbias : nullptr, st);
        if (rc != HOIG_EUNSUPPORTED) return rc;
    }
    WgradArgs a;
    a.X = x; a.DY = dy; a.DW = dw;
    a.g = make_geom(d, true, d->transposed != 0);
    a.M = d->B * d->Ho * d->Wo; a.Co = d->Co; a.K = d->R * d->S * d->Ci;
    if (a.Co <= 32) {
        if (a.K <= 64) return launch_wgrad<32, 64, 1, 2>(a, st);
        return launch_wgrad<32, 128, 1, 4>(a, st);
    }
    if (a.Co <= 64) return launch_wgrad<64, 128, 2, 2>(a, st);
    if (a.K <= 64) return launch_wgrad<128, 64, 2, 2>(a, st);
    return launch_wgrad<128, 128, 2, 2>(a, st);
}
