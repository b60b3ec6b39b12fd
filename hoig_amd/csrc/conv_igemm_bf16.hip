// bf16-operand MFMA variants of the implicit-GEMM convolution (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   HOIG_PREC_BF16X3 : every fp32 operand x is split x = hi + lo (both bf16, hi = rne(x), lo = rne(x - hi));
//                      a*b ~= ah*bh + ah*bl + al*bh  -> 3 MFMAs per k-step, ~2^-16 relative product error, i.e. fp32-class
//                      parity (north_star bound 1e-3) at 1/3 of the 2.5 PFLOP/s dense bf16 rate (5.3x the fp32 MFMA rate).
//   HOIG_PREC_BF16   : hi only, 1 MFMA per k-step (for experiments; ~2^-9 relative per operand).
//
// Operands: activations stay fp32 NHWC in HBM and are split while the gathered tile is staged into LDS; weights are
// pre-split ONCE per optimiser step into K-contiguous bf16 planes by hoig_pack_conv_weight_bf16 ([N][K], K=(r,s,c);
// for the data gradient the plane is the transposed pack [Ci][R][S][Co]), so the B tile needs no conversion and both
// MFMA fragments are plain ds_read_b128 of 8 consecutive k.
// LDS image per operand plane: rows of 32 bf16 (64 B = four 16-B chunks), chunk index XOR-ed with (row>>2)&3 so the
// 16-lane groups of ds_read_b128 hit 16 distinct slots of the 256-B bank row.
// Shapes outside the fast path (gathered channels not a multiple of 32) return HOIG_EUNSUPPORTED and the caller uses the
// exact-fp32 kernels of conv_igemm.hip.
#include "common.h"

namespace {

struct Geom {
    int Bn, Hg, Wg, Cg;
    int Hp, Wp;
    int R, S, stride, pad;
    int gatherT, phase_major, tile_skip;
};

struct Args {
    const float *A;
    const unsigned short *Wh;
    const unsigned short *Wl;
    const float *bias;
    float *C;
    Geom g;
    int M, N, K;
    int act;
    float slope;
    int nblk_n, nblk;
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
    const int r = rs / g.S, s = rs - r * g.S;
    return (((hp + g.pad - r) & 1) == 0) && (((wp + g.pad - s) & 1) == 0);
}

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    return make_uint2((unsigned)hoig_f2bf(a) | ((unsigned)hoig_f2bf(b) << 16),
                      (unsigned)hoig_f2bf(c) | ((unsigned)hoig_f2bf(d) << 16));
}
__device__ __forceinline__ float resid(float x) { return x - hoig_bf2f(hoig_f2bf(x)); }

// byte offset of (row, k) inside one [rows][32] bf16 plane, k a multiple of 4
__device__ __forceinline__ int lds_off(int row, int k) {
    return row * 64 + ((((k >> 3) ^ ((row >> 2) & 3))) << 4) + ((k & 4) << 1);
}

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const Args p) {
    constexpr int BK = 32;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int RA = BM / 32;             // float4 gathers per thread
    constexpr int RB = BN / 64;             // 16-B weight chunks per thread per plane
    constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (PLANE_A + PLANE_B)];
    unsigned char *Ah = smem, *Al = smem + PLANE_A;
    unsigned char *Bh = smem + NS * PLANE_A, *Bl = Bh + PLANE_B;

    const Geom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (tile / p.nblk_n) * BM, n0 = (tile % p.nblk_n) * BN;

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
    const int brow = tid >> 2, bchunk = tid & 3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[RA];
    uint4 rbh[RB], rbl[RB];
    const int nkb = p.K / BK;

    auto next_kb = [&](int kb) {
        if (g.tile_skip)
            while (kb < nkb && !tap_alive(g, t_hp, t_wp, (kb * BK) / g.Cg)) ++kb;
        return kb;
    };
    auto load_tiles = [&](int kb) {
        // one tap per k-block (Cg % 32 == 0)
        const int k = kb * BK;
        const int rs = k / g.Cg, c = k - rs * g.Cg + kc;
        const int r = rs / g.S, s = rs - r * g.S;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pb[i] >= 0) {
                const int hg = gcoord(g, bh[i], r, g.Hg), wg = gcoord(g, bw[i], s, g.Wg);
                if (hg >= 0 && wg >= 0)
                    v = *reinterpret_cast<const float4 *>(p.A + ((size_t)(pb[i] + hg) * g.Wg + wg) * g.Cg + c);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + brow + 64 * i;
            uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
            if (n < p.N) {
                const size_t off = (size_t)n * p.K + k + bchunk * 8;
                vh = *reinterpret_cast<const uint4 *>(p.Wh + off);
                if (NS == 2) vl = *reinterpret_cast<const uint4 *>(p.Wl + off);
            }
            rbh[i] = vh;
            rbl[i] = vl;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int off = lds_off(lrow + 32 * i, kc);
            const float4 v = ra[i];
            *reinterpret_cast<uint2 *>(Ah + off) = pack4(v.x, v.y, v.z, v.w);
            if (NS == 2) *reinterpret_cast<uint2 *>(Al + off) = pack4(resid(v.x), resid(v.y), resid(v.z), resid(v.w));
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = brow + 64 * i;
            const int off = row * 64 + ((bchunk ^ ((row >> 2) & 3)) << 4);
            *reinterpret_cast<uint4 *>(Bh + off) = rbh[i];
            if (NS == 2) *reinterpret_cast<uint4 *>(Bl + off) = rbl[i];
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
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[TM], al[TM], bhf[TN], blf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (TM * 32) + i * 32 + l31;
                const int off = row * 64 + (((2 * ks + lh) ^ ((row >> 2) & 3)) << 4);
                ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + off);
                if (NS == 2) al[i] = *reinterpret_cast<const bf16x8 *>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (TN * 32) + j * 32 + l31;
                const int off = row * 64 + (((2 * ks + lh) ^ ((row >> 2) & 3)) << 4);
                bhf[j] = *reinterpret_cast<const bf16x8 *>(Bh + off);
                if (NS == 2) blf[j] = *reinterpret_cast<const bf16x8 *>(Bl + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NS == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bhf[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], blf[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bhf[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kn < nkb) store_tiles();
        __syncthreads();
        kb = kn;
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
                    if (p.bias) v += p.bias[n];
                    p.C[pix * p.N + n] = hoig_act(v, p.act, p.slope);
                }
            }
        }
    }
}

// w: fp32 [Co][RS][Ci].  mode 0 -> planes [Co][RS][Ci] (forward), mode 1 -> planes [Ci][RS][Co] (data gradient)
__global__ void pack_weight_kernel(const float *__restrict__ w, int Co, int RS, int Ci, int mode,
                                   unsigned short *__restrict__ hi, unsigned short *__restrict__ lo) {
    const int64_t n = (int64_t)Co * RS * Ci;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t src = i;
        if (mode == 1) {   // i enumerates [ci][rs][co]
            const int co = (int)(i % Co);
            const int64_t t = i / Co;
            const int rs = (int)(t % RS), ci = (int)(t / RS);
            src = ((int64_t)co * RS + rs) * Ci + ci;
        }
        const float x = w[src];
        const unsigned short h = hoig_f2bf(x);
        hi[i] = h;
        if (lo) lo[i] = hoig_f2bf(x - hoig_bf2f(h));
    }
}

template <int BM, int BN, int WM, int WN>
int launch(Args a, int ns, hipStream_t st) {
    const int nbm = (int)hoig_cdiv(a.M, BM), nbn = (int)hoig_cdiv(a.N, BN);
    a.nblk_n = nbn;
    a.nblk = nbm * nbn;
    if (a.g.gatherT && a.g.stride == 2 && (a.g.Hp % 2 == 0) && (a.g.Wp % 2 == 0)) {
        a.g.phase_major = 1;
        const long per_phase = (long)a.g.Bn * (a.g.Hp / 2) * (a.g.Wp / 2);
        a.g.tile_skip = (per_phase % BM == 0) ? 1 : 0;
    }
    if (ns == 2) igemm_bf16_kernel<BM, BN, WM, WN, 2><<<a.nblk, 256, 0, st>>>(a);
    else igemm_bf16_kernel<BM, BN, WM, WN, 1><<<a.nblk, 256, 0, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

int run(const hoig_conv_desc *d, const float *a, const unsigned short *wh, const unsigned short *wl, const float *bias,
        float *c, bool dgrad, hipStream_t st) {
    Args p;
    p.A = a; p.Wh = wh; p.Wl = wl; p.bias = bias; p.C = c;
    Geom &g = p.g;
    g.Bn = d->B;
    if (!dgrad) {
        g.Hg = d->Hi; g.Wg = d->Wi; g.Cg = d->Ci; g.Hp = d->Ho; g.Wp = d->Wo;
        g.gatherT = d->transposed ? 1 : 0;
        p.M = d->B * d->Ho * d->Wo; p.N = d->Co; p.K = d->R * d->S * d->Ci;
        p.act = d->act; p.slope = d->slope;
    } else {
        g.Hg = d->Ho; g.Wg = d->Wo; g.Cg = d->Co; g.Hp = d->Hi; g.Wp = d->Wi;
        g.gatherT = d->transposed ? 0 : 1;
        p.M = d->B * d->Hi * d->Wi; p.N = d->Ci; p.K = d->R * d->S * d->Co;
        p.act = HOIG_ACT_NONE; p.slope = 0.f;
    }
    g.R = d->R; g.S = d->S; g.stride = d->stride; g.pad = d->pad;
    g.phase_major = 0; g.tile_skip = 0;
    if (g.Cg % 32 != 0) return HOIG_EUNSUPPORTED;
    if (g.gatherT && g.stride == 2 && ((g.Hp | g.Wp) & 1)) return HOIG_EUNSUPPORTED;
    const int ns = d->precision == HOIG_PREC_BF16X3 ? 2 : 1;
    if (ns == 2 && !wl) return HOIG_EINVAL;
    const long t128 = hoig_cdiv(p.M, 128);
    if (p.N <= 32) return HOIG_EUNSUPPORTED;
    if (p.N <= 64) {
        if (t128 >= 512) return launch<128, 64, 2, 2>(p, ns, st);
        return launch<64, 64, 2, 2>(p, ns, st);
    }
    const long n128 = hoig_cdiv(p.N, 128);
    if (t128 * n128 >= 512) return launch<128, 128, 2, 2>(p, ns, st);
    return launch<64, 128, 2, 2>(p, ns, st);
}

}  // namespace

// fp32-weight entry points cannot use the bf16 path (it needs the pre-split planes): tell the dispatcher to fall back.
int hoig_conv_bf16_fwd_like(const hoig_conv_desc *, const float *, const float *, const float *, float *, bool,
                            hipStream_t) {
    return HOIG_EUNSUPPORTED;
}
int hoig_conv_bf16_wgrad(const hoig_conv_desc *, const float *, const float *, float *, hipStream_t) {
    return HOIG_EUNSUPPORTED;
}

extern "C" int hoig_pack_conv_weight_bf16(const float *w, int Co, int RS, int Ci, int for_dgrad, uint16_t *hi,
                                          uint16_t *lo, hoig_stream_t stream) {
    if (!w || !hi || Co <= 0 || RS <= 0 || Ci <= 0) return HOIG_EINVAL;
    const int64_t n = (int64_t)Co * RS * Ci;
    pack_weight_kernel<<<hoig_stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, Co, RS, Ci, for_dgrad ? 1 : 0, hi, lo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_conv2d_fwd_packed(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint16_t *w_lo,
                                      const float *bias, float *y, hoig_stream_t stream) {
    if (!d || !x || !w_hi || !y) return HOIG_EINVAL;
    if (d->precision != HOIG_PREC_BF16X3 && d->precision != HOIG_PREC_BF16) return HOIG_EINVAL;
    return run(d, x, w_hi, w_lo, bias, y, false, (hipStream_t)stream);
}

extern "C" int hoig_conv2d_bwd_data_packed(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi,
                                           const uint16_t *wt_lo, float *dx, hoig_stream_t stream) {
    if (!d || !dy || !wt_hi || !dx) return HOIG_EINVAL;
    if (d->precision != HOIG_PREC_BF16X3 && d->precision != HOIG_PREC_BF16) return HOIG_EINVAL;
    return run(d, dy, wt_hi, wt_lo, nullptr, dx, true, (hipStream_t)stream);
}
