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
#include "conv_bf16_common.h"
#include "tuning.h"
#include <cstdlib>

// tools/wgrad_knockout.cpp builds this file with HOIG_WG_KO != 0 to time wgrad_halo_bf16_kernel with parts removed (results are
// then wrong): 1 no hi/lo split (bit moves only), 2 no global loads after the first tile, 4 no LDS stores after the first tile,
// 8 no atomic epilogue, 16 no LDS fragment reads (MFMAs on register garbage), 32 no MFMAs of the lo plane
#ifndef HOIG_WG_KO
#define HOIG_WG_KO 0
#endif
#ifndef HOIG_HALO_BSTAGES
#define HOIG_HALO_BSTAGES 1
#endif

namespace {
using namespace hoig_detail;


// LDS plane = [rows][BK] bf16; the 16-B chunk index of a row is XOR-ed with a row-dependent value so that the 16-lane
// groups of ds_read_b128 (rows r..r+3, r+12.., r+20..) fall on 16 distinct slots of the 256-B bank row:
//   BK = 32 (64-B rows, 4 chunks): chunk ^ ((row >> 2) & 3)      BK = 64 (128-B rows, 8 chunks): chunk ^ ((row >> 1) & 7)
template <int BK>
__device__ __forceinline__ int lds_swz(int row) {
    return BK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7);
}
// byte offset of (row, k), k a multiple of 4
template <int BK>
__device__ __forceinline__ int lds_off(int row, int k) {
    return row * (BK * 2) + (((k >> 3) ^ lds_swz<BK>(row)) << 4) + ((k & 4) << 1);
}

template <int BM, int BN, int WM, int WN, int NSX, int BK, bool F16>
__global__ __launch_bounds__(WM * WN * 64) void igemm_bf16_kernel(const Args p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int NT = WM * WN * 64;        // 256 or 512 threads
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int TPR_A = BK / 4, TPR_B = BK / 8;      // threads per tile row: float4 gathers / 16-B weight chunks
    constexpr int RA = BM * TPR_A / NT;     // float4 gathers per thread
    constexpr int RB = BN * TPR_B / NT;     // 16-B weight chunks per thread per plane
    constexpr int AROWS = NT / TPR_A, BROWS = NT / TPR_B;
    constexpr int PLANE_A = BM * BK * 2, PLANE_B = BN * BK * 2;
    constexpr int STAGE = NS * PLANE_A + NB * PLANE_B;
    // two LDS stages: k-block t is multiplied out of one while k-block t+1 is converted into the other -> ONE barrier
    // per k-block; 64 KB at 128x128 (2 workgroups per CU)
    // BK = 32: ONE stage (32 KB at 128x128) so that four or five workgroups share a CU and cover each other's waits
    // (same finding as for wgrad); BK = 64 keeps two stages
    constexpr int NSTAGE = BK == 32 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

    const Geom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (tile / p.nblk_n) * BM, n0 = (tile % p.nblk_n) * BN;

    const int kc = (tid % TPR_A) * 4, lrow = tid / TPR_A;
    int pb[RA], bh[RA], bw[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + AROWS * i;
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
    const int brow = tid / TPR_B, bchunk = tid % TPR_B;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[RA];
    uint4 rbh[RB], rbl[RB];
    // K is walked tap by tap ((r,s) outer, 32-channel blocks inner).  Everything that depends only on the tap -- the
    // validity and the address of each gathered pixel row -- is computed once per tap, so the per-k-block VALU work is
    // one pointer add per load plus the bf16 split (MFMA and VALU of ONE wave serialise; this keeps the matrix pipe fed).
    const int cpb = g.Cg / BK, RS = g.R * g.S;
    const float *aptr[RA];
    const unsigned short *wrow_h[RB], *wrow_l[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + brow + BROWS * i;
        const size_t o = plane_index(n, bchunk * 8, p.K);
        wrow_h[i] = n < p.N ? p.Wh + o : nullptr;
        wrow_l[i] = (NB == 2 && n < p.N) ? p.Wl + o : nullptr;
    }
    int aoff[RA], boff[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) aoff[i] = lds_off<BK>(lrow + AROWS * i, kc);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = brow + BROWS * i;
        boff[i] = row * (BK * 2) + ((bchunk ^ lds_swz<BK>(row)) << 4);
    }
    // split-K: this workgroup multiplies k-blocks [begin, begin + steps_per_split) of the (tap, channel-block) walk
    const int begin = p.ksplit > 1 ? (int)blockIdx.y * p.steps_per_split : 0;
    int steps_left = p.ksplit > 1 ? min(p.steps_per_split, RS * cpb - begin) : 0x7fffffff;
    int rs = begin / cpb - 1, cb = cpb - 1, wk = 0;
    int cb_next_tap = begin - (begin / cpb) * cpb;      // channel block to start the first tap at
    auto advance = [&]() -> bool {      // move (rs, cb) to the next live k-block; false when K is exhausted
        if (steps_left-- <= 0) return false;
        if (++cb < cpb) return true;
        cb = cb_next_tap;
        cb_next_tap = 0;
        do {
            ++rs;
        } while (rs < RS && g.tile_skip && !tap_alive(g, t_hp, t_wp, rs));
        if (rs >= RS) return false;
        const int r = rs / g.S, s_ = rs - r * g.S;
        wk = rs * g.Cg;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            aptr[i] = nullptr;
            if (pb[i] >= 0) {
                const int hg = gcoord(g, bh[i], r, g.Hg), wg = gcoord(g, bw[i], s_, g.Wg);
                if (hg >= 0 && wg >= 0) aptr[i] = p.A + ((size_t)(pb[i] + hg) * g.Wg + wg) * g.Cg + kc;
            }
        }
        return true;
    };
    auto load_tiles = [&]() {
        const int c = cb * BK;
#pragma unroll
        for (int i = 0; i < RA; ++i)
            ra[i] = aptr[i] ? *reinterpret_cast<const float4 *>(aptr[i] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            rbh[i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + (size_t)(wk + c) * 32) : make_uint4(0, 0, 0, 0);
            if (NB == 2)
                rbl[i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + (size_t)(wk + c) * 32) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](int stage) {
        unsigned char *Ah = smem + stage * STAGE, *Al = Ah + PLANE_A;
        unsigned char *Bh = Ah + NS * PLANE_A, *Bl = Bh + PLANE_B;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            uint2 hi, lo;
            split4t<F16>(ra[i], hi, lo);
            *reinterpret_cast<uint2 *>(Ah + aoff[i]) = hi;
            if (NS == 2) *reinterpret_cast<uint2 *>(Al + aoff[i]) = lo;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            *reinterpret_cast<uint4 *>(Bh + boff[i]) = rbh[i];
            if (NB == 2) *reinterpret_cast<uint4 *>(Bl + boff[i]) = rbl[i];
        }
    };
    int aread[TM], bread[TN];       // ds_read_b128 offsets of this lane's fragments for ks = 0 (ks = 1: chunk ^ 2)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + l31;
        aread[i] = row * (BK * 2) + ((lh ^ lds_swz<BK>(row)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + l31;
        bread[j] = row * (BK * 2) + ((lh ^ lds_swz<BK>(row)) << 4);
    }

    bool more = advance();
    if (more) {
        load_tiles();
        store_tiles(0);
    }
    __syncthreads();
    int cur = 0;
    while (more) {
        const bool nxt = advance();
        if (nxt) load_tiles();
        const unsigned char *Ah = smem + cur * STAGE, *Al = Ah + PLANE_A;
        const unsigned char *Bh = Ah + NS * PLANE_A, *Bl = Bh + PLANE_B;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 ah[TM], al[TM], bhf[TN], blf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = aread[i] ^ (ks << 5);
                ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + off);
                if (NS == 2) al[i] = *reinterpret_cast<const bf16x8 *>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = bread[j] ^ (ks << 5);
                bhf[j] = *reinterpret_cast<const bf16x8 *>(Bh + off);
                if (NB == 2) blf[j] = *reinterpret_cast<const bf16x8 *>(Bl + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NS == 2)
                        acc[i][j] = mfma16<F16>(al[i], bhf[j], acc[i][j]);
                        if (NB == 2) acc[i][j] = mfma16<F16>(ah[i], blf[j], acc[i][j]);
                    acc[i][j] = mfma16<F16>(ah[i], bhf[j], acc[i][j]);
                }
        }
        if (NSTAGE == 2) {
            if (nxt) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();
            if (nxt) store_tiles(0);
            __syncthreads();
        }
        more = nxt;
    }

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
                    float v = acc[i][j][r] * p.oscale;
                    if (p.ksplit > 1) {
                        if (blockIdx.y == 0) v += bias_r[j];
                        atomicAdd(&p.C[pix * p.N + n], v);
                    } else {
                        v += bias_r[j];
                        p.C[pix * p.N + n] = fast_act(v, nslope, special, p.act, p.slope);
                    }
                }
            }
        }
    }
}

// forward planes: fp16 split of w * 2^8;   data-gradient planes: bf16 split of w
__device__ __forceinline__ void split_weight(float x, bool f16, unsigned short &h, unsigned short &l) {
    if (f16) {
        const float xs = x * W_SCALE_F16;
        const _Float16 hh = (_Float16)xs;
        const _Float16 ll = (_Float16)(xs - (float)hh);
        h = __builtin_bit_cast(unsigned short, hh);
        l = __builtin_bit_cast(unsigned short, ll);
    } else {
        h = hoig_f2bf(x);
        l = hoig_f2bf(x - hoig_bf2f(h));
    }
}

// w: fp32 [Co][RS][Ci].  mode 0 -> plane rows n = co, k = (rs, ci) (forward); mode 1 -> rows n = ci, k = (rs, co) (data
// gradient); both in the blocked plane layout (plane_index)
__global__ void pack_weight_kernel(const float *__restrict__ w, int Co, int RS, int Ci, int mode,
                                   unsigned short *__restrict__ hi, unsigned short *__restrict__ lo) {
    const int64_t n = (int64_t)Co * RS * Ci;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const int64_t t = i / Ci;
        const int rs = (int)(t % RS), co = (int)(t / RS);
        const size_t dst = mode == 1 ? plane_index(ci, rs * Co + co, RS * Co) : plane_index(co, rs * Ci + ci, RS * Ci);
        unsigned short h, l;
        split_weight(w[i], mode == 0, h, l);
        hi[dst] = h;
        if (lo) lo[dst] = l;
    }
}

// Every conv weight of a network in ONE launch: `segs` holds (offset, Co, RS, Ci, flags, first tile) per weight of the
// flat parameter buffer; the planes of a weight land at the weight's own offset in the plane buffers.  A workgroup owns one
// 32(co) x 32(ci) tile of one weight (binary search over the tile prefix) and walks its RS taps: rows are read coalesced
// along ci, the forward planes written in place, the data-gradient planes ([Ci][RS][Co]) written coalesced along co after a
// transpose through LDS.
__global__ __launch_bounds__(256) void pack_all_kernel(const float *__restrict__ flat, const int64_t *__restrict__ segs,
                                                       int nseg, unsigned short *__restrict__ hi_f,
                                                       unsigned short *__restrict__ lo_f,
                                                       unsigned short *__restrict__ hi_d,
                                                       unsigned short *__restrict__ lo_d) {
    int lo_s = 0, hi_s = nseg - 1;
    while (lo_s < hi_s) {                 // last segment whose first tile <= blockIdx.x
        const int mid = (lo_s + hi_s + 1) >> 1;
        if (segs[(int64_t)mid * 6 + 5] <= (int64_t)blockIdx.x) lo_s = mid; else hi_s = mid - 1;
    }
    const int64_t *sg = segs + (int64_t)lo_s * 6;
    const int64_t off = sg[0];
    const int Co = (int)sg[1], RS = (int)sg[2], Ci = (int)sg[3], flags = (int)sg[4];
    const int t = (int)((int64_t)blockIdx.x - sg[5]);
    const int tiles_ci = (Ci + 31) >> 5;
    const int co0 = (t / tiles_ci) * 32, ci0 = (t % tiles_ci) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    __shared__ unsigned int tile[32][33];
    const float *w = flat + off;
    for (int rs = 0; rs < RS; ++rs) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = co0 + ty + 8 * k, ci = ci0 + tx;
            unsigned int pk = 0;
            if (co < Co && ci < Ci) {
                const int64_t i = ((int64_t)co * RS + rs) * Ci + ci;
                const float x = w[i];
                unsigned short h, l;
                if (flags & 1) {
                    split_weight(x, true, h, l);
                    const size_t o = off + plane_index(co, rs * Ci + ci, RS * Ci);
                    hi_f[o] = h;
                    lo_f[o] = l;
                }
                split_weight(x, false, h, l);
                pk = (unsigned int)h | ((unsigned int)l << 16);
            }
            tile[ty + 8 * k][tx] = pk;
        }
        if (flags & 2) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = ci0 + ty + 8 * k, co = co0 + tx;
                if (co < Co && ci < Ci) {
                    const unsigned int pk = tile[tx][ty + 8 * k];
                    const size_t o = off + plane_index(ci, rs * Co + co, RS * Co);
                    hi_d[o] = (unsigned short)(pk & 0xffffu);
                    lo_d[o] = (unsigned short)(pk >> 16);
                }
            }
            __syncthreads();
        }
    }
}

// Adam and the operand split in ONE pass over the weights (hoig_adam_pack_step): the update of a convolution weight and its
// four 16-bit planes used to be two launches -- Adam over the flat buffer (28 B per parameter), then pack_all_kernel re-reading
// every updated weight (4 B) and writing the planes with 2-byte stores -- at the END of a step, where nothing runs beside them.
// Here a workgroup owns one 32(co) x 32(ci) tile of one weight as in pack_all_kernel; a thread updates FOUR consecutive input
// channels (float4 of p, g, m, v; the arithmetic of adam_dev_kernel, expression for expression) and writes their forward planes
// as 8-byte stores; the data-gradient planes go through the same LDS transpose, four output channels per thread.  Workgroups
// past the tile count do the plain update of the parameters that have no planes (`plain`: 1024-element chunks).
__global__ __launch_bounds__(256) void adam_pack_kernel(float *__restrict__ flat, const float *__restrict__ grad, float *__restrict__ m_,
                                                        float *__restrict__ v_, const float *__restrict__ derived, float gscale,
                                                        const int64_t *__restrict__ segs, int nseg, int64_t ntiles,
                                                        const int64_t *__restrict__ plain, unsigned short *__restrict__ hi_f,
                                                        unsigned short *__restrict__ lo_f, unsigned short *__restrict__ hi_d,
                                                        unsigned short *__restrict__ lo_d) {
    const float step_size = derived[0], omb1 = derived[1], b2 = derived[2], omb2 = derived[3], eps = derived[4], bc2_sqrt = derived[5];
    struct Q { float4 p, g, m, v; };
    auto load4 = [&](int64_t i) -> Q {
        Q q;
        q.p = *reinterpret_cast<const float4 *>(flat + i);
        q.g = *reinterpret_cast<const float4 *>(grad + i);
        q.m = *reinterpret_cast<const float4 *>(m_ + i);
        q.v = *reinterpret_cast<const float4 *>(v_ + i);
        return q;
    };
    auto update4 = [&](Q q, int64_t i) -> float4 {
        float *pe = &q.p.x, *me = &q.m.x, *ve = &q.v.x;
        const float *ge = &q.g.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] = me[k] + (gk - me[k]) * omb1;
            ve[k] = ve[k] * b2 + omb2 * (gk * gk);
            const float denom = sqrtf(ve[k]) / bc2_sqrt + eps;
            pe[k] = pe[k] - step_size * (me[k] / denom);
        }
        *reinterpret_cast<float4 *>(flat + i) = q.p;
        *reinterpret_cast<float4 *>(m_ + i) = q.m;
        *reinterpret_cast<float4 *>(v_ + i) = q.v;
        return q.p;
    };
    if ((int64_t)blockIdx.x >= ntiles) {
        const int64_t *c = plain + 2 * ((int64_t)blockIdx.x - ntiles);
        const int e = 4 * (int)threadIdx.x;
        if (e < (int)c[1]) update4(load4(c[0] + e), c[0] + e);
        return;
    }
    int lo_s = 0, hi_s = nseg - 1;
    while (lo_s < hi_s) {                 // last segment whose first tile <= blockIdx.x
        const int mid = (lo_s + hi_s + 1) >> 1;
        if (segs[(int64_t)mid * 6 + 5] <= (int64_t)blockIdx.x) lo_s = mid; else hi_s = mid - 1;
    }
    const int64_t *sg = segs + (int64_t)lo_s * 6;
    const int64_t off = sg[0];
    const int Co = (int)sg[1], RS = (int)sg[2], Ci = (int)sg[3], flags = (int)sg[4];
    const int t = (int)((int64_t)blockIdx.x - sg[5]);
    const int tiles_ci = Ci >> 5;
    const int co0 = (t / tiles_ci) * 32, ci0 = (t % tiles_ci) * 32;
    const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    __shared__ unsigned int tile[32][33];
    const int co = co0 + row, ci = ci0 + c4;
    const int64_t e0 = off + (int64_t)co * RS * Ci + ci;
    Q nxt = load4(e0);
    for (int rs = 0; rs < RS; ++rs) {
        const Q cur = nxt;
        if (rs + 1 < RS) nxt = load4(e0 + (int64_t)(rs + 1) * Ci);          // the next tap's rows are in flight while this one is split
        const float4 w4 = update4(cur, e0 + (int64_t)rs * Ci);
        const float we[4] = {w4.x, w4.y, w4.z, w4.w};
        unsigned short h[4], l[4];
        if (flags & 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) split_weight(we[k], true, h[k], l[k]);
            const size_t o = off + plane_index(co, rs * Ci + ci, RS * Ci);
            *reinterpret_cast<uint2 *>(hi_f + o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
            *reinterpret_cast<uint2 *>(lo_f + o) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
        }
        if (flags & 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                split_weight(we[k], false, h[k], l[k]);
                tile[row][c4 + k] = (unsigned int)h[k] | ((unsigned int)l[k] << 16);
            }
            __syncthreads();
            unsigned int pk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pk[k] = tile[c4 + k][row];          // (ci = ci0 + row, co = co0 + c4 + k)
            const size_t o = off + plane_index(ci0 + row, rs * Co + co0 + c4, RS * Co);
            *reinterpret_cast<uint2 *>(hi_d + o) = make_uint2((pk[0] & 0xffffu) | (pk[1] << 16), (pk[2] & 0xffffu) | (pk[3] << 16));
            *reinterpret_cast<uint2 *>(lo_d + o) = make_uint2((pk[0] >> 16) | (pk[1] & 0xffff0000u), (pk[2] >> 16) | (pk[3] & 0xffff0000u));
            __syncthreads();
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK = 32>
int launch(Args a, int ns, hipStream_t st) {
    constexpr int NT = WM * WN * 64;
    const int nbm = (int)hoig_cdiv(a.M, BM), nbn = (int)hoig_cdiv(a.N, BN);
    a.nblk_n = nbn;
    a.nblk = nbm * nbn;
    if (a.g.gatherT && a.g.stride == 2 && (a.g.Hp % 2 == 0) && (a.g.Wp % 2 == 0)) {
        a.g.phase_major = 1;
        const long per_phase = (long)a.g.Bn * (a.g.Hp / 2) * (a.g.Wp / 2);
        a.g.tile_skip = (per_phase % BM == 0) ? 1 : 0;
    }
    // few output tiles but a long K (the attention MLP: 128 outputs, K = 25*C): split K over blockIdx.y
    a.ksplit = 1;
    a.steps_per_split = 0;
    const int steps = (a.K / BK);
    constexpr int sk_target = 1024;      // (sweep 512 / 768 / 1024: 2.07 / 1.90 / 1.91 ms for the attention forward GEMMs)
    constexpr int sk_maxblk = 192;
    if (a.nblk < sk_maxblk && steps >= 32 && a.act == HOIG_ACT_NONE && !a.g.tile_skip) {
        int want = (int)hoig_cdiv(sk_target, a.nblk);
        if (want > steps / 8) want = steps / 8;
        if (want > 1) {
            a.steps_per_split = (int)hoig_cdiv(steps, want);
            a.ksplit = (int)hoig_cdiv(steps, a.steps_per_split);
            if (hipMemsetAsync(a.C, 0, (size_t)a.M * a.N * sizeof(float), st) != hipSuccess) return HOIG_ELAUNCH;
        }
    }
    dim3 grid(a.nblk, a.ksplit);
    if (a.f16) HOIG_NS_SWITCH(ns, igemm_bf16_kernel<BM, BN, WM, WN, NSX, BK, true><<<grid, NT, 0, st>>>(a));
    else HOIG_NS_SWITCH(ns, igemm_bf16_kernel<BM, BN, WM, WN, NSX, BK, false><<<grid, NT, 0, st>>>(a));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-1 convolutions with an LDS-resident INPUT HALO tile.  The generic kernel above re-gathers the A tile from
// L2 for every tap (9x for a 3x3) and is bound by the per-CU load path (~60-70 GB/s from L2), not by the matrix
// pipe.  Here a workgroup owns 4 rows x 32 columns of output pixels (BM = 128); for each 32-channel block it stages the
// (4+KS-1) x (32+KS-1) input halo ONCE (split to bf16 hi/lo) and all KS*KS taps read their A fragments out of that same
// LDS image at a tap-dependent, lane-uniform row offset -- only the weight tile streams per tap (double-buffered).
// Halo rows are 80 B apart (64 B of data + 16 B pad): any 16 consecutive rows fall on 16 distinct 16-B bank slots, so
// ds_read_b128 stays conflict-free at every tap shift without an address-dependent swizzle.
// Used for conv fwd (stride 1) and for the data gradient of stride-1 convs (taps walked flipped).

// Instance-norm statistics from a convolution's epilogue.  s1 / s2: the lane's sums of v and v*v over the pixels it stored, per
// 32-channel column group j (channel = n0 + wn * TN * 32 + j * 32 + (lane & 31)).  The two 32-lane halves hold different pixels of
// the same channels (one cross-lane add), the WM waves with the same wn different image rows (LDS, which is dead after the main
// loop), and the workgroup then issues ONE atomic per (channel, moment) into the image's accumulators -- as many per address as
// the streaming statistics kernel it replaces issued (norm.hip: one per 16 pixel rows).
template <int TN, int WM, int WN, int NT>
__device__ __forceinline__ void halo_stats_epilogue(float (&s1)[TN], float (&s2)[TN], unsigned char *lds, float *stats_img, int N,
                                                    int n0, int wm, int wn, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        s1[j] += __shfl_xor(s1[j], 32);
        s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();                                   // every wave has left the main loop's LDS tiles
    float *red = reinterpret_cast<float *>(lds);       // [WM][WN][TN][2][32]
    if (lh == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            red[(((wm * WN + wn) * TN + j) * 2 + 0) * 32 + l31] = s1[j];
            red[(((wm * WN + wn) * TN + j) * 2 + 1) * 32 + l31] = s2[j];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < WN * TN * 64; e += NT) {
        const int l = e & 31, m = (e >> 5) & 1, j = (e >> 6) % TN, w = (e >> 6) / TN;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < WM; ++k) v += red[(((k * WN + w) * TN + j) * 2 + m) * 32 + l];
        const int n = n0 + w * (TN * 32) + j * 32 + l;
        if (n < N) atomicAdd(&stats_img[(size_t)m * N + n], v);
    }
}

template <int KS, int NSX, int WN, int BN, bool F16>
__global__ __launch_bounds__(128 * WN) void conv_halo_bf16_kernel(const HaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int TH = 4, TW = 32;
    constexpr int NT = 128 * WN;                           // 2 x WN waves: 256 or 512 threads
    constexpr int RB = BN * 4 / NT;                        // 16-B weight chunks per thread per plane
    constexpr int HH = TH + KS - 1, HW = TW + KS - 1, HPIX = HH * HW;
    constexpr int AROW = 80;                               // bytes per halo pixel row (32 bf16 + pad)
    constexpr int PLANE_A = HPIX * AROW, PLANE_B = BN * 64;
    constexpr int TM = 2, TN = BN / (32 * WN);
    static_assert(TN >= 1, "BN = 64 needs the 4-wave variant");
    // LDS: one halo stage + two weight stages = 64 KB at KS=3 -> two workgroups per CU
    constexpr int NBST = HOIG_HALO_BSTAGES;                // weight stages in LDS
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * PLANE_A + NBST * NB * PLANE_B];
    unsigned char *Ah = smem, *Al = smem + PLANE_A;
    unsigned char *Bst = smem + NS * PLANE_A;              // two stages of (Bh, Bl)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    int mt = tile / p.nblk_n;
    const int n0 = (tile % p.nblk_n) * BN;
    const int tx_ = mt % p.tiles_x;
    mt /= p.tiles_x;
    const int ty_ = mt % p.tiles_y, b = mt / p.tiles_y;
    const int y0 = ty_ * TH, x0 = tx_ * TW;                // tile origin (output pixels)

    const int brow = tid >> 2, bchunk = tid & 3;
    const unsigned short *wrow_h[RB], *wrow_l[RB];
    int boff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + brow + (NT / 4) * i;
        const size_t o = plane_index(n, bchunk * 8, p.K);
        wrow_h[i] = n < p.N ? p.Wh + o : nullptr;
        wrow_l[i] = (NB == 2 && n < p.N) ? p.Wl + o : nullptr;
        const int row = brow + (NT / 4) * i;
        boff[i] = row * 64 + ((bchunk ^ ((row >> 2) & 3)) << 4);
    }
    // fragment read offsets: wave wm owns output rows 2*wm, 2*wm+1 of the tile (32 pixels each = one MFMA row tile)
    int aread[TM], bread[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aread[i] = ((wm * TM + i) * HW + l31) * AROW + lh * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + l31;
        bread[j] = row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Weight tiles are prefetched TWO steps ahead (step = one tap of one 32-channel block) into two register sets: the
    // weights of a 512x512x3x3 layer (9.4 MB of bf16 planes) live in the Infinity Cache, whose latency under load exceeds
    // one step's MFMA phase (768 cycles per wave).
    uint4 rb0h[RB], rb0l[RB], rb1h[RB], rb1l[RB];
    constexpr int KK = KS * KS;
    const int ncb = p.Cg >> 5, T = ncb * KK;
    auto load_b = [&](int step, uint4 (&rh)[RB], uint4 (&rl)[RB]) {
        const int cb = step / KK, tap = step - cb * KK;
        const int wtap = p.flip ? (KK - 1 - tap) : tap;
        const size_t koff = (size_t)(wtap * p.Cg + cb * 32) * 32;      // k-block index * 1024
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            rh[i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + koff) : make_uint4(0, 0, 0, 0);
            if (NB == 2) rl[i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + koff) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_b = [&](int stage, const uint4 (&rh)[RB], const uint4 (&rl)[RB]) {
        unsigned char *Bh = Bst + stage * NB * PLANE_B, *Bl = Bh + PLANE_B;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            *reinterpret_cast<uint4 *>(Bh + boff[i]) = rh[i];
            if (NB == 2) *reinterpret_cast<uint4 *>(Bl + boff[i]) = rl[i];
        }
    };
    const float *Aimg = p.A + (size_t)b * p.H * p.W * p.Cg;
    constexpr int HSLICES = (HPIX * 8 + NT - 1) / NT;      // halo float4s per thread
    float4 hreg[HSLICES];
    auto halo_load = [&](int cb) {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int gy = y0 - p.pad + hy, gx = x0 - p.pad + hx;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v = *reinterpret_cast<const float4 *>(Aimg + ((size_t)gy * p.W + gx) * p.Cg + cb * 32 + c4 * 4);
            }
            hreg[sl] = v;
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                uint2 hi, lo;
                split4t<F16>(hreg[sl], hi, lo);
                *reinterpret_cast<uint2 *>(Ah + pix * AROW + c4 * 8) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Al + pix * AROW + c4 * 8) = lo;
            }
        }
    };
    auto compute = [&](int stage, int step) {
        const int tap = step % KK;
        const int r = tap / KS, s_ = tap - r * KS;
        const int tapoff = (r * HW + s_) * AROW;
        const unsigned char *Bh = Bst + stage * NB * PLANE_B, *Bl = Bh + PLANE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[TM], al[TM], bhf[TN], blf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = aread[i] + tapoff + ks * 32;
                ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + off);
                if (NS == 2) al[i] = *reinterpret_cast<const bf16x8 *>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = bread[j] ^ (ks << 5);
                bhf[j] = *reinterpret_cast<const bf16x8 *>(Bh + off);
                if (NB == 2) blf[j] = *reinterpret_cast<const bf16x8 *>(Bl + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NS == 2)
                        acc[i][j] = mfma16<F16>(al[i], bhf[j], acc[i][j]);
                        if (NB == 2) acc[i][j] = mfma16<F16>(ah[i], blf[j], acc[i][j]);
                    acc[i][j] = mfma16<F16>(ah[i], bhf[j], acc[i][j]);
                }
        }
    };
    // one step: prefetch weights of step+2, multiply `stage`, then publish the weights of step+1 (and, at a channel-block
    // boundary, the next halo, whose loads were issued before the multiply) behind one barrier
    auto do_step = [&](int step, int stage, uint4 (&nh)[RB], uint4 (&nl)[RB], uint4 (&fh)[RB], uint4 (&fl)[RB]) {
        if (step + 2 < T) load_b(step + 2, fh, fl);
        const bool boundary = (step % KK == KK - 1) && (step + 1 < T);
        if (boundary) halo_load(step / KK + 1);
        compute(NBST == 2 ? stage : 0, step);
        if (step + 1 < T) {
            if (boundary || NBST == 1) __syncthreads();   // every wave has finished reading the halo / the single B stage
            if (boundary) halo_store();
            store_b(NBST == 2 ? (stage ^ 1) : 0, nh, nl);
            __syncthreads();
        }
    };

    halo_load(0);
    load_b(0, rb0h, rb0l);
    halo_store();
    store_b(0, rb0h, rb0l);
    if (T > 1) load_b(1, rb0h, rb0l);
    __syncthreads();
#pragma unroll 1
    for (int step = 0; step < T; step += 2) {
        do_step(step, 0, rb0h, rb0l, rb1h, rb1l);
        if (step + 1 < T) do_step(step + 1, 1, rb1h, rb1l, rb0h, rb0l);
    }

    // epilogue: MFMA row (reg) -> pixel inside the wave's 32-pixel image row; col = lane&31 -> channel
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
        const int oy = y0 + wm * TM + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n < p.N) {
                    float v = acc[i][j][r] * p.oscale;
                    v += bias_r[j];
                    v = fast_act(v, nslope, special, p.act, p.slope);
                    if (p.addend) v += p.addend[pix * p.N + n];
                    p.C[pix * p.N + n] = v;
                }
            }
        }
    }
}

// 3x3 variant with ONE TAP ROW (three taps) per step: the weight tiles of taps (r,0..2) of a 32-channel block are
// published together, so the two barriers, the weight publication and the fragment-read ramp that bracket every step are
// paid once per 3 x 768 MFMA cycles instead of once per 768.  LDS: halo 32 KB + 3 weight tiles 48 KB = 80 KB (dynamic),
// exactly two workgroups per CU.  Weights are prefetched one step (2304 MFMA cycles per wave) ahead.
// MODE 0: single LDS stage (two barriers per step; two 80-KB workgroups per CU)
// MODE 1: halo and weight tiles double-buffered (one 160-KB workgroup per CU, one barrier per step)
// MODE 2: weight tiles double-buffered, halo single (WM = 4: 8 rows x 32 pixels per workgroup, 152 KB) -- the weight tile
//         of a step is shared by twice the pixels, which halves the dominant L2 -> LDS stream: in-kernel stamps
//         (tools/stamp_halo.py) show the 4x32 tile waiting on the per-CU fill path (~30 B/clk/CU), not on the MFMA
template <int NSX, int WM, int WN, int BN, int MODE, bool F16>
__global__ __launch_bounds__(64 * WM * WN) void conv_halo3_bf16_kernel(const HaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int KS = 3, TH = 2 * WM, TW = 32;
    constexpr int NT = 64 * WM * WN;
    constexpr bool DB = MODE == 1;
    constexpr int RB = BN * 4 >= NT ? BN * 4 / NT : 1;     // 16-B weight chunks per thread per plane per tap
    constexpr bool B_PART = BN * 4 < NT;                   // more threads than chunks: only the first BN*4 threads load weights
    constexpr int HH = TH + KS - 1, HW = TW + KS - 1, HPIX = HH * HW;
    constexpr int AROW = 80;
    constexpr int PLANE_A = HPIX * AROW, PLANE_B = BN * 64;
    constexpr int TM = 2, TN = BN / (32 * WN);
    static_assert(TN >= 1, "BN = 64 needs the 4-wave variant");
    // DB (the 8-wave variant: ONE workgroup per CU, so LDS is free): halo and weight tiles are double-buffered -- the next
    // step's weights (and, at a channel-block boundary, the next halo) are written into the other buffer BEFORE this
    // step's multiply, one barrier per step, nothing but barrier skew is exposed.  160 KB exactly.
    constexpr int NBUF_A = MODE == 1 ? 2 : 1, NBUF_B = MODE == 0 ? 1 : 2;
    constexpr int ABUF = NS * PLANE_A, BBUF = KS * NB * PLANE_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NBUF_A * ABUF + NBUF_B * BBUF
    unsigned char *Abase = smem;
    unsigned char *Bbase = smem + NBUF_A * ABUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-contiguous tile ranges (hoig_xcd_remap) enumerate the PIXEL tiles of one channel tile first: an XCD then streams
    // the weights of one or two channel tiles (2.4 MB each at 512x512x3x3) through its 4-MB L2 instead of all of them
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int n_mt = p.nblk / p.nblk_n;
    int mt = p.nmajor ? tile % n_mt : tile / p.nblk_n;
    const int n0 = (p.nmajor ? tile / n_mt : tile % p.nblk_n) * BN;
    const int tx_ = mt % p.tiles_x;
    mt /= p.tiles_x;
    const int ty_ = mt % p.tiles_y, b = mt / p.tiles_y;
    const int y0 = ty_ * TH, x0 = tx_ * TW;

    const int brow = tid >> 2, bchunk = tid & 3;
    const unsigned short *wrow_h[RB], *wrow_l[RB];
    int boff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + brow + (NT / 4) * i;
        const size_t o = plane_index(n, bchunk * 8, p.K);
        const bool ok = n < p.N && (!B_PART || brow < BN);
        wrow_h[i] = ok ? p.Wh + o : nullptr;
        wrow_l[i] = (NB == 2 && ok) ? p.Wl + o : nullptr;
        const int row = brow + (NT / 4) * i;
        boff[i] = row * 64 + ((bchunk ^ ((row >> 2) & 3)) << 4);
    }
    int aread[TM], bread[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aread[i] = ((wm * TM + i) * HW + l31) * AROW + lh * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + l31;
        bread[j] = row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 rbh[KS][RB], rbl[KS][RB];
    const int ncb = p.Cg >> 5, T = ncb * KS;               // step = (channel block, tap row)
    auto load_b = [&](int step) {
        const int cb = step / KS, r = step - cb * KS;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const int tap = r * KS + t;
            const int wtap = p.flip ? (KS * KS - 1 - tap) : tap;
            const size_t koff = (size_t)(wtap * p.Cg + cb * 32) * 32;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                rbh[t][i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + koff) : make_uint4(0, 0, 0, 0);
                if (NB == 2)
                    rbl[t][i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + koff) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            unsigned char *Bh = Bbase + buf * BBUF + t * NB * PLANE_B, *Bl = Bh + PLANE_B;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (B_PART && brow >= BN) continue;
                *reinterpret_cast<uint4 *>(Bh + boff[i]) = rbh[t][i];
                if (NB == 2) *reinterpret_cast<uint4 *>(Bl + boff[i]) = rbl[t][i];
            }
        }
    };
    constexpr int HSLICES = (HPIX * 8 + NT - 1) / NT;
    float4 hreg[HSLICES];
    auto halo_load = [&](int cb) {
        // one or two source tensors along the channel axis (p.A2: see HaloArgs)
        const bool second = p.A2 != nullptr && cb * 32 >= p.cg1;
        const int ld = p.A2 ? (second ? p.Cg - p.cg1 : p.cg1) : p.Cg;
        const float *Aimg = (second ? p.A2 : p.A) + (size_t)b * p.H * p.W * ld + (second ? cb * 32 - p.cg1 : cb * 32);
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int gy = y0 - p.pad + hy, gx = x0 - p.pad + hx;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v = *reinterpret_cast<const float4 *>(Aimg + ((size_t)gy * p.W + gx) * ld + c4 * 4);
            }
            hreg[sl] = v;
        }
    };
    auto halo_store = [&](int buf) {
        unsigned char *Ah = Abase + buf * ABUF, *Al = Ah + PLANE_A;
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                uint2 hi, lo;
                split4t<F16>(hreg[sl], hi, lo);
                *reinterpret_cast<uint2 *>(Ah + pix * AROW + c4 * 8) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Al + pix * AROW + c4 * 8) = lo;
            }
        }
    };
    // fragments of sub-step i+1 (tap t, k-half ks) are read into a second register set before the MFMAs of sub-step i
    // issue: with one or two waves per SIMD the LDS latency of a just-in-time read is otherwise exposed
    struct Frags {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
    };
    auto compute = [&](int r, int abuf, int bbuf) {
        const unsigned char *Ah = Abase + abuf * ABUF, *Al = Ah + PLANE_A;
        auto read = [&](Frags &f, int i) {
            const int t = i >> 1, ks = i & 1;
            const int tapoff = (r * HW + t) * AROW;
            const unsigned char *Bh = Bbase + bbuf * BBUF + t * NB * PLANE_B, *Bl = Bh + PLANE_B;
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) {
                const int off = aread[ii] + tapoff + ks * 32;
                f.ah[ii] = *reinterpret_cast<const bf16x8 *>(Ah + off);
                if (NS == 2) f.al[ii] = *reinterpret_cast<const bf16x8 *>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = bread[j] ^ (ks << 5);
                f.bh[j] = *reinterpret_cast<const bf16x8 *>(Bh + off);
                if (NB == 2) f.bl[j] = *reinterpret_cast<const bf16x8 *>(Bl + off);
            }
        };
        auto mma = [&](const Frags &f) {
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NS == 2)
                        acc[ii][j] = mfma16<F16>(f.al[ii], f.bh[j], acc[ii][j]);
                        if (NB == 2) acc[ii][j] = mfma16<F16>(f.ah[ii], f.bl[j], acc[ii][j]);
                    acc[ii][j] = mfma16<F16>(f.ah[ii], f.bh[j], acc[ii][j]);
                }
        };
        Frags f0, f1;
        read(f0, 0);
#pragma unroll
        for (int i = 0; i < 2 * KS; i += 2) {
            read(f1, i + 1);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads of sub-step i+1 AHEAD of the MFMAs of sub-step i
            mma(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < 2 * KS) read(f0, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // (staging the weight tiles by LDS-DMA -- global_load_lds, the pre-swizzled plane blocks are LDS images -- measured
    // ~9 % SLOWER here than the register-staged ds_write_b128 path below)
#ifdef HOIG_STAMP
    unsigned long long c_issue = 0, c_comp = 0, c_b1 = 0, c_st = 0, c_b2 = 0;
    const unsigned long long t_begin = clock64();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#if HOIG_STAMP == 2                /* clock check: only the two stamps around the loop execute */
#define STAMP(v)
#else
#define STAMP(v) { const unsigned long long t_ = clock64(); v += t_ - t_prev; t_prev = t_; }
#endif
    unsigned long long t_prev = t_begin;
#else
#define STAMP(v)
#endif
    halo_load(0);
    halo_store(0);
    load_b(0);
    store_b(0);
    if (MODE == 2) {
        if (T > 1) load_b(1);
        __syncthreads();
        int bbuf = 0;
#pragma unroll 1
        for (int step = 0; step < T; ++step) {
            const int cb = step / KS, r = step - cb * KS;
            const bool more = step + 1 < T;
            const bool boundary = more && r == KS - 1;
            if (more) store_b(bbuf ^ 1);                  // weights of step+1 (registers loaded during the previous step)
            if (step + 2 < T) load_b(step + 2);
            if (boundary) halo_load(cb + 1);
            STAMP(c_issue)
            compute(r, 0, bbuf);
            STAMP(c_comp)
            if (boundary) {
                __syncthreads();                          // every wave is done with the halo
                STAMP(c_b1)
                halo_store(0);
                STAMP(c_st)
            }
            __syncthreads();
            STAMP(c_b2)
            bbuf ^= 1;
        }
#ifdef HOIG_STAMP
        if (p.dbg && lane == 0) {
            unsigned long long *d = p.dbg + ((size_t)blockIdx.x * (NT / 64) + wave) * 8;
            d[0] = c_issue; d[1] = c_comp; d[2] = c_b1; d[3] = c_st; d[4] = c_b2; d[5] = clock64() - t_begin;
            d[6] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        }
#endif
    } else if (DB) {
        if (T > 1) load_b(1);
        __syncthreads();
        int abuf = 0, bbuf = 0;
#pragma unroll 1
        for (int step = 0; step < T; ++step) {
            const int cb = step / KS, r = step - cb * KS;
            const bool more = step + 1 < T;
            const bool boundary = more && r == KS - 1;
            if (more) store_b(bbuf ^ 1);                  // weights of step+1 (registers loaded during the previous step)
            if (boundary) halo_store(abuf ^ 1);           // halo of the next channel block (loaded during the previous step)
            if (step + 2 < T) load_b(step + 2);
            if (r == KS - 2 && cb + 1 < ncb) halo_load(cb + 1);
            compute(r, abuf, bbuf);
            __syncthreads();
            bbuf ^= 1;
            if (boundary) abuf ^= 1;
        }
    } else {
        __syncthreads();
#pragma unroll 1
        for (int step = 0; step < T; ++step) {
            const int cb = step / KS, r = step - cb * KS;
            const bool more = step + 1 < T;
            const bool boundary = more && r == KS - 1;
            if (more) load_b(step + 1);
            if (boundary) halo_load(cb + 1);
            STAMP(c_issue)
            compute(r, 0, 0);
            STAMP(c_comp)
            if (more) {
                __syncthreads();                  // every wave has finished reading the weight tiles (and the halo)
                STAMP(c_b1)
                if (boundary) halo_store(0);
                store_b(0);
                STAMP(c_st)
                __syncthreads();
                STAMP(c_b2)
            }
        }
#ifdef HOIG_STAMP
        if (p.dbg && lane == 0) {
            unsigned long long *d = p.dbg + ((size_t)blockIdx.x * (NT / 64) + wave) * 8;
            d[0] = c_issue; d[1] = c_comp; d[2] = c_b1; d[3] = c_st; d[4] = c_b2; d[5] = clock64() - t_begin;
            d[6] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        }
#endif
    }

    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    float bias_r[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + l31;
        bias_r[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
    float st1[TN], st2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st1[j] = st2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oy = y0 + wm * TM + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const size_t pix = ((size_t)b * p.H + oy) * p.W + ox;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n < p.N) {
                    float v = acc[i][j][r] * p.oscale;
                    v += bias_r[j];
                    v = fast_act(v, nslope, special, p.act, p.slope);
                    if (p.addend) v += p.addend[pix * p.N + n];
                    st1[j] += v;
                    st2[j] += v * v;
                    if (!p.C2) p.C[pix * p.N + n] = v;
                    else if (n < p.n1) p.C[pix * p.n1 + n] = v;                  // (a whole 32-column group goes one way)
                    else p.C2[pix * (p.N - p.n1) + (n - p.n1)] = v;
                }
            }
        }
    }
    if (p.stats) halo_stats_epilogue<TN, WM, WN, NT>(st1, st2, smem, p.stats + (size_t)b * 2 * p.N, p.N, n0, wm, wn, lane);
}

template <int NS, int WM, int WN, int BN, int MODE>
int launch_halo3_one(const HaloArgs &a, hipStream_t st) {
    constexpr int HPIX = (2 * WM + 2) * 34;
    constexpr size_t shm = (MODE == 1 ? 2 : 1) * (ns_a(NS) * (HPIX * 80)) + (MODE == 0 ? 1 : 2) * (3 * ns_b(NS) * (BN * 64));
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_bf16_kernel<NS, WM, WN, BN, MODE, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_bf16_kernel<NS, WM, WN, BN, MODE, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    if (a.f16) conv_halo3_bf16_kernel<NS, WM, WN, BN, MODE, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
    else conv_halo3_bf16_kernel<NS, WM, WN, BN, MODE, false><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

#ifdef HOIG_STAMP
static unsigned long long *g_stamp_buf = nullptr;
extern "C" void hoig_debug_set_stamp_buffer(unsigned long long *buf) { g_stamp_buf = buf; }
#endif

int launch_halo3(HaloArgs a, int ns, hipStream_t st) {
#ifdef HOIG_STAMP
    a.dbg = g_stamp_buf;
#endif
    a.tiles_x = a.W / 32;
    a.tiles_y = a.H / 4;
    a.nmajor = 1;
    const bool n64 = (a.N % 128) != 0 || (a.C2 && a.n1 % 128 != 0);      // (a channel tile must not straddle the two outputs)
    a.nblk_n = (int)hoig_cdiv(a.N, n64 ? 64 : 128);
    a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
    const bool m16 = hoig_tuning(HOIG_TUNE_MFMA16) != 0;      // the 8-row tilings on v_mfma_f32_16x16x32 (conv_halo16.hip)
    if (n64) {
        // 64-channel layers (the full-resolution levels): 8 x 32 x 64 on the 16x16 MFMA where that leaves every CU a workgroup
        if (m16 && a.H % 8 == 0 && a.nblk / 2 >= 256) {
            a.tiles_y = a.H / 8;
            a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
            return launch_halo3_m16(a, ns, 64, st);
        }
        if (a.a_split || a.b_split || a.in_scale) return HOIG_EUNSUPPORTED;  // (pre-split input, grouped launch: conv_halo16.hip only)
        HOIG_NS_SWITCH(ns, return launch_halo3_one<NSX, 2, 2, 64, 0>(a, st));
    }
    if (m16 && a.H % 8 == 0 && a.nblk / 2 < 256 && a.nblk >= 192 && a.N % 128 == 0) {
        // the 8-image launches of src_model / tsf_model (which run side by side on two streams) on 128-channel tiles -- 128
        // workgroups each, half the chip per launch -- instead of 256 workgroups of 64-channel tiles (round 4: step -0.65 ms,
        // profiles/r04_few128_ab.txt)
        a.tiles_y = a.H / 8;
        a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
        return launch_halo3_m16(a, ns, 128, st);
    }
    if (a.H % 8 == 0 && a.nblk / 2 < 256 && a.nblk >= 192) {   // too few 8-row tiles at BN = 128: 8 rows x 64 channels
        a.tiles_y = a.H / 8;
        a.nblk_n = a.N / 64;
        a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
        if (m16) return launch_halo3_m16(a, ns, 64, st);
        if (a.a_split || a.b_split || a.in_scale) return HOIG_EUNSUPPORTED;
        HOIG_NS_SWITCH(ns, return launch_halo3_one<NSX, 4, 2, 64, 2>(a, st));
    }
    // 8 x 32 pixel tiles (8 waves, weight tile shared by 256 pixels) when that still gives every CU a workgroup
    if (a.H % 8 == 0 && a.nblk / 2 >= 256) {
        a.tiles_y = a.H / 8;
        a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
        if (m16) return launch_halo3_m16(a, ns, 128, st);
        if (a.a_split || a.b_split || a.in_scale) return HOIG_EUNSUPPORTED;
        HOIG_NS_SWITCH(ns, return launch_halo3_one<NSX, 4, 2, 128, 2>(a, st));
    }
    if (a.a_split || a.b_split || a.in_scale) return HOIG_EUNSUPPORTED;
    const bool wide = a.nblk < 384;
    HOIG_NS_SWITCH(ns, return wide ? launch_halo3_one<NSX, 2, 4, 128, 1>(a, st) : launch_halo3_one<NSX, 2, 2, 128, 0>(a, st));
    return HOIG_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 3x3 layers (pad 1; ConvTranspose2d with output_padding 1) on LDS halo tiles.  On the generic kernel these re-gather
// their fp32 input from L2 / HBM for every tap (the 268-MB full-resolution tensors do not stay in L2) and run at 85-150 TF.
// Decomposed by the PARITY of the fine-grid coordinate, every tap becomes a unit-stride read of a small halo image:
//   GATHER mode (Conv2d s2 forward, ConvTranspose2d data gradient): out[o] = sum_{r,s} in[2o - 1 + (r,s)] w[r][s].  The input
//     is read as its four parity phases in[2i+p][2j+q]; tap r uses phase p = (r != 1) at coarse index o + (r == 0 ? -1 : 0).
//     Per 32-channel block: phase (1,1) serves taps (0,0),(0,2),(2,0),(2,2), phase (1,0) taps (0,1),(2,1), phase (0,1) taps
//     (1,0),(1,2), phase (0,0) tap (1,1): five steps of <= 2 taps, four (TH+1) x 33 halo loads (stride-2 source addressing).
//   SCATTER mode (ConvTranspose2d forward, Conv2d s2 data gradient): out[2i - 1 + (r,s)] += in[i] w[r][s].  A workgroup owns ONE
//     output parity phase (P,Q) of a coarse tile: out[2I+P][2J+Q] = sum over the taps with r = 1 (P = 0) or r in {0,2} (P = 1)
//     of in[I + (r == 0)][J + (s == 0)] w[r][s] -- a stride-1 conv with 1, 2 or 4 taps over one halo image; strided stores.
// Tile: 4 x 32 coarse pixels x BN channels, 4 waves, single LDS stage (58 KB: two workgroups per CU).
template <int NSX, int BN, bool SCATTER, bool F16>
__global__ __launch_bounds__(256) void conv_halo_s2_bf16_kernel(const HaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int TH = 4, TW = 32, NT = 256, WN = 2;
    constexpr int RB = BN * 4 / NT;                        // 16-B weight chunks per thread per plane per tap
    constexpr int HH = TH + 1, HW = TW + 1, HPIX = HH * HW;
    constexpr int AROW = 80;
    constexpr int PLANE_A = HPIX * AROW, PLANE_B = BN * 64;
    constexpr int TM = 2, TN = BN / (32 * WN);
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * PLANE_A + 2 * NB * PLANE_B];
    unsigned char *Ah = smem, *Al = smem + PLANE_A;
    unsigned char *Bbase = smem + NS * PLANE_A;            // two tap tiles of (Bh, Bl)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    int P = 0, Q = 0;
    if (SCATTER) {
        P = (tile >> 1) & 1;
        Q = tile & 1;
        tile >>= 2;
    }
    int mt = tile / p.nblk_n;
    const int n0 = (tile % p.nblk_n) * BN;
    const int tx_ = mt % p.tiles_x;
    mt /= p.tiles_x;
    const int ty_ = mt % p.tiles_y, b = mt / p.tiles_y;
    const int y0 = ty_ * TH, x0 = tx_ * TW;                // coarse-grid tile origin

    const int brow = tid >> 2, bchunk = tid & 3;
    const unsigned short *wrow_h[RB], *wrow_l[RB];
    int boff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + brow + (NT / 4) * i;
        const size_t o = plane_index(n, bchunk * 8, p.K);
        wrow_h[i] = n < p.N ? p.Wh + o : nullptr;
        wrow_l[i] = (NB == 2 && n < p.N) ? p.Wl + o : nullptr;
        const int row = brow + (NT / 4) * i;
        boff[i] = row * 64 + ((bchunk ^ ((row >> 2) & 3)) << 4);
    }
    int aread[TM], bread[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aread[i] = ((wm * TM + i) * HW + l31) * AROW + lh * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + l31;
        bread[j] = row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- step table (wave-uniform scalar code) ----
    struct Step {
        int cb, ntap, tap[2], pp, qq;
        bool load;
    };
    const int spc = SCATTER ? ((P ? 2 : 1) * (Q ? 2 : 1) + 1) / 2 : 5;     // steps per 32-channel block
    const int ncb = p.Cg >> 5, T = ncb * spc;
    auto step_info = [&](int step) -> Step {
        Step s_;
        s_.cb = step / spc;
        const int idx = step - s_.cb * spc;
        s_.pp = s_.qq = 0;
        if (!SCATTER) {
            s_.load = idx != 1;
            switch (idx) {
                case 0: s_.pp = 1; s_.qq = 1; s_.ntap = 2; s_.tap[0] = 0; s_.tap[1] = 2; break;
                case 1: s_.pp = 1; s_.qq = 1; s_.ntap = 2; s_.tap[0] = 6; s_.tap[1] = 8; break;
                case 2: s_.pp = 1; s_.qq = 0; s_.ntap = 2; s_.tap[0] = 1; s_.tap[1] = 7; break;
                case 3: s_.pp = 0; s_.qq = 1; s_.ntap = 2; s_.tap[0] = 3; s_.tap[1] = 5; break;
                default: s_.ntap = 1; s_.tap[0] = 4; s_.tap[1] = 4; break;
            }
        } else {
            s_.load = idx == 0;
            // rows of the phase: P = 0 -> r = 1; P = 1 -> r in {0, 2}; same for columns
            const int r0 = P ? 0 : 1, r1 = 2, s0 = Q ? 0 : 1, s1 = 2;
            if (P && Q) {
                s_.ntap = 2;
                s_.tap[0] = (idx ? r1 : r0) * 3 + s0;
                s_.tap[1] = (idx ? r1 : r0) * 3 + s1;
            } else if (P) {
                s_.ntap = 2; s_.tap[0] = r0 * 3 + s0; s_.tap[1] = r1 * 3 + s0;
            } else if (Q) {
                s_.ntap = 2; s_.tap[0] = r0 * 3 + s0; s_.tap[1] = r0 * 3 + s1;
            } else {
                s_.ntap = 1; s_.tap[0] = s_.tap[1] = 4;
            }
        }
        return s_;
    };
    // halo offset (rows, columns in {0,1}) of tap (r,s):  gather: (r != 0, s != 0)   scatter: (r == 0, s == 0)
    auto tap_off = [&](int tap) -> int {
        const int r = tap / 3, s_ = tap - r * 3;
        const int dr = SCATTER ? (r == 0) : (r != 0), dc = SCATTER ? (s_ == 0) : (s_ != 0);
        return (dr * HW + dc) * AROW;
    };

    uint4 rbh[2][RB], rbl[2][RB];
    auto load_b = [&](const Step &s_) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const size_t koff = (size_t)(s_.tap[t] * p.Cg + s_.cb * 32) * 32;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                rbh[t][i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + koff) : make_uint4(0, 0, 0, 0);
                if (NB == 2)
                    rbl[t][i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + koff) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned char *Bh = Bbase + t * NB * PLANE_B, *Bl = Bh + PLANE_B;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                *reinterpret_cast<uint4 *>(Bh + boff[i]) = rbh[t][i];
                if (NB == 2) *reinterpret_cast<uint4 *>(Bl + boff[i]) = rbl[t][i];
            }
        }
    };
    const float *Aimg = p.A + (size_t)b * p.H * p.W * p.Cg;       // p.H x p.W: the gathered tensor (fine grid in gather mode)
    constexpr int HSLICES = (HPIX * 8 + NT - 1) / NT;
    // halo images are fetched TWO steps ahead into alternating register sets: a step is only one or two taps long (768-1536
    // MFMA cycles per wave), less than the HBM latency of the full-resolution tensors
    float4 hregs[2][HSLICES];
    auto halo_load = [&](const Step &s_, float4 (&hreg)[HSLICES]) {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int gy = SCATTER ? y0 + hy : 2 * (y0 - 1 + hy) + s_.pp;
                const int gx = SCATTER ? x0 + hx : 2 * (x0 - 1 + hx) + s_.qq;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v = *reinterpret_cast<const float4 *>(Aimg + ((size_t)gy * p.W + gx) * p.Cg + s_.cb * 32 + c4 * 4);
            }
            hreg[sl] = v;
        }
    };
    auto halo_store = [&](const float4 (&hreg)[HSLICES]) {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                uint2 hi, lo;
                split4t<F16>(hreg[sl], hi, lo);
                *reinterpret_cast<uint2 *>(Ah + pix * AROW + c4 * 8) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Al + pix * AROW + c4 * 8) = lo;
            }
        }
    };
    auto compute = [&](const Step &s_) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t >= s_.ntap) break;
            const int tapoff = tap_off(s_.tap[t]);
            const unsigned char *Bh = Bbase + t * NB * PLANE_B, *Bl = Bh + PLANE_B;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[TM], al[TM], bhf[TN], blf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int off = aread[i] + tapoff + ks * 32;
                    ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + off);
                    if (NS == 2) al[i] = *reinterpret_cast<const bf16x8 *>(Al + off);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int off = bread[j] ^ (ks << 5);
                    bhf[j] = *reinterpret_cast<const bf16x8 *>(Bh + off);
                    if (NB == 2) blf[j] = *reinterpret_cast<const bf16x8 *>(Bl + off);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (NS == 2)
                            acc[i][j] = mfma16<F16>(al[i], bhf[j], acc[i][j]);
                            if (NB == 2) acc[i][j] = mfma16<F16>(ah[i], blf[j], acc[i][j]);
                        acc[i][j] = mfma16<F16>(ah[i], bhf[j], acc[i][j]);
                    }
            }
        }
    };

    Step cur = step_info(0);
    halo_load(cur, hregs[0]);
    load_b(cur);
    halo_store(hregs[0]);
    store_b();
    Step nxt = cur;
    if (T > 1) {
        nxt = step_info(1);
        if (nxt.load) halo_load(nxt, hregs[1]);           // (the set that "step -1" would have filled)
    }
    __syncthreads();
    // two-fold unrolled so that the register-set index is static
    auto one_step = [&](int step, float4 (&mine)[HSLICES], float4 (&other)[HSLICES]) {
        const bool more = step + 1 < T;
        if (more) load_b(nxt);
        if (step + 2 < T) {
            const Step n2 = step_info(step + 2);
            if (n2.load) halo_load(n2, mine);             // stored at the end of step+1
        }
        compute(cur);
        if (more) {
            __syncthreads();                  // every wave has finished reading the weight tiles (and the halo)
            if (nxt.load) halo_store(other);  // fetched during step-1
            store_b();
            __syncthreads();
            cur = nxt;
            if (step + 2 < T) nxt = step_info(step + 2);
        }
    };
#pragma unroll 1
    for (int step = 0; step < T; step += 2) {
        one_step(step, hregs[0], hregs[1]);
        if (step + 1 < T) one_step(step + 1, hregs[1], hregs[0]);
    }

    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    float bias_r[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + l31;
        bias_r[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
    float st1[TN], st2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) st1[j] = st2[j] = 0.f;
    // output grid: gather mode = the coarse grid (tiles_y*TH x tiles_x*TW); scatter mode = twice the coarse grid, phase (P,Q)
    const int Ho = SCATTER ? 2 * p.H : p.tiles_y * TH, Wo = SCATTER ? 2 * p.W : p.tiles_x * TW;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int cy = y0 + wm * TM + i;
        const int oy = SCATTER ? 2 * cy + P : cy;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ox = SCATTER ? 2 * cx + Q : cx;
            const size_t pix = ((size_t)b * Ho + oy) * Wo + ox;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (TN * 32) + j * 32 + l31;
                if (n < p.N) {
                    float v = acc[i][j][r] * p.oscale;
                    v += bias_r[j];
                    v = fast_act(v, nslope, special, p.act, p.slope);
                    if (p.addend) v += p.addend[pix * p.N + n];
                    st1[j] += v;
                    st2[j] += v * v;
                    p.C[pix * p.N + n] = v;
                }
            }
        }
    }
    if (p.stats) halo_stats_epilogue<TN, 2, WN, NT>(st1, st2, smem, p.stats + (size_t)b * 2 * p.N, p.N, n0, wm, wn, lane);
}

// a: H, W = spatial size of the GATHERED tensor (gather mode: the fine grid, output is H/2 x W/2; scatter mode: the coarse
// grid, output is 2H x 2W)
template <bool SCATTER>
int launch_halo_s2(HaloArgs a, int ns, hipStream_t st) {
    const int ch = SCATTER ? a.H : a.H / 2, cw = SCATTER ? a.W : a.W / 2;      // coarse grid
    a.tiles_x = cw / 32;
    a.tiles_y = ch / 4;
    const bool n64 = (a.N % 128) != 0;
    a.nblk_n = a.N / (n64 ? 64 : 128);
    a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n * (SCATTER ? 4 : 1);
    a.nmajor = 0;
    // on v_mfma_f32_16x16x32 (conv_halo16.hip): +10..24 % (profiles/r04_s2_16_ab.txt) except the THREE-term scatter launches with
    // 64-channel tiles (ConvTranspose2d 128 -> 64 forward at full resolution: -5 %; the two-term data gradient of the same
    // shape: +24 %), which stay on the 32x32 kernel
    if (hoig_tuning(HOIG_TUNE_S2_16) != 0 && !(SCATTER && n64 && ns == 2)) return launch_halo_s2_m16(a, ns, SCATTER, st);
    if (n64) {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_s2_bf16_kernel<NSX, 64, SCATTER, true><<<a.nblk, 256, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_s2_bf16_kernel<NSX, 64, SCATTER, false><<<a.nblk, 256, 0, st>>>(a));
    } else {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_s2_bf16_kernel<NSX, 128, SCATTER, true><<<a.nblk, 256, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_s2_bf16_kernel<NSX, 128, SCATTER, false><<<a.nblk, 256, 0, st>>>(a));
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

template <int KS>
int launch_halo(HaloArgs a, int ns, hipStream_t st) {
    a.tiles_x = a.W / 32;
    a.tiles_y = a.H / 4;
    const bool n64 = (a.N % 128) != 0;        // 64-channel layers (the full-resolution levels, VGG conv1): BN = 64 tiles
    a.nblk_n = (int)hoig_cdiv(a.N, n64 ? 64 : 128);
    a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
    if (n64) {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_bf16_kernel<KS, NSX, 2, 64, true><<<a.nblk, 256, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_bf16_kernel<KS, NSX, 2, 64, false><<<a.nblk, 256, 0, st>>>(a));
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    // fewer than ~1.5 workgroups per CU: 8 waves per workgroup keep two waves on every SIMD
    const bool wide = a.nblk < 384;
    if (wide) {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_bf16_kernel<KS, NSX, 4, 128, true><<<a.nblk, 512, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_bf16_kernel<KS, NSX, 4, 128, false><<<a.nblk, 512, 0, st>>>(a));
    } else {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_bf16_kernel<KS, NSX, 2, 128, true><<<a.nblk, 256, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_bf16_kernel<KS, NSX, 2, 128, false><<<a.nblk, 256, 0, st>>>(a));
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

// a2 / cg1: the gathered tensor is [a | a2] along channels; c2 / n1: the output is [c | c2] (3x3 stride-1 halo kernel only:
// HOIG_EUNSUPPORTED for every other shape, the caller then concatenates / slices itself)
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

// the second problem of a grouped launch (hoig_conv2d_*_pair): same descriptor, its own tensors
// the norm a forward launch applies to its gathered tensor (HaloArgs::in_scale ...)
struct InNorm {
    const float *scale, *shift;
    int relu_c0;
};
struct PairSet {
    const float *a;
    const unsigned short *wh, *wl;
    const float *bias, *addend;
    float *c;
};

int run(const hoig_conv_desc *d, const float *a, const unsigned short *wh, const unsigned short *wl, const float *bias,
        float *c, bool dgrad, hipStream_t st, const float *a2 = nullptr, int cg1 = 0, float *c2 = nullptr, int n1 = 0,
        const float *addend = nullptr, float *stats = nullptr, bool a_split = false, const PairSet *g2 = nullptr,
        const InNorm *in = nullptr) {
    if (in && (dgrad || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->transposed || a_split || g2 ||
               hoig_tuning(HOIG_TUNE_MFMA16) == 0))
        return HOIG_EUNSUPPORTED;
    Args p;
    p.A = a; p.Wh = wh; p.Wl = wl; p.bias = bias; p.C = c;
    p.f16 = dgrad ? 0 : 1;                       // forward: fp16-split operands over the 2^8-scaled forward planes
    p.oscale = dgrad ? 1.f : 1.f / W_SCALE_F16;
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
    const int ns = ns_of_precision(d->precision);
    if (ns == 2 && !wl) return HOIG_EINVAL;
    const long t128 = hoig_cdiv(p.M, 128);
    if (p.N <= 32 || p.N % 32 != 0) return HOIG_EUNSUPPORTED;
    if (dgrad && !d->transposed && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && g.Cg == 128 &&
        p.N >= 1024 && p.N % 64 == 0 && p.M % 128 == 0 && !a2 && !c2 && !addend && !stats)
        return launch_dgrad_thin(a, wh, wl, c, p.M, p.N, ns, st);
    // stride-1 "same" convolutions (and their data gradients): LDS-resident input halo, weights streamed per tap
    if (!d->transposed && d->stride == 1 && d->R == d->S && 2 * d->pad == d->R - 1 && (d->R == 1 || d->R == 3 || d->R == 5) &&
        d->Wi % 32 == 0 && d->Hi % 4 == 0 && p.N % 64 == 0 &&
        (long)(g2 ? 2 : 1) * d->B * (d->Hi / 4) * (d->Wi / 32) * ((p.N + 127) / 128) >= 160) {   // fewer tiles: the generic kernel splits K
        HaloArgs h;
        h.b_split = 0;
        h.A_g2 = nullptr; h.Wh_g2 = h.Wl_g2 = nullptr; h.bias_g2 = h.addend_g2 = nullptr; h.C_g2 = nullptr;
        h.A = a; h.Wh = wh; h.Wl = wl; h.bias = bias; h.C = c;
        h.A2 = a2; h.cg1 = cg1; h.C2 = c2; h.n1 = n1;
        h.addend = addend;
        h.stats = stats;
        h.a_split = a_split ? 1 : 0;
        h.in_scale = in ? in->scale : nullptr; h.in_shift = in ? in->shift : nullptr; h.in_relu_c0 = in ? in->relu_c0 : 0;
        if (a_split && (d->R != 3 || a2 || !dgrad)) return HOIG_EUNSUPPORTED;
        if ((addend || stats) && c2) return HOIG_EUNSUPPORTED;
        if (stats && d->R != 3) return HOIG_EUNSUPPORTED;          // (only the 3x3 kernel has the statistics epilogue)
        if ((a2 || c2) && d->R != 3) return HOIG_EUNSUPPORTED;
        if (a2 && (cg1 % 32 || cg1 <= 0 || cg1 >= g.Cg)) return HOIG_EINVAL;
        if (c2 && (n1 % 64 || n1 <= 0 || n1 >= p.N)) return HOIG_EINVAL;
        h.Bn = d->B; h.H = d->Hi; h.W = d->Wi; h.Cg = g.Cg; h.N = p.N; h.K = p.K;
        if (g2) {                            // grouped launch: both problems' images in one grid (3x3 kernel of conv_halo16.hip only)
            if (d->R != 3 || a2 || c2 || stats) return HOIG_EUNSUPPORTED;
            h.Bn = 2 * d->B; h.b_split = d->B;
            h.A_g2 = g2->a; h.Wh_g2 = g2->wh; h.Wl_g2 = g2->wl; h.bias_g2 = g2->bias; h.addend_g2 = g2->addend; h.C_g2 = g2->c;
        }
        h.pad = d->pad;                      // dgrad: KS-1-pad == pad for "same" convolutions
        h.flip = dgrad ? 1 : 0;
        h.act = p.act; h.slope = p.slope;
            h.f16 = p.f16; h.oscale = p.oscale;
        if (d->R == 1) return launch_halo<1>(h, ns, st);
        if (d->R == 3) return launch_halo3(h, ns, st);
        return launch_halo<5>(h, ns, st);
    }
    if (a2 || c2 || a_split || g2 || in) return HOIG_EUNSUPPORTED;
    // 3x3 "same" layers with too few tiles for the halo kernel above (N = 128 at 32 x 32: the data gradient of SPADE's 128 -> 1024
    // convolutions): a valid convolution over the zero-padded canvas on the flattened-axis kernel, split over the channel blocks
    if (hoig_tuning(HOIG_TUNE_FLAT5) >= 2 && !d->transposed && d->stride == 1 && d->R == 3 && d->S == 3 && d->pad == 1 &&
        p.N % 128 == 0 && !stats) {
        FlatArgs f;
        f.A = a; f.Wh = wh; f.Wl = wl; f.bias = bias; f.C = c; f.addend = addend;
        f.Bn = d->B; f.Hc = d->Hi + 2; f.Wc = d->Wi + 2; f.KS = 3; f.act = p.act; f.slope = p.slope;
        f.Cg = g.Cg; f.N = p.N; f.K = p.K; f.flip = dgrad ? 1 : 0; f.f16 = p.f16; f.oscale = p.oscale;
        f.Hs = d->Hi; f.Ws = d->Wi; f.oy = f.ox = 1; f.Hd = d->Hi; f.Wd = d->Wi;
        const int rc = launch_flat_m16(f, ns, st);
        if (rc != HOIG_EUNSUPPORTED) return rc;
    }
    // the attention's VALID 5x5 convolutions over narrow maps (and their data gradients) on the flattened-axis halo kernel
    if (hoig_tuning(HOIG_TUNE_FLAT5) != 0 && !d->transposed && d->stride == 1 && d->R == 5 && d->S == 5 && d->pad == 0 &&
        d->Ho == d->Hi - 4 && d->Wo == d->Wi - 4 && p.N % 128 == 0 && !stats) {
        FlatArgs f;
        f.A = a; f.Wh = wh; f.Wl = wl; f.bias = bias; f.C = c; f.addend = addend;
        f.Bn = d->B; f.Hc = d->Hi; f.Wc = d->Wi; f.KS = 5; f.act = p.act; f.slope = p.slope;
        f.Cg = g.Cg; f.N = p.N; f.K = p.K; f.flip = dgrad ? 1 : 0; f.f16 = p.f16; f.oscale = p.oscale;
        if (!dgrad) {
            f.Hs = d->Hi; f.Ws = d->Wi; f.oy = f.ox = 0; f.Hd = d->Ho; f.Wd = d->Wo;
        } else {
            f.Hs = d->Ho; f.Ws = d->Wo; f.oy = f.ox = 4; f.Hd = d->Hi; f.Wd = d->Wi;
        }
        const int rc = launch_flat_m16(f, ns, st);
        if (rc != HOIG_EUNSUPPORTED) return rc;
    }
    // stride-2 3x3 pad-1 layers on the parity-phase halo kernel.  gather: Conv2d forward / ConvTranspose2d data gradient;
    // scatter: ConvTranspose2d forward / Conv2d data gradient
    if (d->stride == 2 && d->R == 3 && d->S == 3 && d->pad == 1 && d->Hi % 2 == 0 && d->Wi % 2 == 0 &&
        p.N % 64 == 0) {
        // fine / coarse grids: Conv2d: fine = input (Hi), coarse = output (Ho = Hi/2); ConvTranspose2d: fine = output
        const int fine_h = d->transposed ? d->Ho : d->Hi, fine_w = d->transposed ? d->Wo : d->Wi;
        const int coarse_h = d->transposed ? d->Hi : d->Ho, coarse_w = d->transposed ? d->Wi : d->Wo;
        if (fine_h == 2 * coarse_h && fine_w == 2 * coarse_w && coarse_w % 32 == 0 && coarse_h % 4 == 0) {
            HaloArgs h;
            h.A = a; h.Wh = wh; h.Wl = wl; h.bias = bias; h.C = c;
            h.Bn = d->B; h.Cg = g.Cg; h.N = p.N; h.K = p.K;
            h.pad = 1; h.flip = 0;
            h.A2 = nullptr; h.cg1 = 0; h.C2 = nullptr; h.n1 = 0; h.addend = addend; h.stats = stats; h.a_split = 0;
            h.b_split = 0;
            h.act = p.act; h.slope = p.slope;
            h.f16 = p.f16; h.oscale = p.oscale;
            const bool gather = !g.gatherT;      // the operand is read at 2*o - 1 + tap (fine grid) -> gather mode
            if (gather) {
                h.H = fine_h; h.W = fine_w;
                return launch_halo_s2<false>(h, ns, st);
            }
            h.H = coarse_h; h.W = coarse_w;
            return launch_halo_s2<true>(h, ns, st);
        }
    }
    if (addend || stats) return HOIG_EUNSUPPORTED;
    // on v_mfma_f32_16x16x32 (conv_igemm16.hip): 1 = forward launches (three fp16 terms: +16 % on the attention's 5x5 convolutions),
    // 2 = data gradients too (two bf16 terms per k-block leave less to hide the single LDS stage behind: measured 0-25 % slower)
    const int t16 = hoig_tuning(HOIG_TUNE_IGEMM16);
    const bool m16 = t16 >= 2 || (t16 == 1 && !dgrad);
    if (p.N <= 64) {
        if (t128 >= 512) return m16 ? launch_igemm_m16(p, ns, 3, st) : launch<128, 64, 2, 2>(p, ns, st);
        return m16 ? launch_igemm_m16(p, ns, 4, st) : launch<64, 64, 2, 2>(p, ns, st);
    }
    const long n128 = hoig_cdiv(p.N, 128);
    // fewer than two 128x128 workgroups per CU: run 8 waves per workgroup so every SIMD still holds two waves and one
    // wave's bf16 split (VALU) overlaps the other's MFMAs.  (A BK = 64, double-buffered variant measured slower.)
    if (t128 * n128 >= 512) return m16 ? launch_igemm_m16(p, ns, 0, st) : launch<128, 128, 2, 2>(p, ns, st);
    if (t128 * n128 >= 128) return m16 ? launch_igemm_m16(p, ns, 1, st) : launch<128, 128, 2, 4>(p, ns, st);
    return m16 ? launch_igemm_m16(p, ns, 2, st) : launch<64, 128, 2, 2>(p, ns, st);
}

}  // namespace

// fp32-weight entry points cannot use the bf16 path (it needs the pre-split planes): tell the dispatcher to fall back.
int hoig_conv_bf16_fwd_like(const hoig_conv_desc *, const float *, const float *, const float *, float *, bool,
                            hipStream_t) {
    return HOIG_EUNSUPPORTED;
}

extern "C" int hoig_pack_conv_weight_bf16(const float *w, int Co, int RS, int Ci, int for_dgrad, uint16_t *hi,
                                          uint16_t *lo, hoig_stream_t stream) {
    if (!w || !hi || Co <= 0 || RS <= 0 || Ci <= 0) return HOIG_EINVAL;
    if ((Co & 31) || (Ci & 31)) return HOIG_EUNSUPPORTED;      // the blocked plane layout needs whole 32x32 blocks
    const int64_t n = (int64_t)Co * RS * Ci;
    pack_weight_kernel<<<hoig_stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, Co, RS, Ci, for_dgrad ? 1 : 0, hi, lo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_pack_conv_weights_bf16_all(const float *flat, const int64_t *segs, int nseg, int64_t ntiles,
                                               uint16_t *hi_f, uint16_t *lo_f, uint16_t *hi_d, uint16_t *lo_d,
                                               hoig_stream_t stream) {
    if (!flat || !segs || nseg <= 0 || ntiles <= 0 || !hi_f || !lo_f || !hi_d || !lo_d) return HOIG_EINVAL;
    pack_all_kernel<<<(unsigned)ntiles, 256, 0, (hipStream_t)stream>>>(flat, segs, nseg, hi_f, lo_f, hi_d, lo_d);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_adam_pack_step(float *flat, const float *grad, float *exp_avg, float *exp_avg_sq, const float *derived,
                                   float grad_scale, const int64_t *segs, int nseg, int64_t ntiles, const int64_t *plain,
                                   int64_t nplain, uint16_t *hi_f, uint16_t *lo_f, uint16_t *hi_d, uint16_t *lo_d,
                                   hoig_stream_t stream) {
    if (!flat || !grad || !exp_avg || !exp_avg_sq || !derived || !segs || nseg <= 0 || ntiles <= 0 || nplain < 0 || (nplain && !plain) ||
        !hi_f || !lo_f || !hi_d || !lo_d)
        return HOIG_EINVAL;
    adam_pack_kernel<<<(unsigned)(ntiles + nplain), 256, 0, (hipStream_t)stream>>>(flat, grad, exp_avg, exp_avg_sq, derived, grad_scale, segs,
                                                                                 nseg, ntiles, plain, hi_f, lo_f, hi_d, lo_d);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_conv2d_fwd_packed(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint16_t *w_lo,
                                      const float *bias, float *y, hoig_stream_t stream) {
    if (!d || !x || !w_hi || !y) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, x, w_hi, w_lo, bias, y, false, (hipStream_t)stream);
}

extern "C" int hoig_conv2d_bwd_data_packed(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi,
                                           const uint16_t *wt_lo, float *dx, hoig_stream_t stream) {
    if (!d || !dy || !wt_hi || !dx) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, dy, wt_hi, wt_lo, nullptr, dx, true, (hipStream_t)stream);
}

// y = conv(x) and, from the same epilogue, stats[b][0/1][co] += sum / sum of squares of y over image b (HOIG_EUNSUPPORTED where the
// layer's kernel has no such epilogue: 3x3 stride-1 "same" and 3x3 stride-2 layers on the halo kernels have it)
extern "C" int hoig_conv2d_fwd_packed_stats(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint16_t *w_lo,
                                            const float *bias, float *y, float *stats, hoig_stream_t stream) {
    if (!d || !x || !w_hi || !y || !stats) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, x, w_hi, w_lo, bias, y, false, (hipStream_t)stream, nullptr, 0, nullptr, 0, nullptr, stats);
}
extern "C" int hoig_conv2d_cat_fwd_packed_stats(const hoig_conv_desc *d, const float *x1, int C1, const float *x2,
                                                const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, float *y,
                                                float *stats, hoig_stream_t stream) {
    if (!d || !x1 || !x2 || !w_hi || !y || !stats) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, x1, w_hi, w_lo, bias, y, false, (hipStream_t)stream, x2, C1, nullptr, 0, nullptr, stats);
}

// forward of conv -> instance norm (+ affine) -> ReLU -> THIS 3x3 stride-1 convolution in INFERENCE: x (and x2) are RAW, the loader
// applies x * in_scale + in_shift per (image, gathered channel) and ReLU on channels >= in_relu_c0 (include/hoig_kernels.h)
extern "C" int hoig_conv2d_fwd_packed_normin(const hoig_conv_desc *d, const float *x, int C1, const float *x2, const uint16_t *w_hi,
                                             const uint16_t *w_lo, const float *bias, const float *in_scale, const float *in_shift,
                                             int in_relu_c0, float *y, float *stats, hoig_stream_t stream) {
    if (!d || !x || !w_hi || !y || !in_scale || !in_shift || in_relu_c0 < 0) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    const InNorm in{in_scale, in_shift, in_relu_c0};
    return run(d, x, w_hi, w_lo, bias, y, false, (hipStream_t)stream, x2, x2 ? C1 : 0, nullptr, 0, nullptr, stats, false, nullptr, &in);
}

// dx = data gradient + addend (HOIG_EUNSUPPORTED where the layer's kernel has no such epilogue: the caller adds separately)
extern "C" int hoig_conv2d_bwd_data_packed_add(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi,
                                               const uint16_t *wt_lo, const float *addend, float *dx, hoig_stream_t stream) {
    if (!d || !dy || !wt_hi || !dx || !addend) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, dy, wt_hi, wt_lo, nullptr, dx, true, (hipStream_t)stream, nullptr, 0, nullptr, 0, addend);
}

// data gradient (+ addend, nullable) from PRE-SPLIT dy (include/hoig_kernels.h): the 3x3 stride-1 "same" layers on conv_halo16.hip
extern "C" int hoig_conv2d_bwd_data_packed_split(const hoig_conv_desc *d, const uint16_t *dy_split, const uint16_t *wt_hi,
                                                 const uint16_t *wt_lo, const float *addend, float *dx, hoig_stream_t stream) {
    if (!d || !dy_split || !wt_hi || !dx) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    if (d->precision == HOIG_PREC_BF16X3 || d->transposed || d->stride != 1 || d->R != 3 || d->S != 3) return HOIG_EUNSUPPORTED;
    return run(d, reinterpret_cast<const float *>(dy_split), wt_hi, wt_lo, nullptr, dx, true, (hipStream_t)stream, nullptr, 0, nullptr, 0,
               addend, nullptr, true);
}

// GROUPED launches (include/hoig_kernels.h): two convolutions of ONE descriptor -- different tensors, different weights -- as one grid
extern "C" int hoig_conv2d_fwd_packed_pair(const hoig_conv_desc *d, const float *xa, const float *xb, const uint16_t *wa_hi,
                                           const uint16_t *wa_lo, const uint16_t *wb_hi, const uint16_t *wb_lo, const float *bias_a,
                                           const float *bias_b, float *ya, float *yb, hoig_stream_t stream) {
    if (!d || !xa || !xb || !wa_hi || !wb_hi || !ya || !yb) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    if (d->transposed || d->stride != 1 || d->R != 3 || d->S != 3) return HOIG_EUNSUPPORTED;
    const PairSet g2{xb, wb_hi, wb_lo, bias_b, nullptr, yb};
    return run(d, xa, wa_hi, wa_lo, bias_a, ya, false, (hipStream_t)stream, nullptr, 0, nullptr, 0, nullptr, nullptr, false, &g2);
}
extern "C" int hoig_conv2d_bwd_data_packed_split_pair(const hoig_conv_desc *d, const uint16_t *dys_a, const uint16_t *dys_b,
                                                      const uint16_t *wta_hi, const uint16_t *wta_lo, const uint16_t *wtb_hi,
                                                      const uint16_t *wtb_lo, const float *addend_a, const float *addend_b, float *dxa,
                                                      float *dxb, hoig_stream_t stream) {
    if (!d || !dys_a || !dys_b || !wta_hi || !wtb_hi || !dxa || !dxb) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    if (d->precision == HOIG_PREC_BF16X3 || d->transposed || d->stride != 1 || d->R != 3 || d->S != 3) return HOIG_EUNSUPPORTED;
    if ((addend_a == nullptr) != (addend_b == nullptr)) return HOIG_EINVAL;
    const PairSet g2{reinterpret_cast<const float *>(dys_b), wtb_hi, wtb_lo, nullptr, addend_b, dxb};
    return run(d, reinterpret_cast<const float *>(dys_a), wta_hi, wta_lo, nullptr, dxa, true, (hipStream_t)stream, nullptr, 0, nullptr, 0,
               addend_a, nullptr, true, &g2);
}

// conv(cat[x1, x2]) and its data gradient [dx1 | dx2] without materialising the concatenation (3x3 stride-1 "same" only)
extern "C" int hoig_conv2d_cat_fwd_packed(const hoig_conv_desc *d, const float *x1, int C1, const float *x2,
                                          const uint16_t *w_hi, const uint16_t *w_lo, const float *bias, float *y,
                                          hoig_stream_t stream) {
    if (!d || !x1 || !x2 || !w_hi || !y) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, x1, w_hi, w_lo, bias, y, false, (hipStream_t)stream, x2, C1, nullptr, 0);
}
extern "C" int hoig_conv2d_cat_bwd_data_packed(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi,
                                               const uint16_t *wt_lo, float *dx1, int C1, float *dx2,
                                               hoig_stream_t stream) {
    if (!d || !dy || !wt_hi || !dx1 || !dx2) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision)) return HOIG_EINVAL;
    return run(d, dy, wt_hi, wt_lo, nullptr, dx1, true, (hipStream_t)stream, nullptr, 0, dx2, C1);
}

// =====================================================================================================================
// Weight gradient on the bf16 MFMA:  dW[co][j] += sum_m P[m][co] * Q[m][j],  j = (r,s,ci), m = pixels.
//   Conv2d          : m walks the OUTPUT grid, P = dy (plain), Q = x gathered at (hp*stride - pad + r, ...)
//   ConvTranspose2d : m walks the INPUT grid (4x fewer pixels, no structural zeros), Q = x (plain),
//                     P = dy gathered at (hi*stride - pad + r, ...) with the column tile's tap (needs Ci % 128 == 0)
// Both operands arrive as [pixel][channel] rows (channel-contiguous, coalesced) but the MFMA wants 8 consecutive
// REDUCTION indices (pixels) per lane, i.e. the transpose.  The tiles are therefore stored as they arrive,
// [m][channel] bf16 with a 320-B row stride, and the fragments are read with ds_read_b64_tr_b16 (the LDS transpose
// read of gfx950; lane semantics verified on hardware by tools/trtest.hip): per 16-lane group a 4(m) x 16(channel)
// block is delivered column-major, two reads give the 8 k-values of one 32x32x16 operand.  Row stride 320 B puts the
// four rows of a block and the two blocks of a 32-lane half on disjoint banks.
namespace {

struct WArgs {
    const float *P, *Q;
    float *DW;
    int gatherP;               // 1: P is the gathered operand (ConvTranspose), 0: Q is (Conv)
    int Bn, Hp, Wp;            // pixel grid walked by m
    int Hg, Wg, Cg;            // gathered tensor
    int Cplain;                // channel count of the plain tensor
    int R, S, stride, pad;
    int M, Co, Ci, K;
    int nblk_n, nblk_mn, m_per_split;
    int lw, lh;                // log2 of Wp / Hp when both are powers of two, else -1
};

struct Pix {
    int b, h, w;
};
__device__ __forceinline__ void pix_advance(Pix &p, int step, int Hp, int Wp) {
    p.w += step;
    while (p.w >= Wp) {
        p.w -= Wp;
        if (++p.h >= Hp) {
            p.h = 0;
            ++p.b;
        }
    }
}

template <int BM, int NSX>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int BN = 128, BK = 32, TM = BM / 64, TN = 2;
    constexpr int RSTR = 320;                              // LDS row stride in bytes (128 bf16 + pad)
    constexpr int PLANE_P = BK * RSTR, PLANE_Q = BK * RSTR;
    constexpr int STAGE = NS * PLANE_P + NB * PLANE_Q;
    constexpr int RP = BM / 32;                            // float4 loads per thread for P (BM/4 columns, 8 row lanes)
    // ONE LDS stage (40 KB): four workgroups (16 waves) share a CU, and their unsynchronised phases cover each other's
    // load / LDS / barrier waits -- measured better than two 80 KB double-buffered workgroups (SQ_WAIT_ANY was 48 %)
    constexpr int NSTAGE = 1;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk_mn);
    const int c0 = (tile / p.nblk_n) * BM, j0 = (tile % p.nblk_n) * BN;
    const int m_begin = blockIdx.y * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);

    // loader roles: Q tile = 32 rows x 32 float4 columns (4 passes of 8 rows); P tile = 32 rows x BM/4 columns
    const int qcol = tid & 31, qrow = tid >> 5;
    constexpr int PCOLS = BM / 4;
    const int pcol = tid % PCOLS, prow = tid / PCOLS;      // BM=128: 8 row lanes, 4 passes; BM=64: 16 row lanes, 2 passes
    constexpr int PROWS = 256 / PCOLS;

    // the tap(s): for the gathered operand.  Conv: per-thread tap from its Q column; ConvT: the tile's tap.
    const int jq = j0 + qcol * 4;
    int tap_r, tap_s, gch;                                 // tap and channel offset inside the gathered tensor
    {
        const int jj = p.gatherP ? j0 : jq;
        const int rs = jj / p.Ci;
        tap_r = rs / p.S;
        tap_s = rs - tap_r * p.S;
        gch = p.gatherP ? (c0 + pcol * 4) : (jq - rs * p.Ci);
    }
    const int plain_ch = p.gatherP ? (jq - (j0 / p.Ci) * p.Ci) : (c0 + pcol * 4);
    const bool q_ok = jq < p.K, p_ok = (c0 + pcol * 4) < p.Co;

    // pixel decode of the reduction index m: shifts/masks when the grid sides are powers of two (every HOGAN layer),
    // divisions otherwise.  The plain operand needs no decode at all: its rows are enumerated exactly like m.
    auto gather = [&](const float *T, int m) -> float4 {
        int bq, h, w;
        if (p.lw >= 0) {
            w = m & (p.Wp - 1);
            h = (m >> p.lw) & (p.Hp - 1);
            bq = m >> (p.lw + p.lh);
        } else {
            const int hw = p.Hp * p.Wp;
            bq = m / hw;
            const int rem = m - bq * hw;
            h = rem / p.Wp;
            w = rem - h * p.Wp;
        }
        const int hg = h * p.stride - p.pad + tap_r, wg = w * p.stride - p.pad + tap_s;
        if (hg < 0 || hg >= p.Hg || wg < 0 || wg >= p.Wg) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4 *>(T + (((size_t)bq * p.Hg + hg) * p.Wg + wg) * p.Cg + gch);
    };
    auto plain = [&](const float *T, int m) -> float4 {
        return *reinterpret_cast<const float4 *>(T + (size_t)m * p.Cplain + plain_ch);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 rp[RP], rq[4];
    auto load_tiles = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb + qrow + 8 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && q_ok) v = p.gatherP ? plain(p.Q, m) : gather(p.Q, m);
            rq[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int m = mb + prow + PROWS * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && p_ok) v = p.gatherP ? gather(p.P, m) : plain(p.P, m);
            rp[i] = v;
        }
    };
    auto store_tiles = [&](int stage) {
        unsigned char *Ph = smem + stage * STAGE, *Pl = Ph + PLANE_P;
        unsigned char *Qh = Ph + NS * PLANE_P, *Ql = Qh + PLANE_Q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 hi, lo;
            split4(rq[i], hi, lo);
            const int off = (qrow + 8 * i) * RSTR + qcol * 8;
            *reinterpret_cast<uint2 *>(Qh + off) = hi;
            if (NB == 2) *reinterpret_cast<uint2 *>(Ql + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            uint2 hi, lo;
            split4(rp[i], hi, lo);
            const int off = (prow + PROWS * i) * RSTR + pcol * 8;
            *reinterpret_cast<uint2 *>(Ph + off) = hi;
            if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
        }
    };

    // transpose-read addressing: 16-lane group g, lane 4q+p -> row (8*(g>>1) + q), channels 16*(g&1) + 4p ..
    const int grp = lane >> 4, li = lane & 15;
    const int tr_off = ((grp >> 1) * 8 + (li >> 2)) * RSTR + ((grp & 1) * 16 + (li & 3) * 4) * 2;
    typedef short s4_t __attribute__((ext_vector_type(4)));
    auto frag = [&](const unsigned char *plane, int chan_base, int ks) -> bf16x8 {
        const unsigned char *a = plane + tr_off + ks * 16 * RSTR + chan_base * 2;
        const s4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)a);
        const s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a + 4 * RSTR));
        bf16x8 f;
        f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
        f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
        return f;
    };

    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles(0);
    }
    __syncthreads();
    int cur = 0;
    for (int mb = m_begin; mb < m_end; mb += BK) {
        const bool nxt = mb + BK < m_end;
        if (nxt) load_tiles(mb + BK);
        const unsigned char *Ph = smem + cur * STAGE, *Pl = Ph + PLANE_P;
        const unsigned char *Qh = Ph + NS * PLANE_P, *Ql = Qh + PLANE_Q;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = frag(Ph, wm * (TM * 32) + i * 32, ks);
                if (NS == 2) al[i] = frag(Pl, wm * (TM * 32) + i * 32, ks);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = frag(Qh, wn * 64 + j * 32, ks);
                if (NB == 2) bl[j] = frag(Ql, wn * 64 + j * 32, ks);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (NS == 2)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        if (NB == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (NSTAGE == 2) {
            if (nxt) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();                  // every wave is done reading the stage
            if (nxt) store_tiles(0);
            __syncthreads();
        }
    }

    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = c0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co >= p.Co) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int jj = j0 + wn * 64 + j * 32 + l31;
                if (jj < p.K) atomicAdd(&p.DW[(size_t)co * p.K + jj], acc[i][j][r]);
            }
        }
}

template <int BM>
int launch_wgrad_bf16(WArgs a, int ns, hipStream_t st) {
    const int nbm = (int)hoig_cdiv(a.Co, BM), nbn = (int)hoig_cdiv(a.K, 128);
    a.nblk_n = nbn;
    a.nblk_mn = nbm * nbn;
    constexpr int target_blocks = 512;
    int splits = (int)hoig_cdiv(target_blocks, a.nblk_mn);
    // every split adds |dW| fp32 atomics (~235 G/s chip-wide, i.e. as slow as the MFMA work of ~2000 pixels) while fewer
    // than ~2 workgroups per CU leave SIMDs idle: measured optimum ~512 workgroups (sweep 512/1024/2048: 32.7/33.1/33.1 ms
    // of weight gradients per step), splits of at least 512 pixels
    constexpr int min_px = 512;
    // tiny K (the SPADE label convs: 12 channels x 9 taps): one column tile, next to no atomics, and a workgroup's pixel loop
    // is pure load latency -- split four times finer
    constexpr int small_k_px = 128;
    const int max_splits = (int)hoig_cdiv(a.M, a.K <= 128 ? (small_k_px < min_px ? small_k_px : min_px) : min_px);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int mps = (int)hoig_cdiv(hoig_cdiv(a.M, splits), 32) * 32;
    a.m_per_split = mps;
    splits = (int)hoig_cdiv(a.M, mps);
    dim3 grid(a.nblk_mn, splits);
    HOIG_NS_SWITCH(ns, wgrad_bf16_kernel<BM, NSX><<<grid, 256, 0, st>>>(a));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of stride-1 "same" 3x3 convolutions with an LDS-resident INPUT HALO.  The kernel above re-reads both
// operands from L2 for every (co tile, tap, ci tile) pair -- 30 KB per MFLOP, which at full matrix rate would need
// ~42 B/clk/CU from a 64 B/clk load path -- and is bound there.  Here a workgroup owns dW[64 co][9 taps][32 ci]: per
// m-tile (2 rows x 32 output pixels) it stages dy[64 px][64 co] and the x halo [4 x 34 px][32 ci] ONCE (split to bf16
// hi/lo, rows as they arrive) and all nine taps read their x fragments out of the same halo image at a tap-dependent
// row offset: 14 KB per MFLOP.  Six waves: wave = (co half, tap row); each accumulates 32 co x 32 ci for the three taps
// of its row (48 accumulator registers, so three workgroups = 18 waves share a CU).
// Both operands want pixels along k, so both are read with ds_read_b64_tr_b16; the halo rows are 64 B apart (no pad):
// the four rows a 32-lane half reads (256 B) cover all 64 banks once for any row offset.  dy rows are 192 B apart.

// KS = 3: "same" 3x3 (pad 1, input = output size).  KS = 5: the attention's 5x5 VALID convolution over the replicate-padded
// target (input (H+4) x (W+4), pad 0): ten waves = co half x tap row, five taps each.
// CM = 2: 128 output channels per workgroup on twice the waves (wave = co quarter x tap row): the same work per wave, but the x
// halo is loaded and split once for twice the MFMAs -- the kernel is short of VALU issue slots, not of clock.
// S2: Conv2d stride 2, pad 1 (KS = 3): the output tile's 2 x 32 pixels read x at (2y + r - 1, 2x + s - 1), a 5 x 65 halo.  It is
// stored split by COLUMN PARITY -- row index ((hy * 2 + (hx & 1)) * 33 + (hx >> 1)) -- so that tap (r, s) reads sixteen
// consecutive output pixels at sixteen consecutive rows again (parity s & 1, first row (s >> 1)): the transpose reads stay
// unit-stride and conflict-free, exactly as for stride 1.
// TH_ = 4: pixel tiles of 4 x 32 (one workgroup per CU, 94 KB of dynamic LDS): half the barriers, staging rounds and read ramps per
// MFMA, and a 6-row halo for 4 rows instead of two 4-row halos.
template <int NSX, int KS, int CM, bool S2 = false, int TH_ = 2>
__global__ __launch_bounds__(128 * KS * CM) void wgrad_halo_bf16_kernel(const WHaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: A (activations / dy), B (weights / x)
    constexpr int TH = TH_, TW = 32, BM = 64 * CM, BC = 32, NT = 128 * KS * CM;
    constexpr int CQ = 2 * CM, C4 = 16 * CM;               // 32-channel groups / float4s of a dy pixel row
    constexpr int SD = S2 ? 2 : 1;
    constexpr int HH = SD * (TH - 1) + KS, HWID = SD * (TW - 1) + KS, HPIX = HH * HWID;     // 4 x 34 (stride 2: 5 x 65) halo pixels
    constexpr int HWP = (HWID + 1) / 2;                    // stride 2: pixels per column-parity run
    constexpr int HROWS = S2 ? HH * 2 * HWP : HPIX;        // rows of the LDS halo image
    constexpr int PSTR = 128 * CM + 64, QSTR = 64;         // (192 / 320 B: four consecutive rows cover the 64 banks once)
    constexpr int PLANE_P = TH * TW * PSTR, PLANE_Q = ((HROWS * QSTR + 255) / 256) * 256;
    constexpr int LDS_BYTES = NS * PLANE_P + NB * PLANE_Q;
    __shared__ __attribute__((aligned(16))) unsigned char smem_static[TH_ > 2 ? 16 : LDS_BYTES];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dynamic[];
    unsigned char *smem = TH_ > 2 ? smem_dynamic : smem_static;
    unsigned char *Ph = smem, *Pl = smem + PLANE_P;
    unsigned char *Qh = smem + NS * PLANE_P, *Ql = Qh + PLANE_Q;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave % CQ, tr = wave / CQ;              // 32-channel group of co, tap row
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int c0 = (tile / p.nblk_ci) * BM, ci0 = (tile % p.nblk_ci) * BC;
    const int mt_begin = blockIdx.y * p.mt_per_split;
    const int mt_end = min(p.n_mtiles, mt_begin + p.mt_per_split);

    constexpr int PSL = (TH * TW * C4 + NT - 1) / NT;      // dy float4s per thread (3)
    constexpr int QSL = (HPIX * 8 + NT - 1) / NT;          // halo float4s per thread (3)
    float4 rp[PSL], rq[QSL];
    const bool do_bias = p.DB != nullptr && ci0 == 0;      // the workgroups of the first ci tile also own the bias gradient
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tiles = [&](int mt) {
        const int tx = mt % p.tiles_x;
        const int t2 = mt / p.tiles_x;
        const int ty = t2 % p.tiles_y, b = t2 / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        const float *dyb = p.DY + (((size_t)b * p.H + y0) * p.W + x0) * p.Co + c0;
#pragma unroll
        for (int i = 0; i < PSL; ++i) {
            const int idx = tid + NT * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TH * TW * C4) {
                const int pp = idx / C4, c4 = idx % C4;    // pixel of the tile: row pp>>5, column pp&31
                v = *reinterpret_cast<const float4 *>(dyb + ((size_t)(pp >> 5) * p.W + (pp & 31)) * p.Co + c4 * 4);
            }
            rp[i] = v;
            if (do_bias) {                 // this thread always loads the same four channels (NT % C4 == 0)
                bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
            }
        }
        const bool second = p.X2 != nullptr && ci0 >= p.ci1;
        const int ldx = p.X2 ? (second ? p.Ci - p.ci1 : p.ci1) : p.Ci;
        const float *xb = (second ? p.X2 : p.X) + (size_t)b * p.Hin * p.Win * ldx + (second ? ci0 - p.ci1 : ci0);
#pragma unroll
        for (int i = 0; i < QSL; ++i) {
            const int idx = tid + NT * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < HPIX * 8) {
                const int hp = idx >> 3, c4 = idx & 7;
                const int hy = hp / HWID, hx = hp - hy * HWID;
                const int gy = y0 * SD - p.pad + hy, gx = x0 * SD - p.pad + hx;
                if (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win)
                    v = *reinterpret_cast<const float4 *>(xb + ((size_t)gy * p.Win + gx) * ldx + c4 * 4);
            }
            rq[i] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < PSL; ++i) {
            const int idx = tid + NT * i;
            if (idx < TH * TW * C4) {
                uint2 hi, lo;
                if (HOIG_WG_KO & 1) {
                    hi = make_uint2(__float_as_uint(rp[i].x) >> 16 | (__float_as_uint(rp[i].y) & 0xffff0000u),
                                    __float_as_uint(rp[i].z) >> 16 | (__float_as_uint(rp[i].w) & 0xffff0000u));
                    lo = make_uint2(0u, 0u);
                } else
                    split4(rp[i], hi, lo);
                const int off = (idx / C4) * PSTR + (idx % C4) * 8;
                *reinterpret_cast<uint2 *>(Ph + off) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < QSL; ++i) {
            const int idx = tid + NT * i;
            if (idx < HPIX * 8) {
                uint2 hi, lo;
                if (HOIG_WG_KO & 1) {
                    hi = make_uint2(__float_as_uint(rq[i].x) >> 16 | (__float_as_uint(rq[i].y) & 0xffff0000u),
                                    __float_as_uint(rq[i].z) >> 16 | (__float_as_uint(rq[i].w) & 0xffff0000u));
                    lo = make_uint2(0u, 0u);
                } else
                    split4(rq[i], hi, lo);
                int st = idx * 8;
                if (S2) {
                    const int hp = idx >> 3, hy = hp / HWID, hx = hp - hy * HWID;
                    st = ((hy * 2 + (hx & 1)) * HWP + (hx >> 1)) * QSTR + (idx & 7) * 8;
                }
                *reinterpret_cast<uint2 *>(Qh + st) = hi;
                if (NB == 2) *reinterpret_cast<uint2 *>(Ql + st) = lo;
            }
        }
    };

    // transpose-read addressing (see wgrad_bf16_kernel): 16-lane group g, lane 4q+c -> row 8*(g>>1)+q, channels 16*(g&1)+4c
    const int grp = lane >> 4, li = lane & 15;
    const int trow = (grp >> 1) * 8 + (li >> 2), tch = ((grp & 1) * 16 + (li & 3) * 4) * 2;
    const int trP = trow * PSTR + tch + cb * 64, trQ = (trow + (S2 ? 0 : tr * HWID)) * QSTR + tch;
    typedef short s4_t __attribute__((ext_vector_type(4)));
    auto frag = [&](const unsigned char *a, int stride4) -> bf16x8 {
        const s4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)a);
        const s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a + stride4));
        bf16x8 f;
        f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
        f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
        return f;
    };

    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#ifdef HOIG_STAMP
    unsigned long long w_issue = 0, w_comp = 0, w_b1 = 0, w_st = 0, w_b2 = 0;
    const unsigned long long wt_begin = clock64(), wrt_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long wt_prev = wt_begin;
#if HOIG_STAMP == 2
#define WSTAMP(v)
#else
#define WSTAMP(v) { const unsigned long long t_ = clock64(); v += t_ - wt_prev; wt_prev = t_; }
#endif
#else
#define WSTAMP(v)
#endif
    if (mt_begin < mt_end) {
        load_tiles(mt_begin);
        store_tiles();
    }
    __syncthreads();
    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const bool nxt = mt + 1 < mt_end;
        if (nxt && !(HOIG_WG_KO & 2)) load_tiles(mt + 1);
        WSTAMP(w_issue)
        // the fragments of k-step kk + 1 are read before the MFMAs of k-step kk are issued (fences: the compiler would sink the reads
        // below the MFMAs to shorten live ranges, and every k-step would then start with an exposed LDS round trip)
        struct KFrag { bf16x8 ah, al, bh[KS], bl[KS]; };
        auto read_k = [&](KFrag &f, int kk) {
            const int prow0 = kk * 16;
            const int qrow0 = S2 ? (kk & 1) * 16 : (kk >> 1) * HWID + (kk & 1) * 16;
            if (HOIG_WG_KO & 16) {
                for (int q = 0; q < 8; ++q) { f.ah[q] = (short)(0x3f80 + lane + q); f.al[q] = (short)(0x3c00 + lane * 3 + q); }
                for (int t = 0; t < KS; ++t)
                    for (int q = 0; q < 8; ++q) f.bh[t][q] = f.bl[t][q] = (short)(0x3f00 + lane * 5 + q + t);
                return;
            }
            f.ah = frag(Ph + trP + prow0 * PSTR, 4 * PSTR);
            if (NS == 2) f.al = frag(Pl + trP + prow0 * PSTR, 4 * PSTR);
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int qoff = trQ + (S2 ? ((2 * (kk >> 1) + tr) * 2 + (t & 1)) * HWP + qrow0 + (t >> 1) : qrow0 + t) * QSTR;
                f.bh[t] = frag(Qh + qoff, 4 * QSTR);
                if (NB == 2) f.bl[t] = frag(Ql + qoff, 4 * QSTR);
            }
        };
        auto mma_k = [&](const KFrag &f) {
            // term-major: the KS accumulators take turns, so no MFMA waits on the one issued just before it
            if (NS == 2 && !(HOIG_WG_KO & 32)) {
#pragma unroll
                for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[t], acc[t], 0, 0, 0);
            }
            if (NB == 2) {
#pragma unroll
                for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bl[t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[t], acc[t], 0, 0, 0);
        };
        constexpr bool AHEAD = KS == 3 && NB == 1;         // (5x5, and x split too: two fragment sets do not fit the registers)
        KFrag f0, f1;
        if (AHEAD) read_k(f0, 0);
#pragma unroll
        for (int kk = 0; kk < TH * 2; kk += 2) {           // 16 consecutive pixels of one tile row per k-step
            if (AHEAD) {
                read_k(f1, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_k(f0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < TH * 2) read_k(f0, kk + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma_k(f1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                read_k(f0, kk);
                mma_k(f0);
                read_k(f0, kk + 1);
                mma_k(f0);
            }
        }
        WSTAMP(w_comp)
        __syncthreads();                      // every wave is done reading the stage
        WSTAMP(w_b1)
        if (nxt && !(HOIG_WG_KO & 4)) store_tiles();
        WSTAMP(w_st)
        __syncthreads();
        WSTAMP(w_b2)
    }
#ifdef HOIG_STAMP
    const unsigned long long wt_loop = clock64() - wt_begin;
#endif

    if (do_bias) {                         // 24 threads hold partial sums of the same four channels: combine in LDS
        float *red = reinterpret_cast<float *>(smem);          // (the tiles are dead: the loop ended with a barrier)
        if (tid < BM) red[tid] = 0.f;
        __syncthreads();
        const int ch = (tid % C4) * 4;
        atomicAdd(&red[ch + 0], bsum.x);
        atomicAdd(&red[ch + 1], bsum.y);
        atomicAdd(&red[ch + 2], bsum.z);
        atomicAdd(&red[ch + 3], bsum.w);
        __syncthreads();
        if (tid < BM) atomicAdd(&p.DB[c0 + tid], red[tid]);
    }
    const int l31 = lane & 31, lh = lane >> 5;
    if ((HOIG_WG_KO & 8) && acc[0][0] != 12345.f) return;
    if (!p.tout) {
        const int K = KS * KS * p.Ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = c0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float *row = p.DW + (size_t)co * K + (tr * KS) * p.Ci + ci0 + l31;
#pragma unroll
            for (int t = 0; t < KS; ++t) atomicAdd(row + t * p.Ci, acc[t][r]);
        }
    } else {
        // packed [gathered channel q][tap][plain channel pc]: the accumulator has q on the lanes, so adding it as it stands
        // would spread every atomic instruction over 32 rows of DW (K * 4 bytes apart).  Each wave turns its 32 x 32 tile
        // through LDS first (the operand tiles are dead: the loop ended with a barrier; the bias path is not used here), so
        // that the lanes of an atomic instruction cover two contiguous 128-B runs of plain channels.
        const int K = KS * KS * p.Co;
        float *tile = reinterpret_cast<float *>(smem) + wave * (32 * 33);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = acc[t][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0): this wave's LDS writes have landed (no other wave reads them)
            __builtin_amdgcn_wave_barrier();
            const int pc = c0 + cb * 32 + l31;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int q = 2 * k + lh;
                atomicAdd(p.DW + (size_t)(ci0 + q) * K + (tr * KS + t) * p.Co + pc, tile[l31 * 33 + q]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#ifdef HOIG_STAMP
    if (p.dbg && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);         // (the atomics are fire-and-forget: this only bounds their ISSUE)
        unsigned long long *d = p.dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + wave) * 8;
        d[0] = w_issue; d[1] = w_comp; d[2] = w_b1; d[3] = w_st; d[4] = w_b2; d[5] = wt_loop;
        d[6] = clock64() - wt_begin; d[7] = __builtin_amdgcn_s_memrealtime() - wrt_begin;
    }
#endif
}

struct WPairSet {
    const float *x, *dy;
    float *dw;
};
int launch_wgrad_halo(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, float *dbias, int ns,
                      hipStream_t st, const float *x2 = nullptr, int ci1 = 0, bool dy_split = false, const WPairSet *g2 = nullptr) {
    WHaloArgs a;
    a.DY = dy; a.X = x; a.DW = dw; a.DB = dbias; a.X2 = x2; a.ci1 = ci1;
    a.Bn = d->B; a.H = d->Ho; a.W = d->Wo; a.Co = d->Co; a.Ci = d->Ci;
    a.b_split = 0; a.DY_g2 = a.X_g2 = nullptr; a.DW_g2 = nullptr;
    if (g2) {                  // grouped launch: both problems' pixel tiles in one grid (LDS-DMA kernel only: dy_split)
        if (!dy_split || d->transposed || x2) return HOIG_EUNSUPPORTED;
        a.Bn = 2 * d->B; a.b_split = d->B;
        a.DY_g2 = g2->dy; a.X_g2 = g2->x; a.DW_g2 = g2->dw;
    }
    a.Hin = d->Hi; a.Win = d->Wi; a.pad = d->pad;
    a.tout = 0;
    if (d->transposed) {       // dW[ci][co][r][s] = sum_i x[i] dy[2i - 1 + (r,s)]: x is the plain operand, dy the gathered one
        a.DY = x; a.X = dy; a.DB = nullptr;
        a.H = d->Hi; a.W = d->Wi; a.Co = d->Ci; a.Ci = d->Co;
        a.Hin = d->Ho; a.Win = d->Wo;
        a.tout = 1;
    }
    const bool s2 = d->stride == 2;
    a.nblk_ci = a.Ci / 32;
    constexpr int cm_env = 2;
    const int cm = (d->R == 3 && cm_env == 2 && a.Co % 128 == 0) ? 2 : 1;      // (a.Co: channels of the plain operand)
    a.nblk = (a.Co / (64 * cm)) * a.nblk_ci;
    // 4-row pixel tiles for the stride-1 3x3 layers on 128-channel workgroups, where every workgroup still gets >= 8 of them
    constexpr int th_env = 4;
    const bool th4 = th_env == 4 && cm == 2 && d->R == 3 && !s2 && !d->transposed && ns != 2 && a.H % 4 == 0 &&
                     ((int64_t)a.Bn * (a.W / 32) * (a.H / 4) * a.nblk >= 8 * 256 || dy_split);
    if (dy_split && (!th4 || dbias)) return HOIG_EUNSUPPORTED;      // (pre-split dy: the LDS-DMA kernel only, wgrad_dma.hip)
    a.tiles_x = a.W / 32;
    a.tiles_y = a.H / (th4 ? 4 : 2);
    a.n_mtiles = a.Bn * a.tiles_x * a.tiles_y;
    constexpr int target_blocks = 512;
    // 5x5: a workgroup owns 25 taps x 64 x 32 outputs, so every pixel split costs 2.8x the atomics of a 3x3 one: 256 (measured)
    // the 8-image 32 x 32 launches (64 four-row tiles) run side by side on two branch streams in G's backward: 128 workgroups each --
    // half the pixel splits, half the atomics -- instead of 256 (round 4: step -0.35 ms, profiles/r04_wflat5_ab.txt)
    const int target = (th4 && a.n_mtiles <= 64) ? target_blocks / 2 : target_blocks;
    int splits = (int)hoig_cdiv((d->R == 5 ? target / 2 : target) / cm, a.nblk);
    if (splits > a.n_mtiles) splits = a.n_mtiles;
    if (splits < 1) splits = 1;
    if (g2) splits = 2 * (int)hoig_cdiv(splits, 2);                       // (the same number of tile ranges in either problem)
    a.mt_per_split = (int)hoig_cdiv(a.n_mtiles, splits);
    splits = (int)hoig_cdiv(a.n_mtiles, a.mt_per_split);
    dim3 grid(a.nblk, splits);
#ifdef HOIG_STAMP
    a.dbg = g_stamp_buf;
#endif
    if (dy_split) return launch_wgrad_dma(a, ns, grid, st);
    if (th4) {
        constexpr int LDS4 = 2 * (4 * 32 * 320) + (((6 * 34 * 64) + 255) / 256) * 256;      // dy hi, lo | x hi
        static hoig_once once;
        if (!once.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_halo_bf16_kernel<1, 3, 2, false, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS4) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_halo_bf16_kernel<3, 3, 2, false, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS4) != hipSuccess)
                return HOIG_ELAUNCH;
            once.set();
        }
        if (ns == 3) wgrad_halo_bf16_kernel<3, 3, 2, false, 4><<<grid, 768, LDS4, st>>>(a);
        else wgrad_halo_bf16_kernel<1, 3, 2, false, 4><<<grid, 768, LDS4, st>>>(a);
    } else if (s2 && cm == 2) HOIG_NS_SWITCH(ns, wgrad_halo_bf16_kernel<NSX, 3, 2, true><<<grid, 768, 0, st>>>(a));
    else if (s2) HOIG_NS_SWITCH(ns, wgrad_halo_bf16_kernel<NSX, 3, 1, true><<<grid, 384, 0, st>>>(a));
    else if (d->R == 5) HOIG_NS_SWITCH(ns, wgrad_halo_bf16_kernel<NSX, 5, 1><<<grid, 640, 0, st>>>(a));
    else if (cm == 2) HOIG_NS_SWITCH(ns, wgrad_halo_bf16_kernel<NSX, 3, 2><<<grid, 768, 0, st>>>(a));
    else HOIG_NS_SWITCH(ns, wgrad_halo_bf16_kernel<NSX, 3, 1><<<grid, 384, 0, st>>>(a));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace

bool hoig_conv_bf16_wgrad_fuses_bias(const hoig_conv_desc *d) {
    if (d->precision == HOIG_PREC_F32 || d->R != d->S) return false;
    if (d->transposed)           // ConvTranspose2d 3x3 stride 2 pad 1 output_padding 1: the same kernel with x and dy swapped
        return d->stride == 2 && d->R == 3 && d->pad == 1 && d->Ho == 2 * d->Hi && d->Wo == 2 * d->Wi &&
               d->Wi % 32 == 0 && d->Hi % 2 == 0 && d->Co % 32 == 0 && d->Ci % 64 == 0;
    if (d->Wo % 32 || d->Ho % 2 || d->Ci % 32 || d->Co % 64) return false;
    if (d->stride == 2)          // Conv2d 3x3 stride 2 pad 1 on the column-parity halo (the generator's down-sampling layers)
        return d->R == 3 && d->pad == 1 && d->Hi == 2 * d->Ho && d->Wi == 2 * d->Wo;
    if (d->stride != 1) return false;
    if (d->R == 3) return d->pad == 1 && d->Hi == d->Ho && d->Wi == d->Wo;
    return d->R == 5 && d->pad == 0 && d->Hi == d->Ho + 4 && d->Wi == d->Wo + 4;      // the attention's valid 5x5
}

// dbias: only passed (non-null) when hoig_conv_bf16_wgrad_fuses_bias(d); every other shape gets its bias gradient from
// hoig_colsum_accum in the caller
int hoig_conv_bf16_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, float *dbias, hipStream_t st) {
    if ((d->Co & 3) || (d->Ci & 3) || d->Co < 32) return HOIG_EUNSUPPORTED;
    WArgs a;
    a.DW = dw;
    a.Bn = d->B;
    a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad;
    a.Co = d->Co; a.Ci = d->Ci; a.K = d->R * d->S * d->Ci;
    if (!d->transposed) {
        a.gatherP = 0;
        a.P = dy; a.Q = x;
        a.Hp = d->Ho; a.Wp = d->Wo;
        a.Hg = d->Hi; a.Wg = d->Wi; a.Cg = d->Ci;
        a.Cplain = d->Co;
    } else {
        if (d->Ci % 128) return HOIG_EUNSUPPORTED;
        a.gatherP = 1;
        a.P = dy; a.Q = x;
        a.Hp = d->Hi; a.Wp = d->Wi;
        a.Hg = d->Ho; a.Wg = d->Wo; a.Cg = d->Co;
        a.Cplain = d->Ci;
    }
    a.M = d->B * a.Hp * a.Wp;
    a.lw = a.lh = -1;
    if ((a.Hp & (a.Hp - 1)) == 0 && (a.Wp & (a.Wp - 1)) == 0) {
        a.lw = __builtin_ctz(a.Wp);
        a.lh = __builtin_ctz(a.Hp);
    }
    const int ns = ns_of_precision(d->precision);
    // the attention's valid 5x5 convolutions on the flattened-axis kernel (wgrad_flat.hip)
    // (1: where the 2 x 32-pixel halo kernel cannot run -- output widths that are not multiples of 32: 193 -> 118 us on the
    // source-side convolution; 2: also where it can -- measured 10 % slower there, profiles/r04_wflat5_ab.txt)
    if (!d->transposed && d->stride == 1 && d->R == 5 && d->S == 5 && d->pad == 0 && d->Ho == d->Hi - 4 && d->Wo == d->Wi - 4 &&
        (hoig_tuning(HOIG_TUNE_WFLAT5) >= 2 || (hoig_tuning(HOIG_TUNE_WFLAT5) == 1 && !hoig_conv_bf16_wgrad_fuses_bias(d)))) {
        const int rc = launch_wgrad_flat5(x, dy, dw, dbias, d->B, d->Hi, d->Wi, d->Ci, d->Co, ns, st);
        if (rc != HOIG_EUNSUPPORTED) return rc;
    }
    if (hoig_conv_bf16_wgrad_fuses_bias(d)) return launch_wgrad_halo(d, x, dy, dw, dbias, ns, st);
    if (a.Co <= 64) return launch_wgrad_bf16<64>(a, ns, st);
    return launch_wgrad_bf16<128>(a, ns, st);
}

// weight gradient from PRE-SPLIT dy (include/hoig_kernels.h): the stride-1 "same" 3x3 layers the LDS-DMA kernel covers
extern "C" int hoig_conv2d_bwd_weight_split(const hoig_conv_desc *d, const float *x, const uint16_t *dy_split, float *dw,
                                            hoig_stream_t stream) {
    if (!d || !x || !dy_split || !dw) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision) || d->precision == HOIG_PREC_BF16X3 || d->transposed || d->stride != 1 || d->R != 3 ||
        d->S != 3 || !hoig_conv_bf16_wgrad_fuses_bias(d) || (d->Co & 127))
        return HOIG_EUNSUPPORTED;
    return launch_wgrad_halo(d, x, reinterpret_cast<const float *>(dy_split), dw, nullptr, ns_of_precision(d->precision),
                             (hipStream_t)stream, nullptr, 0, true);
}

extern "C" int hoig_conv2d_bwd_weight_split_pair(const hoig_conv_desc *d, const float *xa, const float *xb, const uint16_t *dys_a,
                                                 const uint16_t *dys_b, float *dwa, float *dwb, hoig_stream_t stream) {
    if (!d || !xa || !xb || !dys_a || !dys_b || !dwa || !dwb) return HOIG_EINVAL;
    if (!is_16bit_precision(d->precision) || d->precision == HOIG_PREC_BF16X3 || d->transposed || d->stride != 1 || d->R != 3 ||
        d->S != 3 || !hoig_conv_bf16_wgrad_fuses_bias(d) || (d->Co & 127))
        return HOIG_EUNSUPPORTED;
    const WPairSet g2{xb, reinterpret_cast<const float *>(dys_b), dwb};
    return launch_wgrad_halo(d, xa, reinterpret_cast<const float *>(dys_a), dwa, nullptr, ns_of_precision(d->precision),
                             (hipStream_t)stream, nullptr, 0, true, &g2);
}

// weight gradient of conv(cat[x1, x2]) (3x3 stride-1 "same", bf16 halo kernel only)
extern "C" int hoig_conv2d_cat_bwd_weight(const hoig_conv_desc *d, const float *x1, int C1, const float *x2, const float *dy,
                                          float *dw, float *dbias, hoig_stream_t stream) {
    if (!d || !x1 || !x2 || !dy || !dw) return HOIG_EINVAL;
    if (!hoig_conv_bf16_wgrad_fuses_bias(d) || d->R != 3 || C1 % 32 || C1 <= 0 || C1 >= d->Ci) return HOIG_EUNSUPPORTED;
    const int ns = ns_of_precision(d->precision);
    return launch_wgrad_halo(d, x1, dy, dw, dbias, ns, (hipStream_t)stream, x2, C1);
}
