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
// (MODE 2 -- 8 rows x 32 pixels, weight tiles double-buffered, halo single -- was this kernel's form of the tile every large launch
//  takes; since round 4 those launches run on v_mfma_f32_16x16x32 (conv_halo16.hip: conv_halo3_m16_kernel) and round 6 removed the
//  32x32x16 instantiations together with the tuning key `mfma16` that selected them)
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
    static_assert(MODE == 0 || MODE == 1, "the 8-row tilings live in conv_halo16.hip");
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
    halo_load(0);
    halo_store(0);
    load_b(0);
    store_b(0);
    if (DB) {
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
            compute(r, 0, 0);
            if (more) {
                __syncthreads();                  // every wave has finished reading the weight tiles (and the halo)
                if (boundary) halo_store(0);
                store_b(0);
                __syncthreads();
            }
        }
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


int launch_halo3(HaloArgs a, int ns, hipStream_t st) {
    a.tiles_x = a.W / 32;
    a.tiles_y = a.H / 4;
    a.nmajor = 1;
    const bool n64 = (a.N % 128) != 0 || (a.C2 && a.n1 % 128 != 0);      // (a channel tile must not straddle the two outputs)
    a.nblk_n = (int)hoig_cdiv(a.N, n64 ? 64 : 128);
    a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
    if (n64) {
        // 64-channel layers (the full-resolution levels): 8 x 32 x 64 on the 16x16 MFMA where that leaves every CU a workgroup
        if (a.H % 8 == 0 && a.nblk / 2 >= 256) {
            a.tiles_y = a.H / 8;
            a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
            return launch_halo3_m16(a, ns, 64, st);
        }
        if (a.a_split || a.b_split || a.in_scale) return HOIG_EUNSUPPORTED;  // (pre-split input, grouped launch: conv_halo16.hip only)
        HOIG_NS_SWITCH(ns, return launch_halo3_one<NSX, 2, 2, 64, 0>(a, st));
    }
    if (a.H % 8 == 0 && a.nblk / 2 < 256 && a.nblk >= 192 && a.N % 128 == 0) {
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
        return launch_halo3_m16(a, ns, 64, st);
    }
    // 8 x 32 pixel tiles (8 waves, weight tile shared by 256 pixels) when that still gives every CU a workgroup
    if (a.H % 8 == 0 && a.nblk / 2 >= 256) {
        a.tiles_y = a.H / 8;
        a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
        return launch_halo3_m16(a, ns, 128, st);
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
    if (in && (dgrad || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->transposed || a_split || g2))
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

