// Weight gradients of the 16-bit convolution path on v_mfma_f32_32x32x16_bf16 (split off conv_igemm_bf16.hip in round 6: that file
// keeps the forward / data-gradient kernels and the weight packing): the generic pixel-split kernel (wgrad_bf16_kernel), the halo
// kernels of the 3x3 / 5x5 / stride-2 layers (wgrad_halo_bf16_kernel) and their launchers; the LDS-DMA kernel for pre-split dy is
// wgrad_dma.hip, the flattened-axis one wgrad_flat.hip.
#include "conv_bf16_common.h"
#include "tuning.h"
#include <cstdlib>

// tools/wgrad_knockout.cpp builds this file with HOIG_WG_KO != 0 to time wgrad_halo_bf16_kernel with parts removed (results are
// then wrong): 1 no hi/lo split (bit moves only), 2 no global loads after the first tile, 4 no LDS stores after the first tile,
// 8 no atomic epilogue, 16 no LDS fragment reads (MFMAs on register garbage), 32 no MFMAs of the lo plane
#ifndef HOIG_WG_KO
#define HOIG_WG_KO 0
#endif

namespace {
using namespace hoig_detail;
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
            rp[i] = v;       // (the bias sum takes it in store_tiles: summed HERE, hipcc turns the branch into a select and every
                             //  workgroup waits for the loads it has just issued -- vmcnt(0) in front of the step's MFMAs)
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
            if (do_bias) {                 // this thread always holds the same four channels (NT % C4 == 0); idle slices hold zeros
                bsum.x += rp[i].x; bsum.y += rp[i].y; bsum.z += rp[i].z; bsum.w += rp[i].w;
            }
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

    if (mt_begin < mt_end) {
        load_tiles(mt_begin);
        store_tiles();
    }
    __syncthreads();
    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const bool nxt = mt + 1 < mt_end;
        if (nxt && !(HOIG_WG_KO & 2)) load_tiles(mt + 1);
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
        __syncthreads();                      // every wave is done reading the stage
        if (nxt && !(HOIG_WG_KO & 4)) store_tiles();
        __syncthreads();
    }

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
