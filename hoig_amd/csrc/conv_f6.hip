// 3x3 stride-1 "same" convolution FORWARD on  fp16 (hi*hi)  +  two block-scaled fp6 cross terms  (HOIG_PREC_F16F6).
//
// Every product a*w of the split arithmetic is  ah*wh + al*wh + ah*wl  (a = ah + al, w*2^8 = wh + wl, fp16 halves).
// HOIG_PREC_BF16X3 issues three v_mfma_f32_32x32x16_f16 for it.  The two cross terms carry 2^-11 of the product, so 4
// significant bits are enough for them (measured: DESIGN.md section 4 -- tools/emulate_split_terms.py predicts 3e-4 on the
// six forward outputs against north_star's 1e-3; dropping them costs 4.6e-3, i.e. they cannot be dropped).  Here they run
// on v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 operands and one E8M0 scale per (row, 32 channels): K = 64 per instruction at
// 3.4x the bf16 FLOP rate (tools/mfma_f6_probe.hip, measured), so a product costs 4 + 2 * 1.17 = 6.3 bf16-MFMA units per 64
// channels instead of 12.
//
// Operand layouts (all checked on hardware with exact integers, tools/mfma_f6_probe.hip / tools/cvt_f6_probe.hip):
//   * MFMA operand: lane l holds row (A) / column (B) l & 31 and the 32 consecutive k of block l >> 5, element j in bits
//     [6j, 6j+6) of the lane's first six dwords; the scale VGPR's byte 0 is that lane's E8M0 exponent.
//   * v_cvt_scalef32_pk32_fp6_f16 turns 32 fp16 into exactly that 24-byte fragment (value / scale, RNE, saturating at 7.5).
// So a fragment is one (pixel | output channel, 32-channel block) RECORD: 24 B block 0 | 24 B block 1 | 2 scale bytes | pad
// = 56 B (14 banks: conflict-free 8-B reads across 32 lanes), in LDS and -- for the weights -- in HBM alike.
// The block scale is 2^(e-2), e = exponent of the block's largest |value| (largest element in [4, 8)); the residual's scale is
// 2^(e-13): |lo| <= 2^(e-11) by construction of the fp16 split, so no second reduction is needed.
//
// Tile / pipeline: the 8x32-pixel x 128-channel tile of conv_halo3_bf16_kernel MODE 2 (8 waves, one workgroup per CU), but
// over 64-channel blocks; a step = one tap: fp16 weight tile 16 KB + two fp6 record arrays 7 KB each, double-buffered
// (register-staged one step ahead); the fp32 halo (10 x 34 pixels x 64 channels) is split once per block into the fp16 plane
// (144-B rows) and the two fp6 record arrays, by threads that each own one (pixel, 32-channel block).  148 KB of LDS.
#include "common.h"
#include <cstdlib>

// tools/f6_knockout.cpp builds this file with HOIG_F6_KO != 0 to time the kernel with parts removed (results are then wrong):
// 1 no halo split/store after the first block, 2 no fp6 terms, 4 no fp16 term, 8 no halo loads after the first block,
// 16 no weight loads/stores after the prologue, 32 no per-step barrier
#ifndef HOIG_F6_KO
#define HOIG_F6_KO 0
#endif
// 1 (default since round 6): the weight stages go global -> LDS by LDS-DMA (global_load_lds_dwordx4 from inline asm with M0 = the piece's
// LDS address; no staging registers, no ds_write); 0: through registers, one step ahead.  Both are parity-tested.  Round 2 measured the
// BUILTIN form of the copy 3.7 % slower than the registers (170.4 against 164.3 us on the dominant launch, profiles/r02_f6_knockout.txt):
// hipcc guards the step's first ds_read with vmcnt(0) for it, so the copy landed in front of the MFMAs.  The asm form (conv_halo16.hip's
// WDMA recipe) rides behind them: 137.0 -> 134.4 us at 16 images, 262.0 -> 256.5 at 32, 516.9 -> 505.6 at 64 (alternating libraries,
// profiles/r06_f6_ab.txt), generator forward 2.21-2.22 -> 2.17-2.22 ms per image.
#ifndef HOIG_F6_DMA
#define HOIG_F6_DMA 1
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

constexpr int REC = 56;                       // bytes of an fp6 record: 2 x 24 B of elements, scale bytes at 48 / 49
constexpr float W_SCALE = 256.f;

struct F6Args {
    const float *A, *A2;                      // A2 (nullable): the input is [A | A2] along channels, A holding the first cg1 (% 32)
    int cg1;
    const unsigned short *Wh;                 // fp16 hi plane of w * 2^8, blocked 32(n) x 32(k) (plane_index of conv_igemm_bf16.hip)
    const unsigned char *Qh, *Ql;             // fp6 records of hi / lo: [(tap * ncb64 + cb64)][n][REC]
    const float *bias;
    float *C;
    int Bn, H, W, Cg, N, K;
    int act;
    float slope;
    int nblk_n, nblk, tiles_x, tiles_y;
    // NORMIN (inference chains, as conv_halo16.hip's): x * in_scale + in_shift per (image, gathered channel), ReLU on channel blocks
    // >= in_relu_c0 (a multiple of 32), applied when the halo is converted; the zero padding stays zero
    const float *in_scale, *in_shift;
    int in_relu_c0;
    float *stats;                             // nullable: per image [sum | sum of squares][N] of the stored values, atomically added
};

__device__ __forceinline__ size_t plane_index(int n, int k, int K) {
    return ((size_t)(n >> 5) * (K >> 5) + (k >> 5)) * 1024 + (n & 31) * 32 + (((((k & 31) >> 3) ^ ((n >> 2) & 3))) << 3) +
           (k & 7);
}

// 32 fp32 values of one (row, 32-channel block) -> fp16 hi (64 B), fp6 records of hi and of the residual, E8M0 scale bytes
struct Split32 {
    uint4 hi[4];
    u32x6 qh, ql;
    unsigned sh, sl;
};
__device__ __forceinline__ Split32 split32(const float4 (&v)[8], float pre) {
    Split32 o;
    f16x32 h, l;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x[4] = {v[i].x * pre, v[i].y * pre, v[i].z * pre, v[i].w * pre};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const _Float16 hh = (_Float16)x[k];
            h[i * 4 + k] = hh;
            l[i * 4 + k] = (_Float16)(x[k] - (float)hh);
            amax = fmaxf(amax, fabsf(x[k]));
        }
    }
    // exponent of the block maximum (zero block: any scale will do, take 2^-20)
    const int e = amax > 0.f ? (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 : -20;
    const int eh = max(min(e - 2, 100), -100), el = max(eh - 11, -120);
    const float sch = __uint_as_float((unsigned)(eh + 127) << 23), scl = __uint_as_float((unsigned)(el + 127) << 23);
    o.qh = __builtin_bit_cast(u32x6, __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, sch));
    o.ql = __builtin_bit_cast(u32x6, __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(l, scl));
    o.sh = (unsigned)(eh + 127);
    o.sl = (unsigned)(el + 127);
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    const u32x16 hb = __builtin_bit_cast(u32x16, h);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.hi[i] = make_uint4(hb[4 * i], hb[4 * i + 1], hb[4 * i + 2], hb[4 * i + 3]);
    return o;
}

// weights: one thread per (output channel, tap, 32-channel block)
__global__ void pack_f6_kernel(const float *__restrict__ w, int Co, int RS, int Ci, unsigned char *__restrict__ qh,
                               unsigned char *__restrict__ ql) {
    const int nkb = Ci >> 5;
    const int64_t n = (int64_t)Co * RS * nkb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int kb = (int)(i % nkb);
        const int64_t t = i / nkb;
        const int rs = (int)(t % RS), co = (int)(t / RS);
        const float *src = w + ((size_t)co * RS + rs) * Ci + kb * 32;
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(src + k * 4);
        const Split32 s = split32(v, W_SCALE);
        const size_t rec = ((size_t)(rs * (nkb >> 1) + (kb >> 1)) * Co + co) * REC;
        unsigned *dh = reinterpret_cast<unsigned *>(qh + rec + (kb & 1) * 24), *dl = reinterpret_cast<unsigned *>(ql + rec + (kb & 1) * 24);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            dh[k] = s.qh[k];
            dl[k] = s.ql[k];
        }
        qh[rec + 48 + (kb & 1)] = (unsigned char)s.sh;
        ql[rec + 48 + (kb & 1)] = (unsigned char)s.sl;
    }
}

// Every eligible weight of a network in ONE launch (the per-weight kernel above is the stand-alone form): `rows` holds (offset
// into the flat parameter buffer, Co, RS, Ci, byte offset of the weight's records, first task) per weight; a task = one
// (output channel, tap, 32-channel block) half record.
__global__ void pack_f6_all_kernel(const float *__restrict__ flat, const int64_t *__restrict__ rows, int nrows, int64_t ntasks,
                                   unsigned char *__restrict__ qh, unsigned char *__restrict__ ql) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntasks; i += (int64_t)gridDim.x * blockDim.x) {
        int lo_s = 0, hi_s = nrows - 1;
        while (lo_s < hi_s) {                 // last row whose first task <= i
            const int mid = (lo_s + hi_s + 1) >> 1;
            if (rows[(int64_t)mid * 6 + 5] <= i) lo_s = mid; else hi_s = mid - 1;
        }
        const int64_t *rw = rows + (int64_t)lo_s * 6;
        const int Co = (int)rw[1], RS = (int)rw[2], Ci = (int)rw[3];
        unsigned char *oh = qh + rw[4], *ol = ql + rw[4];
        const int64_t t0 = i - rw[5];
        const int nkb = Ci >> 5;
        const int kb = (int)(t0 % nkb);
        const int64_t t = t0 / nkb;
        const int rs = (int)(t % RS), co = (int)(t / RS);
        const float *src = flat + rw[0] + ((size_t)co * RS + rs) * Ci + kb * 32;
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(src + k * 4);
        const Split32 s = split32(v, W_SCALE);
        const size_t rec = ((size_t)(rs * (nkb >> 1) + (kb >> 1)) * Co + co) * REC;
        unsigned *dh = reinterpret_cast<unsigned *>(oh + rec + (kb & 1) * 24), *dl = reinterpret_cast<unsigned *>(ol + rec + (kb & 1) * 24);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            dh[k] = s.qh[k];
            dl[k] = s.ql[k];
        }
        oh[rec + 48 + (kb & 1)] = (unsigned char)s.sh;
        ol[rec + 48 + (kb & 1)] = (unsigned char)s.sl;
    }
}

constexpr int WM = 4, WN = 2, TH = 8, TW = 32, KS = 3, NT = 512;
constexpr int HH = TH + KS - 1, HW = TW + KS - 1, HPIX = HH * HW;              // 10 x 34 = 340 halo pixels
constexpr int AROW = 144;                                                        // fp16 plane: 64 channels = 128 B + 16 B pad
constexpr int A_HI = HPIX * AROW, A_Q = ((HPIX * REC + 15) / 16) * 16;           // 48960, 19040
constexpr int A_BYTES = A_HI + 2 * A_Q;
constexpr int TM = 2;
template <int BN> struct BTile {                                                 // BN = 128: 16384 + 2 * 7168 = 30720 B per stage
    static constexpr int HI = 2 * BN * 64, Q = BN * REC, STAGE = HI + 2 * Q, SMEM = A_BYTES + 2 * STAGE, TN = BN / (32 * WN);
};

__device__ __forceinline__ i32x8 read_rec(const unsigned char *p) {
    const uint2 a = *reinterpret_cast<const uint2 *>(p), b = *reinterpret_cast<const uint2 *>(p + 8),
                c = *reinterpret_cast<const uint2 *>(p + 16);
    i32x8 r;
    r[0] = (int)a.x; r[1] = (int)a.y; r[2] = (int)b.x; r[3] = (int)b.y; r[4] = (int)c.x; r[5] = (int)c.y; r[6] = 0; r[7] = 0;
    return r;
}

// BN = 64: the same tile with half the output channels (twice the workgroups: launches with too few 8x32 pixel tiles for BN = 128)
constexpr int NORMIN_MAX_CG = 1024;                       // NORMIN keeps the image's scale | shift rows in LDS: 2 x 4 KB
constexpr int NORMIN_BYTES = 2 * NORMIN_MAX_CG * 4;
template <int BN, bool NORMIN = false>
__global__ __launch_bounds__(NT) void conv_halo3_f6_kernel(const F6Args p) {
    constexpr int B_HI = BTile<BN>::HI, B_Q = BTile<BN>::Q, B_STAGE = BTile<BN>::STAGE, TN = BTile<BN>::TN;
    constexpr int QCHUNKS = B_Q / 16;                     // 16-B chunks of one fp6 record array
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Ah = smem, *Aqh = smem + A_HI, *Aql = Aqh + A_Q, *Bbase = smem + A_BYTES;
    const float4 *const Nsc = reinterpret_cast<const float4 *>(smem + BTile<BN>::SMEM);          // NORMIN: [scale: Cg][shift: Cg]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int n_mt = p.nblk / p.nblk_n;
    int mt = tile % n_mt;                                   // pixel tiles of one channel tile first (weights stay in the XCD's L2)
    const int n0 = (tile / n_mt) * BN;
    const int tx_ = mt % p.tiles_x;
    mt /= p.tiles_x;
    const int ty_ = mt % p.tiles_y, b = mt / p.tiles_y;
    const int y0 = ty_ * TH, x0 = tx_ * TW;
    const int ncb = p.Cg >> 6, T = ncb * KS * KS;

    // ---- weight tile of a step: fp16 plane (two 32-k blocked tiles: one 16-B chunk per thread each) + fp6 records (linear copy)
    const int brow = tid >> 2, bchunk = tid & 3;            // 128 rows x 4 chunks (BN = 64: the first 256 threads)
    const bool bload = brow < BN;
    const unsigned short *wrow = p.Wh + plane_index(n0 + (bload ? brow : 0), bchunk * 8, p.K);
    const int boff = brow * 64 + ((bchunk ^ ((brow >> 2) & 3)) << 4);
    const int qchunk = min(tid, QCHUNKS - 1);
    uint4 rbh0 = make_uint4(0, 0, 0, 0), rbh1 = rbh0, rbq0 = rbh0, rbq1 = rbh0;
    // Every load of the loop is UNCONDITIONAL (idle threads re-read a valid address and skip the store): a load under a branch makes
    // the compiler's wait-count pass merge the two paths, and the waits after it -- for OLDER loads -- degrade to vmcnt(0)
    // (conv_wino.hip found the same; here it put a full L2 round trip of the weight loads in front of every halo prefetch).
    auto load_b = [&](int step) {
        const int cb = step / 9, tap = step - cb * 9;
        const size_t koff = (size_t)(tap * p.Cg + cb * 64) * 32;           // k-block index * 1024 elements
        rbh0 = *reinterpret_cast<const uint4 *>(wrow + koff);
        rbh1 = *reinterpret_cast<const uint4 *>(wrow + koff + 1024);
        const size_t rec0 = ((size_t)(tap * ncb + cb) * p.N + n0) * REC;   // BN records (BN = 128: 7168 B = 448 chunks per plane)
        rbq0 = *reinterpret_cast<const uint4 *>(p.Qh + rec0 + (size_t)qchunk * 16);
        rbq1 = *reinterpret_cast<const uint4 *>(p.Ql + rec0 + (size_t)qchunk * 16);
    };
    // The same stage by LDS-DMA: the fp16 planes are stored in HBM in exactly the LDS image (plane_index carries the XOR), so a
    // 32-row x 32-k block is 2 KB contiguous on both sides, and so are the record arrays: lane t moves bytes [16 t, 16 t + 16) of
    // each piece; a wave-instruction fills 1 KB of LDS at M0 = the wave's uniform base.  Issued at the top of step s for stage
    // s + 1, awaited (vmcnt) before the barrier that ends step s.
    auto dma_b = [&](int step, int buf) {
        const int cb = step / 9, tap = step - cb * 9;
        const size_t koff = (size_t)(tap * p.Cg + cb * 64) * 32;
        const size_t rec0 = ((size_t)(tap * ncb + cb) * p.N + n0) * REC;
        // (inline asm, M0 = the piece's LDS address: through the builtin hipcc guards the step's first ds_read with vmcnt(0) and the copy
        //  lands BEFORE the MFMAs instead of behind them -- conv_halo16.hip's WDMA recipe)
        const unsigned to0 = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(Bbase + buf * B_STAGE + wave * 1024));
        auto dma16 = [&](const void *src, unsigned to) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(to) : "memory");
        };
        if (wave * 16 < BN) {                                   // 16 rows of 64 B per wave-instruction
            const unsigned short *src = p.Wh + ((size_t)((n0 >> 5) + (tid >> 7)) * (p.K >> 5)) * 1024 + koff + (tid & 127) * 8;
            dma16(src, to0);
            dma16(src + 1024, to0 + BN * 64);
        }
        if (wave * 1024 < B_Q) {                                // 7168 B = 7 wave-instructions per record array (BN = 64: 3.5)
            if (tid < QCHUNKS) {
                dma16(p.Qh + rec0 + (size_t)tid * 16, to0 + B_HI);
                dma16(p.Ql + rec0 + (size_t)tid * 16, to0 + B_HI + B_Q);
            }
        }
    };
    auto store_b = [&](int buf) {
        unsigned char *B = Bbase + buf * B_STAGE;
        if (bload) {
            *reinterpret_cast<uint4 *>(B + boff) = rbh0;
            *reinterpret_cast<uint4 *>(B + BN * 64 + boff) = rbh1;
        }
        if (tid < QCHUNKS) {
            *reinterpret_cast<uint4 *>(B + B_HI + tid * 16) = rbq0;
            *reinterpret_cast<uint4 *>(B + B_HI + B_Q + tid * 16) = rbq1;
        }
    };

    // ---- halo: task = (halo pixel, 32-channel block); 680 tasks over 512 threads
    float4 hreg0[8], hreg1[8];
    auto halo_load = [&](int cb, int slot, float4 (&hr)[8]) {
        const int task = min(tid + slot * NT, HPIX * 2 - 1);
        const int pix = task >> 1, kb = task & 1;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int gy = min(max(y0 - 1 + hy, 0), p.H - 1), gx = min(max(x0 - 1 + hx, 0), p.W - 1);      // (clamped: the frame is zeroed at the store)
        const int c0 = cb * 64 + kb * 32;
        const bool second = p.A2 != nullptr && c0 >= p.cg1;
        const int ld = p.A2 ? (second ? p.Cg - p.cg1 : p.cg1) : p.Cg;
        const float *src = (second ? p.A2 : p.A) + (((size_t)b * p.H + gy) * p.W + gx) * ld + (second ? c0 - p.cg1 : c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) hr[k] = *reinterpret_cast<const float4 *>(src + k * 4);
    };
    auto halo_store = [&](int cb, int slot, float4 (&hr)[8]) {
        const int task = tid + slot * NT;
        if (task < HPIX * 2) {
            const int pix = task >> 1, kb = task & 1;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            if (!inside) {
#pragma unroll
                for (int k = 0; k < 8; ++k) hr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if constexpr (NORMIN) {
                if (inside) {                                           // (the frame is padding of the NORMALISED tensor: zero)
                    const int c0 = cb * 64 + kb * 32;
                    const bool relu = c0 >= p.in_relu_c0;
                    const float4 *sc = Nsc + (c0 >> 2), *sh = Nsc + ((p.Cg + c0) >> 2);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 a = sc[k], c = sh[k];
                        float4 v = hr[k];
                        v.x = fmaf(v.x, a.x, c.x); v.y = fmaf(v.y, a.y, c.y); v.z = fmaf(v.z, a.z, c.z); v.w = fmaf(v.w, a.w, c.w);
                        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        hr[k] = v;
                    }
                }
            }
            const Split32 s = split32(hr, 1.f);
            uint4 *dh = reinterpret_cast<uint4 *>(Ah + pix * AROW + kb * 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) dh[k] = s.hi[k];
            unsigned char *rh = Aqh + pix * REC + kb * 24, *rl = Aql + pix * REC + kb * 24;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                *reinterpret_cast<uint2 *>(rh + k * 8) = make_uint2(s.qh[2 * k], s.qh[2 * k + 1]);
                *reinterpret_cast<uint2 *>(rl + k * 8) = make_uint2(s.ql[2 * k], s.ql[2 * k + 1]);
            }
            Aqh[pix * REC + 48 + kb] = (unsigned char)s.sh;
            Aql[pix * REC + 48 + kb] = (unsigned char)s.sl;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int bread[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + l31;
        bread[j] = row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
    }

    auto compute = [&](int tap, int bbuf) {
        const int r = tap / 3, s_ = tap - r * 3;
        const unsigned char *B = Bbase + bbuf * B_STAGE;
        int apix[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) apix[i] = (wm * TM + i + r) * HW + l31 + s_;
        // fp6 fragments + scales
        i32x8 aqh[TM], aql[TM], bqh[TN], bql[TN];
        int sah[TM], sal[TM], sbh[TN], sbl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned char *rh = Aqh + apix[i] * REC, *rl = Aql + apix[i] * REC;
            aqh[i] = read_rec(rh + lh * 24);
            aql[i] = read_rec(rl + lh * 24);
            sah[i] = rh[48 + lh];
            sal[i] = rl[48 + lh];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nrow = wn * (TN * 32) + j * 32 + l31;
            const unsigned char *rh = B + B_HI + nrow * REC, *rl = B + B_HI + B_Q + nrow * REC;
            bqh[j] = read_rec(rh + lh * 24);
            bql[j] = read_rec(rl + lh * 24);
            sbh[j] = rh[48 + lh];
            sbl[j] = rl[48 + lh];
        }
        // the two cross terms: lo(a) * hi(w) and hi(a) * lo(w), e2m3 x e2m3 (cbsz = blgp = 2), K = 64
        if (!(HOIG_F6_KO & 2))
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aql[i], bqh[j], acc[i][j], 2, 2, 0, sal[i], 0, sbh[j]);
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aqh[i], bql[j], acc[i][j], 2, 2, 0, sah[i], 0, sbl[j]);
            }
        // hi * hi on fp16, four k-steps of 16 channels
        if (!(HOIG_F6_KO & 4))
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 ah[TM], bh[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const f16x8 *>(Ah + apix[i] * AROW + ks * 32 + lh * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bh[j] = *reinterpret_cast<const f16x8 *>(B + (ks >> 1) * (BN * 64) + (bread[j] ^ ((ks & 1) << 5)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };

    halo_load(0, 0, hreg0);
    halo_load(0, 1, hreg1);
    if (HOIG_F6_DMA) dma_b(0, 0);
    else load_b(0);
    if constexpr (NORMIN) {                   // this image's scale | shift rows -> LDS, once
        float4 *dst = reinterpret_cast<float4 *>(smem + BTile<BN>::SMEM);
        const int q4 = p.Cg >> 2;
        for (int i = tid; i < 2 * q4; i += NT)
            dst[i] = *reinterpret_cast<const float4 *>((i < q4 ? p.in_scale : p.in_shift) + (size_t)b * p.Cg + (i < q4 ? i : i - q4) * 4);
        __syncthreads();
    }
    halo_store(0, 0, hreg0);
    halo_store(0, 1, hreg1);
    if (HOIG_F6_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
        store_b(0);
        if (T > 1) load_b(1);
    }
    __syncthreads();
    int bbuf = 0;
#pragma unroll 1
    for (int step = 0; step < T; ++step) {
        const int cb = step / 9, tap = step - cb * 9;
        const bool more = step + 1 < T;
        const bool boundary = more && tap == 8;
        if (HOIG_F6_DMA) {
            if (more && !(HOIG_F6_KO & 16)) dma_b(step + 1, bbuf ^ 1);          // stage s + 1: free since the previous barrier
        } else if (!(HOIG_F6_KO & 16)) {
            if (more) store_b(bbuf ^ 1);                  // weights of step+1 (registers loaded during the previous step)
            load_b(min(step + 2, T - 1));                 // (the last two steps re-request the last stage: no branch around a load)
        }
        if (!(HOIG_F6_KO & 8)) {                          // (AFTER the weight prefetch: the next store_b's wait finds them a step old)
            if (tap == 7 && cb + 1 < ncb) halo_load(cb + 1, 0, hreg0);              // next block's halo: first half of the tasks ...
            if (boundary) halo_load(cb + 1, 1, hreg1);                               // ... second half
        }
        compute(tap, bbuf);
        if (boundary && !(HOIG_F6_KO & 1)) {
            __syncthreads();                              // every wave is done with the halo
            halo_store(cb + 1, 0, hreg0);
            halo_store(cb + 1, 1, hreg1);
        }
        if (HOIG_F6_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage s + 1 has landed (and this wave's halo loads)
        if (!(HOIG_F6_KO & 32)) __syncthreads();
        bbuf ^= 1;
    }

    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    float bias_r[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + l31;
        bias_r[j] = p.bias ? p.bias[n] : 0.f;
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
                const float v = fast_act(acc[i][j][r] * (1.f / W_SCALE) + bias_r[j], nslope, special, p.act, p.slope);
                p.C[pix * p.N + n] = v;
                st1[j] += v;
                st2[j] += v * v;
            }
        }
    }
    if (p.stats) {                            // per-image channel sums for the instance norm that follows (hoig_inorm_stats_from_sums):
        // lane = channel; its 2 x 16 pixels above, + the other half-wave's, + the four pixel-row waves' through LDS (free: the last
        // step closed with a barrier), one atomic per (workgroup, channel, moment)
        float *red = reinterpret_cast<float *>(smem);             // [WM][2][BN]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            st1[j] += __shfl_xor(st1[j], 32);
            st2[j] += __shfl_xor(st2[j], 32);
            if (lh == 0) {
                const int cl = wn * (TN * 32) + j * 32 + l31;
                red[(wm * 2 + 0) * BN + cl] = st1[j];
                red[(wm * 2 + 1) * BN + cl] = st2[j];
            }
        }
        __syncthreads();
        float *stats_img = p.stats + (size_t)b * 2 * p.N;
        for (int e = tid; e < 2 * BN; e += NT) {
            const int mom = e / BN, cl = e - mom * BN;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < WM; ++k) v += red[(k * 2 + mom) * BN + cl];
            atomicAdd(&stats_img[(size_t)mom * p.N + n0 + cl], v);
        }
    }
}

int g_f6_min_tiles = 192;

}  // namespace

extern "C" int64_t hoig_f6_plane_bytes(int Co, int RS, int Ci) {
    if (Co <= 0 || RS <= 0 || Ci <= 0 || (Ci & 63)) return 0;
    return (int64_t)RS * (Ci >> 6) * Co * REC;
}

extern "C" int hoig_pack_conv_weight_f6(const float *w, int Co, int RS, int Ci, uint8_t *q_hi, uint8_t *q_lo, hoig_stream_t stream) {
    if (!w || !q_hi || !q_lo || Co <= 0 || RS <= 0) return HOIG_EINVAL;
    if (Ci & 63) return HOIG_EUNSUPPORTED;
    const int64_t n = (int64_t)Co * RS * (Ci >> 5);
    pack_f6_kernel<<<hoig_stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(w, Co, RS, Ci, q_hi, q_lo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

// forward 3x3 stride-1 pad-1 convolution; HOIG_EUNSUPPORTED for every shape outside this kernel's tiling (the caller then uses
// hoig_conv2d_fwd_packed with HOIG_PREC_BF16X3)
static int launch_f6(const hoig_conv_desc *d, const float *x, const float *x2, int cg1, const uint16_t *w_hi, const uint8_t *q_hi,
                     const uint8_t *q_lo, const float *bias, float *y, hipStream_t st, const float *in_scale = nullptr,
                     const float *in_shift = nullptr, int in_relu_c0 = 0, float *stats = nullptr) {
    if (d->transposed || d->stride != 1 || d->R != 3 || d->S != 3 || d->pad != 1 || d->Hi != d->Ho || d->Wi != d->Wo)
        return HOIG_EUNSUPPORTED;
    if ((d->Ci & 63) || (d->Co & 63) || (d->Hi & 7) || (d->Wi & 31)) return HOIG_EUNSUPPORTED;
    if (x2 && (cg1 <= 0 || cg1 >= d->Ci || (cg1 & 31))) return HOIG_EINVAL;
    F6Args a;
    a.A = x; a.A2 = x2; a.cg1 = cg1; a.Wh = w_hi; a.Qh = q_hi; a.Ql = q_lo; a.bias = bias; a.C = y;
    a.Bn = d->B; a.H = d->Hi; a.W = d->Wi; a.Cg = d->Ci; a.N = d->Co; a.K = 9 * d->Ci;
    a.act = d->act; a.slope = d->slope;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu_c0 = in_relu_c0; a.stats = stats;
    if (in_scale && (a.Cg > NORMIN_MAX_CG || (in_relu_c0 & 31))) return HOIG_EUNSUPPORTED;
    a.tiles_x = a.W / TW; a.tiles_y = a.H / TH;
    const int ptiles = a.Bn * a.tiles_x * a.tiles_y;
    // 128-channel tiles when they fill the chip, 64-channel tiles (twice the workgroups) otherwise; launches that stay below
    // g_f6_min_tiles workgroups even so are left to the 4-row variants of the three-term path
    const bool n64 = (a.N & 127) != 0 || ptiles * (a.N / 128) < 192;
    a.nblk_n = a.N / (n64 ? 64 : 128);
    a.nblk = ptiles * a.nblk_n;
    if (a.nblk < g_f6_min_tiles) return HOIG_EUNSUPPORTED;
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_f6_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BTile<128>::SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_f6_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                BTile<64>::SMEM) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    if (in_scale) {
        static hoig_once once_n;
        if (!once_n.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_f6_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    BTile<128>::SMEM + NORMIN_BYTES) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_f6_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    BTile<64>::SMEM + NORMIN_BYTES) != hipSuccess)
                return HOIG_ELAUNCH;
            once_n.set();
        }
        if (n64) conv_halo3_f6_kernel<64, true><<<a.nblk, NT, BTile<64>::SMEM + NORMIN_BYTES, st>>>(a);
        else conv_halo3_f6_kernel<128, true><<<a.nblk, NT, BTile<128>::SMEM + NORMIN_BYTES, st>>>(a);
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    if (n64) conv_halo3_f6_kernel<64><<<a.nblk, NT, BTile<64>::SMEM, st>>>(a);
    else conv_halo3_f6_kernel<128><<<a.nblk, NT, BTile<128>::SMEM, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

// forward 3x3 stride-1 pad-1 convolution; HOIG_EUNSUPPORTED for every shape outside this kernel's tiling (the caller then uses
// hoig_conv2d_fwd_packed with HOIG_PREC_BF16X3)
extern "C" int hoig_conv2d_fwd_f6(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint8_t *q_hi,
                                  const uint8_t *q_lo, const float *bias, float *y, hoig_stream_t stream) {
    if (!d || !x || !w_hi || !q_hi || !q_lo || !y) return HOIG_EINVAL;
    return launch_f6(d, x, nullptr, 0, w_hi, q_hi, q_lo, bias, y, (hipStream_t)stream);
}
// the same over the channel concatenation [x1 | x2] without materialising it (the decoder's skip convolutions)
extern "C" int hoig_conv2d_cat_fwd_f6(const hoig_conv_desc *d, const float *x1, int C1, const float *x2, const uint16_t *w_hi,
                                      const uint8_t *q_hi, const uint8_t *q_lo, const float *bias, float *y,
                                      hoig_stream_t stream) {
    if (!d || !x1 || !x2 || !w_hi || !q_hi || !q_lo || !y) return HOIG_EINVAL;
    return launch_f6(d, x1, x2, C1, w_hi, q_hi, q_lo, bias, y, (hipStream_t)stream);
}

// hoig_conv2d_fwd_f6 / hoig_conv2d_cat_fwd_f6 with the loader and epilogue options of the three-term path (include/hoig_kernels.h):
// x2 (nullable) = the second tensor of a channel concatenation; in_scale / in_shift (nullable, together) = the instance norm + ReLU of
// the gathered tensor applied in the loader; stats (nullable) = the per-image channel sums of y for the norm that follows
extern "C" int hoig_conv2d_fwd_f6_ex(const hoig_conv_desc *d, const float *x, int C1, const float *x2, const uint16_t *w_hi,
                                     const uint8_t *q_hi, const uint8_t *q_lo, const float *bias, const float *in_scale,
                                     const float *in_shift, int in_relu_c0, float *y, float *stats, hoig_stream_t stream) {
    if (!d || !x || !w_hi || !q_hi || !q_lo || !y || (in_scale == nullptr) != (in_shift == nullptr) || in_relu_c0 < 0) return HOIG_EINVAL;
    return launch_f6(d, x, x2, x2 ? C1 : 0, w_hi, q_hi, q_lo, bias, y, (hipStream_t)stream, in_scale, in_shift, in_relu_c0, stats);
}

extern "C" int hoig_pack_conv_weights_f6_all(const float *flat, const int64_t *rows, int nrows, int64_t ntasks, uint8_t *q_hi,
                                             uint8_t *q_lo, hoig_stream_t stream) {
    if (!flat || !rows || nrows <= 0 || ntasks <= 0 || !q_hi || !q_lo) return HOIG_EINVAL;
    pack_f6_all_kernel<<<hoig_stream_grid(ntasks, 256), 256, 0, (hipStream_t)stream>>>(flat, rows, nrows, ntasks, q_hi, q_lo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_set_f6_min_tiles(int n) {
    const int old = g_f6_min_tiles;
    if (n > 0) g_f6_min_tiles = n;
    return old;
}
