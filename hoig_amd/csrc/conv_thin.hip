// Stride-1 "same" convolutions with a THIN side -- at most 8 channels on the input (the 7x7 stems over 3 / 8-channel images,
// SPADE's 3x3 over the 3-channel condition map, VGG's first layer) or on the output (the 7x7 image / mask heads, 64 -> 1 | 3)
// -- on the MFMA.  The 32-wide tiles of the implicit-GEMM kernels waste > 10x of their work on such layers, so rounds 1-2 ran
// them on fp32 VALU kernels at 25-90 TFLOP/s: 7 ms of an 80 ms training step for 3 % of its FLOPs (profiles/r02_conv_table.txt).
// The remedy is to put the TAPS where the missing channels are:
//
//   thin_gemm_kernel   out[p][j] = sum_{tap,f} T[p + off(tap)][f] * Wm[(tap,f)][j]        M = pixels, K = taps x thin, N = 64 dense
//       forward of a thin-INPUT conv (T = x, j = output channel) and data gradient of a thin-OUTPUT conv (T = dy, taps mirrored,
//       j = input channel).  The A operand is an im2col of a planar 16-bit halo image of T in LDS (eight 2-byte gathers per
//       fragment through a k -> offset table), the B operand the 64 x K weight matrix, resident in LDS for a strip of tiles.
//   thin_wgrad_kernel  G[j][(tap,f)] = sum_p D[p][j] * T[p + off(tap)][f]                  M = 64 dense, N = taps x thin, K = pixels
//       weight gradient of both kinds: thin input (D = dy, T = x: dW[j][tap][f]) and thin output (D = x, T = dy, taps mirrored:
//       dW[f][tap][j]).  D tiles are stored as they arrive ([pixel][channel]) and read transposed (ds_read_b64_tr_b16), the
//       B operand is eight consecutive pixels of T's halo image at the lane's tap offset.  Persistent workgroups accumulate in
//       registers over many tiles and add their 64 x N partial once (LDS-staged so that the atomics are contiguous runs).
//
// Arithmetic follows the precision modes of the other kernels: forward on fp16 halves (weights scaled by 2^8), gradients on
// bf16 halves; HOIG_PREC_BF16X3 = three terms (both operands split hi + lo), HOIG_PREC_F16X2 = the gathered / gradient operand
// split, the other one single, HOIG_PREC_BF16 = one term; fp32 accumulate.
#include "common.h"
#include <map>
#include <mutex>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 32, NTHR = 256;

struct ThinArgs {
    const float *T;          // thin tensor [B][H][W][F]
    const float *D;          // wgrad: dense tensor [B][H][W][CD];  gemm: the packed fp32 conv weight
    const float *bias;       // gemm forward: per dense channel (nullable)
    float *Out;              // gemm: [B][H][W][CD];  wgrad: dW, accumulated with atomics
    float *Ws;               // wgrad: per-workgroup partials [gridDim.y][gridDim.x][64][NP] (nullable: atomics straight into dW)
    int Bn, H, W, F, CD, KS, pad;
    int flip;                // taps mirrored: the thin tensor is a gradient (thin-OUTPUT convolutions)
    int sj, st, sf;          // weight / dW element index = j * sj + tap * st + f * sf   (j dense channel, f thin channel)
    int act;
    float slope;
    int accumulate;          // gemm: Out += result
    float wscale;            // gemm: weights are multiplied by this before the 16-bit split, the result by its inverse
    int K, KP, N, NP;        // K = KS*KS*F (gemm reduction, padded to 16) ; N = the same count as wgrad's columns (padded to 32)
    int tiles_x, tiles_y, ntiles, strip;
    int HR, CW;              // halo image: rows, row pitch (elements)
    float *stats;            // gemm forward (nullable): per image [sum | sum of squares][CD] of the stored values, added atomically
};

template <bool FP16>
__device__ __forceinline__ void split16(float x, unsigned short &hi, unsigned short &lo) {
    if (FP16) {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    } else {
        const __bf16 h = (__bf16)x;                        // v_cvt_pk_bf16_f32 (round to nearest even)
        hi = __builtin_bit_cast(unsigned short, h);
        const __bf16 l = (__bf16)(x - hoig_bf2f(hi));
        lo = __builtin_bit_cast(unsigned short, l);
    }
}

template <bool FP16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if (FP16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// The halo image of one tile of the thin tensor: planes [NT][F][HR][CW] of 16-bit values, zero outside the picture.  Loaded into
// registers one tile ahead (HMAX values per thread cover the largest image: 10 x 38 pixels x 8 channels) and written to LDS
// when the tile's turn comes, so that the load latency hides behind the previous tile's MFMAs.
// Which element of the halo image a thread moves in round `it` does not depend on the tile: the (row, column, channel) of every
// round is worked out ONCE per kernel (HaloMap: the integer divisions cost more than the tile's MFMAs when redone per tile).
constexpr int HMAX = 12;
struct HaloRegs {
    float v[HMAX];
};
struct HaloMap {
    int lds[HMAX];                 // element offset in the LDS plane, -1: no element in this round
    int rc[HMAX];                  // halo row | column << 8 | channel << 16
    int rounds;
};
__device__ __forceinline__ void halo_map(const ThinArgs &p, HaloMap &m) {
    const int wr = TW + p.KS - 1;                          // halo pixels per row
    const int row_elems = wr * p.F;                        // contiguous in memory: (x0 - pad .. x0 - pad + wr) x F
    const int total = p.HR * row_elems;
    m.rounds = (total + NTHR - 1) / NTHR;
#pragma unroll
    for (int it = 0; it < HMAX; ++it) {
        const int i = threadIdx.x + it * NTHR;
        m.lds[it] = -1;
        m.rc[it] = 0;
        if (i < total) {
            const int hy = i / row_elems, e = i - hy * row_elems;
            const int hx = e / p.F, f = e - hx * p.F;
            m.lds[it] = (f * p.HR + hy) * p.CW + hx;
            m.rc[it] = hy | (hx << 8) | (f << 16);
        }
    }
}
__device__ __forceinline__ void halo_load(const ThinArgs &p, const HaloMap &m, int b, int y0, int x0, HaloRegs &h) {
    const float *img = p.T + (size_t)b * p.H * p.W * p.F;
#pragma unroll
    for (int it = 0; it < HMAX; ++it) {
        h.v[it] = 0.f;
        if (it < m.rounds && m.lds[it] >= 0) {             // (first test uniform)
            const int gy = y0 - p.pad + (m.rc[it] & 255), gx = x0 - p.pad + ((m.rc[it] >> 8) & 255);
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) h.v[it] = img[((size_t)gy * p.W + gx) * p.F + (m.rc[it] >> 16)];
        }
    }
}
template <bool FP16, int NT>
__device__ __forceinline__ void halo_store(const ThinArgs &p, const HaloMap &m, const HaloRegs &h, unsigned short *Tl) {
    const int plane = p.F * p.HR * p.CW;
#pragma unroll
    for (int it = 0; it < HMAX; ++it)
        if (it < m.rounds && m.lds[it] >= 0) {
            unsigned short hi, lo;
            split16<FP16>(h.v[it], hi, lo);
            Tl[m.lds[it]] = hi;
            if (NT == 2) Tl[plane + m.lds[it]] = lo;
        }
}

__device__ __forceinline__ void tile_coords(const ThinArgs &p, int t, int &b, int &y0, int &x0) {
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    b = t / p.tiles_y;
    y0 = ty * TH;
    x0 = tx * TW;
}

// ---------------------------------------------------------------------------------------------------------------- gemm
template <bool FP16, int NT, int NW>
__global__ __launch_bounds__(NTHR) void thin_gemm_kernel(const ThinArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wstride = p.KP + 8;                                        // elements per weight row (16-B pad: conflict-free b128)
    unsigned short *Wl = reinterpret_cast<unsigned short *>(smem);       // [NW][64][wstride]
    unsigned short *toff = Wl + NW * 64 * wstride;                       // [KP] halo offset of reduction index k
    unsigned short *Tl = toff + p.KP;                                    // [NT][F][HR][CW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int cbase = blockIdx.y * 64;
    const int KK = p.KS * p.KS;
    const int tplane = p.F * p.HR * p.CW;

    for (int i0 = tid; i0 < 64 * p.KP; i0 += 4 * NTHR) {                  // (four independent loads in flight per thread)
        float v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NTHR;
            v[u] = 0.f;
            dst[u] = -1;
            if (i < 64 * p.KP) {
                const int n = i / p.KP, k = i - n * p.KP;
                dst[u] = n * wstride + k;
                if (k < p.K) {
                    const int tap = k / p.F, f = k - tap * p.F;
                    const int tw = p.flip ? KK - 1 - tap : tap;
                    v[u] = p.D[(size_t)(cbase + n) * p.sj + (size_t)tw * p.st + (size_t)f * p.sf];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) {
                unsigned short hi, lo;
                split16<FP16>(v[u] * p.wscale, hi, lo);
                Wl[dst[u]] = hi;
                if (NW == 2) Wl[64 * wstride + dst[u]] = lo;
            }
    }
    for (int k = tid; k < p.KP; k += NTHR) {
        const int kc = k < p.K ? k : p.K - 1;                            // (padding: any valid address, the weight is zero)
        const int tap = kc / p.F, f = kc - tap * p.F;
        const int r = tap / p.KS, s = tap - r * p.KS;
        toff[k] = (unsigned short)((f * p.HR + r) * p.CW + s);
    }

    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    const float inv = 1.f / p.wscale;
    float bias_r[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_r[j] = p.bias ? p.bias[cbase + j * 32 + l31] : 0.f;
    const int ksteps = p.KP >> 4;
    const int pix = wave * p.CW + l31;                                   // this lane's pixel inside the halo image (tap 0,0)

    const int t_end = min(p.ntiles, (int)(blockIdx.x + 1) * p.strip);
    HaloRegs hreg;
    HaloMap hmap;
    halo_map(p, hmap);
    {
        int b, y0, x0;
        tile_coords(p, blockIdx.x * p.strip, b, y0, x0);
        halo_load(p, hmap, b, y0, x0, hreg);
    }
    // channel sums for the instance norm that follows (the 7x7 stems: hoig_conv2d_fwd_stats): lane = channel, accumulated over the tiles of
    // ONE image (a workgroup's strip rarely crosses an image) and added atomically when the image changes and at the end
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};
    int sb = -1;
    auto flush_stats = [&]() {
        if (sb >= 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s1 = st1[j] + __shfl_xor(st1[j], 32), s2 = st2[j] + __shfl_xor(st2[j], 32);
                if (lh == 0) {
                    float *dst = p.stats + (size_t)sb * 2 * p.CD + cbase + j * 32 + l31;
                    atomicAdd(dst, s1);
                    atomicAdd(dst + p.CD, s2);
                }
                st1[j] = st2[j] = 0.f;
            }
        }
    };
    for (int t = blockIdx.x * p.strip; t < t_end; ++t) {
        int b, y0, x0;
        tile_coords(p, t, b, y0, x0);
        if (p.stats && b != sb) {
            flush_stats();
            sb = b;
        }
        __syncthreads();                                                 // the previous tile's reads (and the tables) are done
        halo_store<FP16, NT>(p, hmap, hreg, Tl);
        __syncthreads();
        if (t + 1 < t_end) {                                             // the next tile's halo flies during this tile's MFMAs
            int b2, y2, x2;
            tile_coords(p, t + 1, b2, y2, x2);
            halo_load(p, hmap, b2, y2, x2, hreg);
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
        for (int kk = 0; kk < ksteps; ++kk) {
            const int k0 = kk * 16 + lh * 8;
            const bf16x8 offs = *reinterpret_cast<const bf16x8 *>(toff + k0);
            bf16x8 a0, a1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int o = (int)(unsigned short)offs[e] + pix;
                a0[e] = (short)Tl[o];
                if (NT == 2) a1[e] = (short)Tl[tplane + o];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned short *wr = Wl + (j * 32 + l31) * wstride + k0;
                const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(wr);
                if (NT == 2) acc[j] = mfma16<FP16>(a1, w0, acc[j]);
                if (NW == 2) acc[j] = mfma16<FP16>(a0, *reinterpret_cast<const bf16x8 *>(wr + 64 * wstride), acc[j]);
                acc[j] = mfma16<FP16>(a0, w0, acc[j]);
            }
        }
        const int oy = y0 + wave;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float *dst = p.Out + (((size_t)b * p.H + oy) * p.W + ox) * p.CD + cbase + l31;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = acc[j][r] * inv + bias_r[j];
                v = fast_act(v, nslope, special, p.act, p.slope);
                if (p.accumulate) v += dst[j * 32];
                dst[j * 32] = v;
                st1[j] += v;
                st2[j] += v * v;
            }
        }
    }
    if (p.stats) flush_stats();
}

// --------------------------------------------------------------------------------------------------------------- wgrad
constexpr int DROW = 192;            // bytes per pixel row of a D tile: 64 channels x 2 B + 64 B (transposed reads conflict-free)

template <int NPW, int WN, int NT, int ND>
__global__ __launch_bounds__(NTHR) void thin_wgrad_kernel(const ThinArgs p) {
    constexpr int WK = 4 / WN, NPIX = TH * TW, KSTEPS = NPIX / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Dl = smem;                                             // [ND][NPIX][DROW]
    unsigned short *Tl = reinterpret_cast<unsigned short *>(smem + ND * NPIX * DROW);      // [NT][F][HR][CW]
    float *Gl = reinterpret_cast<float *>(smem);                          // epilogue: [64][NP] over the tiles' space
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wn = wave % WN, wk = wave / WN;
    const int cbase = blockIdx.y * 64;
    const int KK = p.KS * p.KS;
    const int tplane = p.F * p.HR * p.CW;

    // this lane's columns n = (tap, f): halo offset of its tap (mirrored for gradients), or invalid beyond N
    int noff[NPW];
    bool nval[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int n = (wn * NPW + j) * 32 + l31;
        nval[j] = n < p.N;
        const int nc = nval[j] ? n : 0;
        const int tap = nc / p.F, f = nc - tap * p.F;
        const int tt = p.flip ? KK - 1 - tap : tap;
        const int r = tt / p.KS, s = tt - r * p.KS;
        noff[j] = (f * p.HR + r) * p.CW + s + lh * 8;
    }
    // transposed-read addresses of the A operand (tools/trtest.hip): lane 16g + 4q + pp supplies row (pixel) 8 (g >> 1) + q,
    // channels 16 (g & 1) + 4 pp .. + 3; a second read four pixels further completes the eight k of the fragment
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int a_off = (8 * (g >> 1) + q) * DROW + (16 * (g & 1) + 4 * pp) * 2;

    f32x16 acc[2][NPW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // D tile loader: thread -> (pixel, 8 channels): NPIX * 8 tasks over 256 threads, 32 B of fp32 each
    constexpr int DT = NPIX * 8 / NTHR;                                   // 4
    float4 dreg[DT][2];
    auto load_d = [&](int t) {
        int b, y0, x0;
        tile_coords(p, t, b, y0, x0);
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const int task = tid + i * NTHR, px = task >> 3, c8 = task & 7;
            const int yy = y0 + (px >> 5), xx = x0 + (px & 31);
            const float *src = p.D + (((size_t)b * p.H + yy) * p.W + xx) * p.CD + cbase + c8 * 8;
            dreg[i][0] = *reinterpret_cast<const float4 *>(src);
            dreg[i][1] = *reinterpret_cast<const float4 *>(src + 4);
        }
    };
    auto store_d = [&]() {
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const int task = tid + i * NTHR, px = task >> 3, c8 = task & 7;
            const float v[8] = {dreg[i][0].x, dreg[i][0].y, dreg[i][0].z, dreg[i][0].w,
                                dreg[i][1].x, dreg[i][1].y, dreg[i][1].z, dreg[i][1].w};
            bf16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {               // two values per v_cvt_pk_bf16_f32
                typedef float f2v __attribute__((ext_vector_type(2)));
                typedef __bf16 b2v __attribute__((ext_vector_type(2)));
                const f2v x2 = {v[e], v[e + 1]};
                const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(x2, b2v));
                hi[e] = (short)(hp & 0xffffu);
                hi[e + 1] = (short)(hp >> 16);
                if (ND == 2) {
                    const f2v r2 = {v[e] - __uint_as_float(hp << 16), v[e + 1] - __uint_as_float(hp & 0xffff0000u)};
                    const unsigned lp = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, b2v));
                    lo[e] = (short)(lp & 0xffffu);
                    lo[e + 1] = (short)(lp >> 16);
                }
            }
            *reinterpret_cast<bf16x8 *>(Dl + px * DROW + c8 * 16) = hi;
            if (ND == 2) *reinterpret_cast<bf16x8 *>(Dl + NPIX * DROW + px * DROW + c8 * 16) = lo;
        }
    };

    int t = blockIdx.x;
    HaloRegs hreg;
    HaloMap hmap;
    halo_map(p, hmap);
    if (t < p.ntiles) {
        int b, y0, x0;
        tile_coords(p, t, b, y0, x0);
        load_d(t);
        halo_load(p, hmap, b, y0, x0, hreg);
    }
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                                                  // the previous tile has been consumed
        store_d();
        halo_store<false, NT>(p, hmap, hreg, Tl);
        __syncthreads();
        if (t + (int)gridDim.x < p.ntiles) {                              // the next tile's loads fly during the MFMAs
            int b, y0, x0;
            tile_coords(p, t + gridDim.x, b, y0, x0);
            load_d(t + gridDim.x);
            halo_load(p, hmap, b, y0, x0, hreg);
        }
#pragma unroll 1
        for (int ks = wk; ks < KSTEPS; ks += WK) {
            const int prow = ks >> 1, pcol = (ks & 1) * 16;               // sixteen pixels of row `prow`, from column `pcol`
            bf16x8 a[2][ND];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const unsigned char *src = Dl + d * NPIX * DROW + (prow * 32 + pcol) * DROW + a_off + i * 64;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(src));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(src + 4 * DROW));
                    a[i][d] = (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
            const int base = prow * p.CW + pcol;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                bf16x8 b0, b1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = base + noff[j] + e;
                    b0[e] = nval[j] ? (short)Tl[o] : (short)0;
                    if (NT == 2) b1[e] = nval[j] ? (short)Tl[tplane + o] : (short)0;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (ND == 2) acc[i][j] = mfma16<false>(a[i][1], b0, acc[i][j]);
                    if (NT == 2) acc[i][j] = mfma16<false>(a[i][0], b1, acc[i][j]);
                    acc[i][j] = mfma16<false>(a[i][0], b0, acc[i][j]);
                }
            }
        }
    }

    // ---- the workgroup's 64 x N partial: summed over the k-waves in LDS, then added to dW in contiguous runs
    __syncthreads();
    for (int i = tid; i < 64 * p.NP; i += NTHR) Gl[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int n = (wn * NPW + j) * 32 + l31;
            if (n < p.NP)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (WK > 1) atomicAdd(&Gl[m * p.NP + n], acc[i][j][r]);
                    else Gl[m * p.NP + n] = acc[i][j][r];
                }
        }
    __syncthreads();
    if (p.Ws) {          // hundreds of workgroups adding into the same few KB of dW serialise at the memory-side atomic units
                         // (measured: 78 us for the 64 -> 1 heads, 73 of them atomics): plain stores here, thin_reduce_kernel adds
        float *dst = p.Ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 * p.NP;
        for (int i = tid; i < 64 * p.NP; i += NTHR) dst[i] = Gl[i];
        return;
    }
    if (p.sj == 1) {                                                      // dW[f][tap][j]: consecutive threads along j
        for (int i = tid; i < 64 * p.N; i += NTHR) {
            const int n = i >> 6, m = i & 63;
            const int tap = n / p.F, f = n - tap * p.F;
            atomicAdd(p.Out + (size_t)f * p.sf + (size_t)tap * p.st + cbase + m, Gl[m * p.NP + n]);
        }
    } else {                                                              // dW[j][tap][f]: consecutive threads along (tap, f)
        for (int i = tid; i < 64 * p.N; i += NTHR) {
            const int m = i / p.N, n = i - m * p.N;
            const int tap = n / p.F, f = n - tap * p.F;
            atomicAdd(p.Out + (size_t)(cbase + m) * p.sj + (size_t)tap * p.st + (size_t)f * p.sf, Gl[m * p.NP + n]);
        }
    }
}

// dW += sum over the workgroups' partials (one thread per dW element of this 64-channel group, coalesced over the partials)
__global__ __launch_bounds__(NTHR) void thin_reduce_kernel(const ThinArgs p, int nwg) {
    const int cbase = blockIdx.y * 64;
    const float *src = p.Ws + (size_t)blockIdx.y * nwg * 64 * p.NP;
    const int i = blockIdx.x * NTHR + threadIdx.x;
    if (i >= 64 * p.N) return;
    const int m = i / p.N, n = i - m * p.N;                // (threads along n: the partials are read in contiguous runs)
    // blockIdx.z: a chunk of the partials (a thread sums at most 32 of them, four loads in flight; <= 16 atomics per element)
    const int wper = (nwg + (int)gridDim.z - 1) / (int)gridDim.z;
    const int w0 = blockIdx.z * wper, w1 = min(nwg, w0 + wper);
    const size_t step = (size_t)64 * p.NP;
    const float *col = src + m * p.NP + n + (size_t)w0 * step;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int w = w0;
    for (; w + 3 < w1; w += 4, col += 4 * step) {
        v0 += col[0];
        v1 += col[step];
        v2 += col[2 * step];
        v3 += col[3 * step];
    }
    for (; w < w1; ++w, col += step) v0 += col[0];
    const float v = (v0 + v1) + (v2 + v3);
    const int tap = n / p.F, f = n - tap * p.F;
    atomicAdd(p.Out + (size_t)(cbase + m) * p.sj + (size_t)tap * p.st + (size_t)f * p.sf, v);
}

// Workspace for the partials: the launching stream's scratch block, which the CALLER owns and registers
// (hoig_stream_scratch_set, pointwise.hip); without one the kernel falls back to atomics.
}  // namespace
namespace hoig_detail { void *stream_scratch(hipStream_t st, size_t bytes); }
namespace {
constexpr size_t WS_SLOT_BYTES = (size_t)16 << 20;      // = hoig_stream_scratch_bytes(): what a caller registers per stream
float *thin_workspace(hipStream_t st, size_t bytes) { return static_cast<float *>(hoig_detail::stream_scratch(st, bytes)); }

// ------------------------------------------------------------------------------------------------------------ thin OUTPUT
// out[p][f] = sum_tap sum_j D[p + off(tap)][j] * W[f][tap][j]  with F <= 16 output channels: the FORWARD of the image / mask heads
// (64 -> 1 | 3 | 5 fused, 7x7) and the data gradient of a thin-input layer (VGG conv1_1: 64 -> 3, 3x3, taps mirrored).  The
// 16x16x32 MFMA keeps N at 16 (the 32-wide form would waste twice as much): a wave owns two 16-pixel row segments, A fragments
// are read from the 16-bit halo image of D (32 channels per pass, 80-B pixel rows) at the tap's offset, B fragments from the
// LDS-resident weights [f][tap][64]; lanes n >= F multiply zeros.  Per-channel activations (tanh image | sigmoid mask | none)
// come as 4-bit codes.
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct ThinOutArgs {
    const float *D, *Wt, *bias;
    float *Out;
    int Bn, H, W, F, CD, KS, pad, flip;
    int sf, st, sj;          // weight element index = f * sf + tap * st + j * sj
    unsigned long long acts; // activation code of output channel f in bits [4f, 4f+4)
    float slope, wscale;
    int tiles_x, tiles_y, ntiles, strip, HR, HWc;
};

template <bool FP16>
__device__ __forceinline__ f32x4v mfma16x16(const bf16x8 a, const bf16x8 b, const f32x4v c) {
    if (FP16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <bool FP16, int ND, int NW>
__global__ __launch_bounds__(NTHR) void thin_out_kernel(const ThinOutArgs p) {
    constexpr int PROW = 80, WROW = 144;                                  // bytes per halo pixel (32 ch + pad) / weight row (64 ch + pad)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KK = p.KS * p.KS, hpix = p.HR * p.HWc;
    unsigned char *Wl = smem;                                             // [NW][CD / 64][F][KK][WROW]
    unsigned char *Dl = smem + (size_t)NW * (p.CD >> 6) * p.F * KK * WROW;        // [ND][hpix][PROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int wplane = p.F * KK * WROW;

    for (int i0 = tid; i0 < p.F * KK * p.CD; i0 += 4 * NTHR) {            // weights -> 16-bit planes (four loads in flight)
        float v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NTHR;
            v[u] = 0.f;
            dst[u] = -1;
            if (i < p.F * KK * p.CD) {
                const int j = i % p.CD, ft = i / p.CD;
                const int tap = ft % KK, f = ft / KK;
                const int tw = p.flip ? KK - 1 - tap : tap;
                v[u] = p.Wt[(size_t)f * p.sf + (size_t)tw * p.st + (size_t)j * p.sj];
                dst[u] = ((j >> 6) * p.F * KK + f * KK + tap) * WROW + (j & 63) * 2;      // (64-channel groups one after the other)
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) {
                unsigned short hi, lo;
                split16<FP16>(v[u] * p.wscale, hi, lo);
                *reinterpret_cast<unsigned short *>(Wl + dst[u]) = hi;
                if (NW == 2) *reinterpret_cast<unsigned short *>(Wl + (size_t)(p.CD >> 6) * wplane + dst[u]) = lo;
            }
    }

    // this wave's two 16-pixel segments: segment g = 2 * wave + i -> tile row g >> 1, columns 16 * (g & 1) ..
    int apix[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = 2 * wave + i;
        apix[i] = ((g >> 1) * p.HWc + 16 * (g & 1) + l15) * PROW + kg * 16;
    }
    const bool nvalid = l15 < p.F;
    const int wrow0 = l15 * KK * WROW + kg * 16;
    const float inv = 1.f / p.wscale;
    const int act_n = (int)((p.acts >> (4 * l15)) & 15);
    const float bias_n = (p.bias && nvalid) ? p.bias[l15] : 0.f;

    constexpr int HT = 12;                                                // float4 loads per thread: 10 x 38 pixels x 8 quads / 256
    float4 hreg[HT];
    auto halo_load = [&](int t, int kh) {
        int b, y0, x0;
        const int tx = t % p.tiles_x, r1 = t / p.tiles_x;
        b = r1 / p.tiles_y;
        y0 = (r1 % p.tiles_y) * TH;
        x0 = tx * TW;
#pragma unroll
        for (int it = 0; it < HT; ++it) {
            const int i = tid + it * NTHR;
            hreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it * NTHR < hpix * 8 && i < hpix * 8) {
                const int px = i >> 3, c4 = i & 7;
                const int hy = px / p.HWc, hx = px - hy * p.HWc;
                const int gy = y0 - p.pad + hy, gx = x0 - p.pad + hx;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    hreg[it] = *reinterpret_cast<const float4 *>(p.D + (((size_t)b * p.H + gy) * p.W + gx) * p.CD + kh * 32 + c4 * 4);
            }
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int it = 0; it < HT; ++it) {
            const int i = tid + it * NTHR;
            if (it * NTHR < hpix * 8 && i < hpix * 8) {
                const int px = i >> 3, c4 = i & 7;
                const float v[4] = {hreg[it].x, hreg[it].y, hreg[it].z, hreg[it].w};
                unsigned short h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split16<FP16>(v[e], h[e], l[e]);
                *reinterpret_cast<uint2 *>(Dl + px * PROW + c4 * 8) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                if (ND == 2)
                    *reinterpret_cast<uint2 *>(Dl + hpix * PROW + px * PROW + c4 * 8) =
                        make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
            }
        }
    };

    const int nkh = p.CD >> 5;
    const int t_begin = blockIdx.x * p.strip, t_end = min(p.ntiles, t_begin + p.strip);
    if (t_begin < t_end) halo_load(t_begin, 0);
    for (int t = t_begin; t < t_end; ++t) {
        f32x4v acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < nkh; ++kh) {
            __syncthreads();                                              // the previous image (and the weights) are done / ready
            halo_store();
            __syncthreads();
            if (kh + 1 < nkh) halo_load(t, kh + 1);                       // the next image flies during the MFMAs
            else if (t + 1 < t_end) halo_load(t + 1, 0);
            const unsigned char *wbase = Wl + (size_t)(kh >> 1) * wplane + wrow0 + (kh & 1) * 64;
#pragma unroll 1
            for (int tap = 0; tap < KK; ++tap) {
                const int r = tap / p.KS, s_ = tap - r * p.KS;
                const int toff = (r * p.HWc + s_) * PROW;
                bf16x8 bh = {0, 0, 0, 0, 0, 0, 0, 0}, bl = bh;
                if (nvalid) {
                    bh = *reinterpret_cast<const bf16x8 *>(wbase + tap * WROW);
                    if (NW == 2) bl = *reinterpret_cast<const bf16x8 *>(wbase + (size_t)(p.CD >> 6) * wplane + tap * WROW);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned char *ap = Dl + apix[i] + toff;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ap);
                    if (ND == 2) acc[i] = mfma16x16<FP16>(*reinterpret_cast<const bf16x8 *>(ap + hpix * PROW), bh, acc[i]);
                    if (NW == 2) acc[i] = mfma16x16<FP16>(ah, bl, acc[i]);
                    acc[i] = mfma16x16<FP16>(ah, bh, acc[i]);
                }
            }
        }
        if (nvalid) {
            const int tx = t % p.tiles_x, r1 = t / p.tiles_x;
            const int b = r1 / p.tiles_y, y0 = (r1 % p.tiles_y) * TH, x0 = tx * TW;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int g = 2 * wave + i;
                const int oy = y0 + (g >> 1), oxb = x0 + 16 * (g & 1) + 4 * kg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][e] * inv + bias_n;
                    p.Out[(((size_t)b * p.H + oy) * p.W + oxb + e) * p.F + l15] = hoig_act(v, act_n, p.slope);
                }
            }
        }
    }
}

bool thin_geometry(const hoig_conv_desc *d) {
    return !d->transposed && d->stride == 1 && d->R == d->S && (d->R & 1) && d->R <= 7 && 2 * d->pad == d->R - 1 &&
           d->Ho == d->Hi && d->Wo == d->Wi && (d->Hi % TH) == 0 && (d->Wi % TW) == 0;
}

void fill_common(ThinArgs &a, const hoig_conv_desc *d, int F, int CD) {
    a.Bn = d->B; a.H = d->Hi; a.W = d->Wi; a.F = F; a.CD = CD; a.KS = d->R; a.pad = d->pad;
    a.K = d->R * d->S * F;
    a.KP = (a.K + 15) / 16 * 16;
    a.N = a.K;
    a.NP = (a.N + 31) / 32 * 32;
    a.tiles_x = d->Wi / TW; a.tiles_y = d->Hi / TH;
    a.ntiles = d->B * a.tiles_x * a.tiles_y;
    a.HR = TH + d->R - 1;
    a.CW = (TW + d->R - 1 + 8 + 1) & ~1;            // (+8: a fragment's eight-pixel run may start in the last columns)
    a.bias = nullptr; a.act = HOIG_ACT_NONE; a.slope = 0.f; a.accumulate = 0; a.wscale = 1.f; a.strip = 1; a.Ws = nullptr; a.stats = nullptr;
}

template <bool FP16>
int launch_gemm(const ThinArgs &a, int precision, hipStream_t st) {
    const int nt = precision == HOIG_PREC_BF16 ? 1 : 2, nw = precision == HOIG_PREC_BF16X3 ? 2 : 1;
    const size_t smem = (size_t)nw * 64 * (a.KP + 8) * 2 + (size_t)a.KP * 2 + (size_t)nt * a.F * a.HR * a.CW * 2;
    if (smem > 160 * 1024) return HOIG_EUNSUPPORTED;
    dim3 grid((unsigned)hoig_cdiv(a.ntiles, a.strip), a.CD / 64);
#define HOIG_THIN_GEMM(NT_, NW_)                                                                                         \
    do {                                                                                                                 \
        static hoig_once attr;                                                                                        \
        if (!attr.done()) {                                                                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&thin_gemm_kernel<FP16, NT_, NW_>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)               \
                return HOIG_ELAUNCH;                                                                                     \
            attr.set();                                                                                                 \
        }                                                                                                                \
        thin_gemm_kernel<FP16, NT_, NW_><<<grid, NTHR, smem, st>>>(a);                                                   \
    } while (0)
    if (nt == 2 && nw == 2) HOIG_THIN_GEMM(2, 2);
    else if (nt == 2) HOIG_THIN_GEMM(2, 1);
    else HOIG_THIN_GEMM(1, 1);
#undef HOIG_THIN_GEMM
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

template <int NPW, int WN>
int launch_wgrad_t(const ThinArgs &a, int nt, int nd, dim3 grid, size_t smem, hipStream_t st) {
#define HOIG_THIN_WG(NT_, ND_)                                                                                           \
    do {                                                                                                                 \
        static hoig_once attr;                                                                                        \
        if (!attr.done()) {                                                                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&thin_wgrad_kernel<NPW, WN, NT_, ND_>),               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)               \
                return HOIG_ELAUNCH;                                                                                     \
            attr.set();                                                                                                 \
        }                                                                                                                \
        thin_wgrad_kernel<NPW, WN, NT_, ND_><<<grid, NTHR, smem, st>>>(a);                                               \
    } while (0)
    if (nt == 2 && nd == 2) HOIG_THIN_WG(2, 2);
    else if (nt == 2) HOIG_THIN_WG(2, 1);
    else if (nd == 2) HOIG_THIN_WG(1, 2);
    else HOIG_THIN_WG(1, 1);
#undef HOIG_THIN_WG
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace

// forward of a thin-INPUT convolution (Ci <= 8, Co % 64 == 0); HOIG_EUNSUPPORTED otherwise
int hoig_conv_thin_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, hipStream_t st, float *stats) {
    if (d->precision == HOIG_PREC_F32 || !thin_geometry(d) || d->Ci > (d->R <= 3 ? 12 : 8) || d->Ci < 1 || (d->Co % 64)) return HOIG_EUNSUPPORTED;
    ThinArgs a;
    fill_common(a, d, d->Ci, d->Co);
    a.T = x; a.D = w; a.bias = bias; a.Out = y; a.stats = stats;
    a.flip = 0;
    a.sj = d->R * d->S * d->Ci; a.st = d->Ci; a.sf = 1;                  // w[co][tap][ci]
    a.act = d->act; a.slope = d->slope;
    a.wscale = 256.f;
    a.strip = a.ntiles >= 4096 ? 8 : (a.ntiles >= 1024 ? 4 : 1);
    return launch_gemm<true>(a, d->precision, st);
}

// data gradient of a thin-OUTPUT convolution (Co <= 8, Ci % 64 == 0): dx[p][ci] = sum dy[p - off(tap)][co] w[co][tap][ci]
int hoig_conv_thin_dgrad(const hoig_conv_desc *d, const float *dy, const float *w, float *dx, int accumulate, hipStream_t st) {
    if (d->precision == HOIG_PREC_F32 || !thin_geometry(d) || d->Co > 8 || d->Co < 1 || (d->Ci % 64)) return HOIG_EUNSUPPORTED;
    ThinArgs a;
    fill_common(a, d, d->Co, d->Ci);
    a.T = dy; a.D = w; a.Out = dx;
    a.flip = 1;
    a.sj = 1; a.st = d->Ci; a.sf = d->R * d->S * d->Ci;                  // w[co][tap][ci], j = ci, f = co
    a.accumulate = accumulate;
    a.strip = a.ntiles >= 4096 ? 8 : (a.ntiles >= 1024 ? 4 : 1);
    return launch_gemm<false>(a, d->precision, st);
}

// does hoig_conv_thin_wgrad take this layer (and so want the launching stream's scratch block)?
bool hoig_conv_thin_wgrad_applies(const hoig_conv_desc *d) {
    if (d->precision == HOIG_PREC_F32 || !thin_geometry(d)) return false;
    return (d->Ci <= (d->R <= 3 ? 12 : 8) && (d->Co % 64) == 0) || (d->Co <= 8 && (d->Ci % 64) == 0);
}

// weight gradient of a thin-input (Ci <= 8, Co % 64 == 0) or thin-output (Co <= 8, Ci % 64 == 0) convolution
int hoig_conv_thin_wgrad(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, hipStream_t st) {
    if (d->precision == HOIG_PREC_F32 || !thin_geometry(d)) return HOIG_EUNSUPPORTED;
    const bool thin_in = d->Ci <= (d->R <= 3 ? 12 : 8) && (d->Co % 64) == 0, thin_out = d->Co <= 8 && (d->Ci % 64) == 0;
    if (!thin_in && !thin_out) return HOIG_EUNSUPPORTED;
    ThinArgs a;
    int nt, nd;                                  // 16-bit planes of the thin / dense operand
    if (thin_in) {                               // D = dy (split in every multi-term mode), T = x
        fill_common(a, d, d->Ci, d->Co);
        a.T = x; a.D = dy; a.flip = 0;
        a.sj = d->R * d->S * d->Ci; a.st = d->Ci; a.sf = 1;              // dW[co][tap][ci]
        nd = d->precision == HOIG_PREC_BF16 ? 1 : 2;
        nt = d->precision == HOIG_PREC_BF16X3 ? 2 : 1;
    } else {                                     // D = x, T = dy (mirrored taps)
        fill_common(a, d, d->Co, d->Ci);
        a.T = dy; a.D = x; a.flip = 1;
        a.sj = 1; a.st = d->Ci; a.sf = d->R * d->S * d->Ci;              // dW[co][tap][ci], j = ci, f = co
        nt = d->precision == HOIG_PREC_BF16 ? 1 : 2;
        nd = d->precision == HOIG_PREC_BF16X3 ? 2 : 1;
    }
    a.Out = dw;
    const int nfr = a.NP / 32;
    if (nfr > 16) return HOIG_EUNSUPPORTED;
    size_t smem = (size_t)nd * TH * TW * DROW + (size_t)nt * a.F * a.HR * a.CW * 2;
    const size_t gsm = (size_t)64 * a.NP * 4;
    if (gsm > smem) smem = gsm;
    if (smem > 160 * 1024) return HOIG_EUNSUPPORTED;
    int nwg = 512;                               // persistent workgroups: each adds its 64 x N partial once (fp32 atomics)
    if (smem > 80 * 1024 || a.N >= 128) nwg = 256;
    if (nwg > a.ntiles) nwg = a.ntiles;
    const int groups = a.CD / 64;
    while (nwg > 64 && (size_t)nwg * groups * 64 * a.NP * 4 > WS_SLOT_BYTES) nwg >>= 1;
    dim3 grid(nwg, groups);
    a.Ws = nwg > 8 ? thin_workspace(st, (size_t)nwg * groups * 64 * a.NP * 4) : nullptr;
    int rc;
    if (nfr == 1) rc = launch_wgrad_t<1, 1>(a, nt, nd, grid, smem, st);
    else if (nfr == 2) rc = launch_wgrad_t<2, 1>(a, nt, nd, grid, smem, st);
    else if (nfr <= 4) rc = launch_wgrad_t<2, 2>(a, nt, nd, grid, smem, st);
    else if (nfr <= 8) rc = launch_wgrad_t<2, 4>(a, nt, nd, grid, smem, st);
    else rc = launch_wgrad_t<4, 4>(a, nt, nd, grid, smem, st);
    if (rc != HOIG_OK || !a.Ws) return rc;
    thin_reduce_kernel<<<dim3((unsigned)hoig_cdiv(64 * a.N, NTHR), groups, (unsigned)hoig_cdiv(nwg, 32)), NTHR, 0, st>>>(a, nwg);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

// forward of a thin-OUTPUT convolution (Co <= 16, Ci % 64 == 0) with per-channel activation codes (4 bits per output channel;
// HOIG_ACT_*), or -- dgrad != 0 -- the data gradient of a thin-INPUT convolution (Ci <= 16 outputs, Co % 64 == 0 gathered)
int hoig_conv_thin_out(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                       unsigned long long acts, int dgrad, hipStream_t st) {
    if (d->precision == HOIG_PREC_F32 || !thin_geometry(d)) return HOIG_EUNSUPPORTED;
    const int F = dgrad ? d->Ci : d->Co, CD = dgrad ? d->Co : d->Ci;
    if (F < 1 || F > 16 || (CD % 64)) return HOIG_EUNSUPPORTED;
    // Every A fragment feeds ONE 16-wide MFMA here, so the kernel is bound by LDS reads, not by the matrix pipe: measured on the
    // 7x7 heads 228 / 362 us (64 -> 1 / 3, 8 images at 256x256) against 120 / 190 us on the fp32 VALU kernel, whose LDS reads
    // each feed 7 taps x Co FMAs -- but 84 against 299 us on the 3x3 data gradient of VGG conv1_1.  Small kernels only.
    if (d->R > 3) return HOIG_EUNSUPPORTED;
    ThinOutArgs a;
    a.D = x; a.Wt = w; a.bias = bias; a.Out = y;
    a.Bn = d->B; a.H = d->Hi; a.W = d->Wi; a.F = F; a.CD = CD; a.KS = d->R; a.pad = d->pad;
    const int KK = d->R * d->S;
    if (!dgrad) { a.flip = 0; a.sf = KK * CD; a.st = CD; a.sj = 1; a.wscale = 256.f; }       // w[f][tap][j]
    else { a.flip = 1; a.sf = 1; a.st = F; a.sj = KK * F; a.wscale = 1.f; }                 // w[j][tap][f]
    a.acts = acts; a.slope = d->slope;
    a.tiles_x = d->Wi / TW; a.tiles_y = d->Hi / TH; a.ntiles = d->B * a.tiles_x * a.tiles_y;
    a.strip = a.ntiles >= 2048 ? 8 : (a.ntiles >= 512 ? 2 : 1);
    a.HR = TH + d->R - 1; a.HWc = TW + d->R - 1;
    if (a.HR * a.HWc * 8 > 12 * NTHR) return HOIG_EUNSUPPORTED;
    const int nd = d->precision == HOIG_PREC_BF16 ? 1 : 2, nw = d->precision == HOIG_PREC_BF16X3 ? 2 : 1;
    const size_t smem = (size_t)nw * (CD / 64) * F * KK * 144 + (size_t)nd * a.HR * a.HWc * 80;
    if (smem > 160 * 1024) return HOIG_EUNSUPPORTED;
    const unsigned grid = (unsigned)hoig_cdiv(a.ntiles, a.strip);
#define HOIG_THIN_OUT(FP16_, ND_, NW_)                                                                                   \
    do {                                                                                                                 \
        static hoig_once attr;                                                                                        \
        if (!attr.done()) {                                                                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&thin_out_kernel<FP16_, ND_, NW_>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)               \
                return HOIG_ELAUNCH;                                                                                     \
            attr.set();                                                                                                 \
        }                                                                                                                \
        thin_out_kernel<FP16_, ND_, NW_><<<grid, NTHR, smem, st>>>(a);                                                   \
    } while (0)
    if (!dgrad) {
        if (nd == 2 && nw == 2) HOIG_THIN_OUT(true, 2, 2);
        else if (nd == 2) HOIG_THIN_OUT(true, 2, 1);
        else HOIG_THIN_OUT(true, 1, 1);
    } else {
        if (nd == 2 && nw == 2) HOIG_THIN_OUT(false, 2, 2);
        else if (nd == 2) HOIG_THIN_OUT(false, 2, 1);
        else HOIG_THIN_OUT(false, 1, 1);
    }
#undef HOIG_THIN_OUT
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
