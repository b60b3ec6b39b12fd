// 3x3 stride-1 "same" convolution FORWARD through Winograd F(2x2,3x3) on three fp16 terms (round 6).
//
//   Y = A^T [ sum_ci U (.) V ] A,   U = G (2^8 g) G^T  (per output / input channel, 4 x 4: hoig_pack_conv_weight_wino, once per weight
//   version),   V = B^T d B  (per 4 x 4 input patch of a 2 x 2 output tile and input channel: made here, from the fp32 tensor, and split
//   hi | lo AFTER the transform -- tools/emulate_winograd.py: the forward's numerics are those of the direct three-term kernel).
//   16 element-wise products per 2 x 2 outputs instead of 36: 2.25x fewer MFMAs -- and 16 accumulators per 2 x 2 outputs instead of 4.
//
// Workgroup: 256 threads (four waves, ONE per SIMD: the accumulators take 256 registers), a tile of 64 Winograd tiles (8 x 8 tiles =
// 16 x 16 output pixels of one image) x 64 output channels.  Wave w owns the four Winograd positions (w, 0..3) for the whole tile:
// acc[4 positions][4 channel fragments][4 tile fragments] on v_mfma_f32_16x16x32_f16 (A = U: rows = output channels, B = V: columns =
// tiles; a lane ends up with four consecutive channels of one tile).  Per 32-channel block:
//   * the fp32 halo (18 x 18 pixels) comes in two halves of 16 channels (20 KB each would not fit twice beside V), global -> registers
//     (both halves requested a whole MFMA phase ahead) -> LDS;
//   * TRANSFORM phase (all threads; thread = (tile, four channels)): 16 patch pixels from LDS, B^T d B in fp32, split, 16 positions x
//     (hi, lo) x 8 B into the V image -- the same [row][32 B] half-image layout as conv_halo16.hip's operands (conflict-free
//     ds_read_b128 fragments), 16 positions x 2 planes x 4 KB = 132 KB, SINGLE-buffered: V does not fit twice;
//   * MFMA phase (wave = its four positions): U fragments straight from global memory into registers (1 KB per fragment, contiguous:
//     the pack kernel writes them in fragment order; each position's U is read by ONE wave, so LDS would only add a hop), V fragments
//     from LDS, 4 x 4 x 4 x 3 = 192 MFMAs per wave.
// The two phases alternate (V is single-buffered).  MEASURED (profiles/r06_winograd_ab.txt): 87.6 us per round of 256 workgroups
// (8 images of 512 -> 512 at 32 x 32) -- 23 us of MFMA issue floor, ~38 us of transform phase (bound by the LDS STORE path: 128 KB of
// V per 32-channel block through VGPRs at <= 85 B/clk, not by its VALU), halo gathers and barriers; 16 images = two rounds = 183 us
// against the direct kernel's 171.  The product path does NOT call this kernel: it stays an entry point of
// the library, parity-tested (tests/test_conv_wino_gpu.py), for launches one round covers.
// Epilogue: Z_i[b] = sum_j A^T[b][j] M[i][j] in the owning wave, exchanged through LDS (V is dead by then), Y[a][b] = sum_i A^T[a][i] Z_i[b],
// scale 2^-8, bias, activation, 16-B stores.
#include "conv_bf16_common.h"
#include "conv_m16_common.h"
#include <cstdlib>

namespace hoig_detail {
namespace {

constexpr int W_TILES = 64, W_CT = 64, W_NT = 256;
constexpr int W_HS = 18, W_HPX = W_HS * W_HS;                 // halo of a 16 x 16 pixel tile
constexpr int W_HROW = 80;                                    // bytes of a halo pixel in LDS: 16 channels fp32 + 16 B (bank spread)
constexpr int W_VHALF = W_TILES * 32;                         // one half image of a (position, plane): [tile][32 B]
constexpr int W_V23 = W_VHALF + 64;                           // second half image 64 B past a multiple of 128 B (conv_halo16.hip)
constexpr int W_VPLANE = round128(W_V23 + W_VHALF);           // 4224
constexpr int W_VBYTES = 16 * 2 * W_VPLANE;                   // 135 168
constexpr int W_HBYTES = W_HPX * W_HROW;                      // 25 920
constexpr int W_ZROW = 272;                                   // bytes of a (row, b, tile) record of the epilogue exchange: 64 floats + 16 B
constexpr int W_ZBYTES = 4 * 2 * W_TILES * W_ZROW;            // 139 264 (overlays V and the halo)
constexpr int W_LDS = W_VBYTES + W_HBYTES;                    // 161 088 <= 163 840
static_assert(W_ZBYTES <= W_LDS, "the epilogue exchange overlays the V image and the halo");
constexpr int W_HSL = (W_HPX * 4 + W_NT - 1) / W_NT;          // float4 slices of a halo half per thread: 6

struct WinoArgs {
    const float *X;
    const unsigned short *Uh, *Ul;             // [16 positions][Co / 16][Ci / 32][64 lanes][8] fp16: MFMA A fragments, in order
    const float *bias;
    float *Y;
    int Bn, H, W, Ci, Co;
    int act;
    float slope;
    int tiles_x, tiles_y, nblk_n, nblk;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// split4h (conv_bf16_common.h) in half the instructions: lo = fp16(v - hi) as ONE v_fma_mix{lo,hi}_f16 per value (the fp16 source is
// widened inside the FMA; v - hi is exact in fp32, so the single rounding is the same one) instead of convert + subtract + a share of
// a packed convert.  The transform phase is VALU-bound: 16 of these per thread and channel half.
__device__ __forceinline__ unsigned lo_pair(float a, float b, unsigned hi) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo)
        : "v"(a), "v"(hi), "v"(b));
    return lo;
}
__device__ __forceinline__ void split4w(const float4 v, uint2 &hi, uint2 &lo) {
    hi.x = cvt2h(v.x, v.y);
    hi.y = cvt2h(v.z, v.w);
    lo.x = lo_pair(v.x, v.y, hi.x);
    lo.y = lo_pair(v.z, v.w, hi.y);
}

// KO: diagnostic instantiations (HOIG_WINO_KO; results are WRONG with any bit set): 1 no transform, 2 no MFMA phase, 4 no halo traffic
template <int KO>
__global__ __launch_bounds__(W_NT) void conv_wino_kernel(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *const Vb = smem, *const Hb = smem + W_VBYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    // channel-tile major: the workgroups of one channel tile -- the readers of its 2 MB of U -- are neighbours in the remapped order,
    // i.e. share an XCD's L2
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int n_mt = p.nblk / p.nblk_n;
    int mt = tile % n_mt;
    const int n0 = (tile / n_mt) * W_CT;
    const int tx_ = mt % p.tiles_x;
    mt /= p.tiles_x;
    const int ty_ = mt % p.tiles_y, b = mt / p.tiles_y;
    const int y0 = ty_ * 16, x0 = tx_ * 16;
    const int nkb = p.Ci >> 5;

    f32x4 acc[4][4][4];                        // [position of this wave][channel fragment][tile fragment]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[q][j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- halo half: global -> registers -> LDS.  slice i -> (pixel i >> 2, float4 i & 3 of the half's 16 channels).  The loads are
    // UNCONDITIONAL (addresses clamped into the image, padding zeroed at the LDS store): a load under a branch makes the compiler's
    // wait-count bookkeeping merge the two paths, and every later wait for an OLDER load degrades to vmcnt(0).
    float4 hreg0[W_HSL], hreg1[W_HSL];           // the two halves of ONE block: requested together, a whole MFMA phase ahead
    unsigned hoff[W_HSL];                        // BYTE offset of the slice's pixel in the image (clamped); bit 31: padding
#pragma unroll
    for (int sl = 0; sl < W_HSL; ++sl) {
        const int i = min(tid + W_NT * sl, W_HPX * 4 - 1);
        const int pix = i >> 2, c4 = i & 3;
        const int hy = pix / W_HS, hx = pix - hy * W_HS;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const unsigned o = (unsigned)((min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1)) * p.Ci + c4 * 4) * 4u;
        hoff[sl] = in ? o : (o | 0x80000000u);   // (the address stays valid either way; an image is < 2 GB)
    }
    // (uniform base + 32-bit lane offset: the scalar-base form of global_load, no 64-bit address arithmetic per load)
    const unsigned char *const img0 = reinterpret_cast<const unsigned char *>(p.X + (size_t)b * p.H * p.W * p.Ci);
    auto halo_load = [&](int kb, int hh, float4 (&hreg)[W_HSL]) {
        const unsigned char *img = img0 + (kb * 32 + hh * 16) * 4;
#pragma unroll
        for (int sl = 0; sl < W_HSL; ++sl) hreg[sl] = *reinterpret_cast<const float4 *>(img + (hoff[sl] & 0x7FFFFFFFu));
    };
    auto halo_store = [&](const float4 (&hreg)[W_HSL]) {
#pragma unroll
        for (int sl = 0; sl < W_HSL; ++sl) {
            const int i = tid + W_NT * sl;
            const float4 v = (hoff[sl] >> 31) ? make_float4(0.f, 0.f, 0.f, 0.f) : hreg[sl];
            if (i < W_HPX * 4) *reinterpret_cast<float4 *>(Hb + (i >> 2) * W_HROW + (i & 3) * 16) = v;
        }
    };

    // ---- transform: thread = (tile t, channel quad q of the half); V[i][j] = (B^T d B)[i][j], B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    const int tt = tid >> 2, tq = tid & 3;
    const int tty = tt >> 3, ttx = tt & 7;
    auto transform = [&](int hh) {
        const unsigned char *src = Hb + ((2 * tty) * W_HS + 2 * ttx) * W_HROW + tq * 16;
        float4 d[4][4];                        // the 4 x 4 patch, all sixteen reads in flight before the first add
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < 4; ++l) d[k][l] = *reinterpret_cast<const float4 *>(src + (k * W_HS + l) * W_HROW);
        __builtin_amdgcn_sched_barrier(0);
        float4 r[4][4];                        // after the row transform: r[i][l] = sum_k B^T[i][k] d[k][l]
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float4 d0 = d[0][l], d1 = d[1][l], d2 = d[2][l], d3 = d[3][l];
            r[0][l] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            r[1][l] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            r[2][l] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            r[3][l] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
        unsigned char *dst = Vb + hh * W_V23 + tt * 32 + tq * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v[4];
            v[0] = make_float4(r[i][0].x - r[i][2].x, r[i][0].y - r[i][2].y, r[i][0].z - r[i][2].z, r[i][0].w - r[i][2].w);
            v[1] = make_float4(r[i][1].x + r[i][2].x, r[i][1].y + r[i][2].y, r[i][1].z + r[i][2].z, r[i][1].w + r[i][2].w);
            v[2] = make_float4(r[i][2].x - r[i][1].x, r[i][2].y - r[i][1].y, r[i][2].z - r[i][1].z, r[i][2].w - r[i][1].w);
            v[3] = make_float4(r[i][1].x - r[i][3].x, r[i][1].y - r[i][3].y, r[i][1].z - r[i][3].z, r[i][1].w - r[i][3].w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 hi, lo;
                split4w(v[j], hi, lo);
                unsigned char *o = dst + (i * 4 + j) * 2 * W_VPLANE;
                *reinterpret_cast<uint2 *>(o) = hi;
                *reinterpret_cast<uint2 *>(o + W_VPLANE) = lo;
            }
        }
    };

    // ---- U fragments of this wave's positions: global -> registers.  Fragment (position, channel block of 16, channel block kb) = 1 KB
    const size_t ufrag_stride_co = (size_t)nkb * 1024;                         // BYTES between two channel blocks of 16
    const size_t ufrag_stride_pos = (size_t)(p.Co >> 4) * ufrag_stride_co;
    const size_t u_first = (size_t)(wave * 4) * ufrag_stride_pos + (size_t)(n0 >> 4) * ufrag_stride_co;       // (uniform)
    const unsigned char *const uh0 = reinterpret_cast<const unsigned char *>(p.Uh) + u_first;
    const unsigned char *const ul0 = reinterpret_cast<const unsigned char *>(p.Ul) + u_first;
    const unsigned ulane = lane * 16;
    struct UF {
        u32x4 h[4], l[4];                      // [channel fragment] of ONE position
    };
    auto load_u = [&](UF &u, int kb, int q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t o = (size_t)q * ufrag_stride_pos + (size_t)j * ufrag_stride_co + (size_t)kb * 1024;  // (uniform)
            u.h[j] = *reinterpret_cast<const u32x4 *>(uh0 + o + ulane);
            u.l[j] = *reinterpret_cast<const u32x4 *>(ul0 + o + ulane);
        }
    };

    // ---- V fragments: lane -> tile 16 m + (lane & 15), 16-B chunk lane >> 4 of the tile's 64-B k-row (two half images)
    int vread[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) vread[m] = (lg >> 1) * W_V23 + (m * 16 + l15) * 32 + (lg & 1) * 16;
    struct VF {
        bf16x8 h[4], l[4];
    };
    auto read_v = [&](VF &f, int pos) {
        const unsigned char *vh = Vb + pos * 2 * W_VPLANE, *vl = vh + W_VPLANE;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f.h[m] = *reinterpret_cast<const bf16x8 *>(vh + vread[m]);
            f.l[m] = *reinterpret_cast<const bf16x8 *>(vl + vread[m]);
        }
    };
    // One position: 48 MFMAs, tile fragment m outermost -- once the twelve MFMAs that read V fragment m have issued, its registers take
    // the NEXT position's fragment m (ONE V buffer, refilled on the fly: a second one would push the kernel past 512 registers, and the
    // spills that follow put scratch traffic -- and its vmcnt(0) waits -- in the middle of the prefetch chain).
    auto mma_pos = [&](int q, const UF &u, VF &f, int next_pos) {
        const unsigned char *vh = Vb + next_pos * 2 * W_VPLANE, *vl = vh + W_VPLANE;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[q][j][m] = mfma_m16<true>(__builtin_bit_cast(bf16x8, u.h[j]), f.l[m], acc[q][j][m]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[q][j][m] = mfma_m16<true>(__builtin_bit_cast(bf16x8, u.l[j]), f.h[m], acc[q][j][m]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[q][j][m] = mfma_m16<true>(__builtin_bit_cast(bf16x8, u.h[j]), f.h[m], acc[q][j][m]);
            __builtin_amdgcn_sched_barrier(0);
            if (next_pos >= 0) {
                f.h[m] = *reinterpret_cast<const bf16x8 *>(vh + vread[m]);
                f.l[m] = *reinterpret_cast<const bf16x8 *>(vl + vread[m]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // U lives in THREE slots: positions 0, 1, 2 of the next block are requested right after the MFMAs of positions 3 (slot 0), 1, 2 of
    // this one have issued (a transform phase or more of lead); position 3 takes slot 0 as soon as position 0 is done with it, two
    // positions (~1500 cycles) ahead of its use -- its 8 KB come out of the XCD's L2, which holds this channel tile's 2 MB of U.
    UF u0, u1, u2;
    halo_load(0, 0, hreg0);
    halo_load(0, 1, hreg1);
    load_u(u0, 0, 0);
    load_u(u1, 0, 1);
    load_u(u2, 0, 2);
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        if (!(KO & 4)) halo_store(hreg0);      // (the halo buffer is free: a barrier closed the previous phase)
        __syncthreads();
        if (!(KO & 1)) transform(0);
        __syncthreads();
        if (!(KO & 4)) halo_store(hreg1);
        __syncthreads();
        if (!(KO & 1)) transform(1);
        __syncthreads();
        if (KO & 2) continue;
        const int kn = min(kb + 1, nkb - 1);   // (the last block re-requests itself: no branch around a load, see halo_load)
        if (!(KO & 4)) {                       // the next block's halo: in flight behind this block's MFMAs
            halo_load(kn, 0, hreg0);
            halo_load(kn, 1, hreg1);
        }
        VF f;
        const int pos0 = wave * 4;
        read_v(f, pos0);
        __builtin_amdgcn_sched_barrier(0);
        mma_pos(0, u0, f, pos0 + 1);
        load_u(u0, kb, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma_pos(1, u1, f, pos0 + 2);
        load_u(u1, kn, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_pos(2, u2, f, pos0 + 3);
        load_u(u2, kn, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_pos(3, u0, f, -1);
        load_u(u0, kn, 0);
        __syncthreads();                       // every wave has read V: the next block's transform may overwrite it
    }

    // ---- epilogue.  Row i = wave: Z[b] = sum_j A^T[b][j] M[i][j], A^T = [1 1 1 0; 0 1 -1 -1]; record (i, b, tile) = 64 channels
    float *const Zb = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 z0 = acc[0][j][m] + acc[1][j][m] + acc[2][j][m];
            const f32x4 z1 = acc[1][j][m] - acc[2][j][m] - acc[3][j][m];
            const int t = m * 16 + l15, c = j * 16 + lg * 4;
            *reinterpret_cast<f32x4 *>(reinterpret_cast<unsigned char *>(Zb) + ((wave * 2 + 0) * W_TILES + t) * W_ZROW + c * 4) = z0;
            *reinterpret_cast<f32x4 *>(reinterpret_cast<unsigned char *>(Zb) + ((wave * 2 + 1) * W_TILES + t) * W_ZROW + c * 4) = z1;
        }
    __syncthreads();
    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    const int cq = tid & 15;                   // channel quad of this thread (the same for all its items)
    const float4 bq = p.bias ? *reinterpret_cast<const float4 *>(p.bias + n0 + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv = 1.f / W_SCALE_F16;
#pragma unroll
    for (int it = 0; it < (W_TILES * 2 * 16) / W_NT; ++it) {
        const int e = it * W_NT + tid;
        const int bb = (e >> 4) & 1, t = e >> 5;
        const unsigned char *zr = reinterpret_cast<const unsigned char *>(Zb) + (bb * W_TILES + t) * W_ZROW + cq * 16;
        const float4 z0 = *reinterpret_cast<const float4 *>(zr);
        const float4 z1 = *reinterpret_cast<const float4 *>(zr + 2 * W_TILES * W_ZROW);
        const float4 z2 = *reinterpret_cast<const float4 *>(zr + 4 * W_TILES * W_ZROW);
        const float4 z3 = *reinterpret_cast<const float4 *>(zr + 6 * W_TILES * W_ZROW);
        const float ya[2][4] = {{z0.x + z1.x + z2.x, z0.y + z1.y + z2.y, z0.z + z1.z + z2.z, z0.w + z1.w + z2.w},
                                {z1.x - z2.x - z3.x, z1.y - z2.y - z3.y, z1.z - z2.z - z3.z, z1.w - z2.w - z3.w}};
        const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
        const int ox = x0 + 2 * (t & 7) + bb;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = y0 + 2 * (t >> 3) + a;
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fast_act(ya[a][k] * inv + bv[k], nslope, special, p.act, p.slope);
            *reinterpret_cast<float4 *>(p.Y + (((size_t)b * p.H + oy) * p.W + ox) * p.Co + n0 + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// U = G (2^8 g) G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], split to fp16 hi | lo and written in MFMA A-fragment order.  One wave per
// (16 output channels, 32 input channels): lane -> row lane & 15, eight consecutive input channels 8 (lane >> 4) ..; 16 x 1 KB stores.
__global__ __launch_bounds__(256) void pack_wino_kernel(const float *__restrict__ w, int Co, int Ci, unsigned short *__restrict__ uh,
                                                        unsigned short *__restrict__ ul) {
    const int lane = threadIdx.x & 63;
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nkb = Ci >> 5, ncb = Co >> 4;
    if (frag >= ncb * nkb) return;
    const int cb = frag / nkb, kb = frag - cb * nkb;
    const int co = cb * 16 + (lane & 15), ci = kb * 32 + (lane >> 4) * 8;
    float g[3][3][8];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float *src = w + (((size_t)co * 3 + r) * 3 + s) * Ci + ci;
            const float4 a = *reinterpret_cast<const float4 *>(src), c = *reinterpret_cast<const float4 *>(src + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) g[r][s][e] = v[e] * W_SCALE_F16;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t[3];                    // row i of G g: over the columns s
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    t[s] = i == 0 ? g[0][s][e] : i == 3 ? g[2][s][e] : i == 1 ? 0.5f * (g[0][s][e] + g[1][s][e] + g[2][s][e])
                                                                              : 0.5f * (g[0][s][e] - g[1][s][e] + g[2][s][e]);
                u[e] = j == 0 ? t[0] : j == 3 ? t[2] : j == 1 ? 0.5f * (t[0] + t[1] + t[2]) : 0.5f * (t[0] - t[1] + t[2]);
            }
            uint2 h0, l0, h1, l1;
            split4h(make_float4(u[0], u[1], u[2], u[3]), h0, l0);
            split4h(make_float4(u[4], u[5], u[6], u[7]), h1, l1);
            const size_t o = ((((size_t)(i * 4 + j) * ncb + cb) * nkb + kb) * 64 + lane) * 8;
            *reinterpret_cast<uint4 *>(uh + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            *reinterpret_cast<uint4 *>(ul + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
}

}  // namespace
}  // namespace hoig_detail

using namespace hoig_detail;

extern "C" int64_t hoig_wino_plane_halfs(int Co, int Ci) {
    if (Co <= 0 || Ci <= 0 || (Co & 15) || (Ci & 31)) return 0;
    return (int64_t)16 * Co * Ci;
}

extern "C" int hoig_pack_conv_weight_wino(const float *w, int Co, int Ci, uint16_t *u_hi, uint16_t *u_lo, hoig_stream_t stream) {
    if (!w || !u_hi || !u_lo || Co <= 0 || Ci <= 0) return HOIG_EINVAL;
    if ((Co & 15) || (Ci & 31)) return HOIG_EUNSUPPORTED;
    const int frags = (Co >> 4) * (Ci >> 5);
    pack_wino_kernel<<<(frags + 3) / 4, 256, 0, (hipStream_t)stream>>>(w, Co, Ci, u_hi, u_lo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_conv2d_fwd_wino(const hoig_conv_desc *d, const float *x, const uint16_t *u_hi, const uint16_t *u_lo, const float *bias,
                                    float *y, hoig_stream_t stream) {
    if (!d || !x || !u_hi || !u_lo || !y) return HOIG_EINVAL;
    if (d->transposed || d->stride != 1 || d->R != 3 || d->S != 3 || d->pad != 1 || d->Hi != d->Ho || d->Wi != d->Wo ||
        d->precision != HOIG_PREC_BF16X3)
        return HOIG_EUNSUPPORTED;
    if ((d->Hi & 15) || (d->Wi & 15) || (d->Ci & 31) || (d->Co & 63)) return HOIG_EUNSUPPORTED;
    WinoArgs a;
    a.X = x; a.Uh = u_hi; a.Ul = u_lo; a.bias = bias; a.Y = y;
    a.Bn = d->B; a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.act = d->act; a.slope = d->slope;
    a.tiles_x = a.W / 16; a.tiles_y = a.H / 16;
    a.nblk_n = a.Co / W_CT;
    a.nblk = a.Bn * a.tiles_x * a.tiles_y * a.nblk_n;
    static const int ko_env = getenv("HOIG_WINO_KO") ? atoi(getenv("HOIG_WINO_KO")) : 0;
    static hoig_once once;
    if (!once.done()) {
        bool ok = true;
#define HOIG_WINO_ATTR(K_) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wino_kernel<K_>), \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS) == hipSuccess;
        HOIG_WINO_ATTR(0) HOIG_WINO_ATTR(1) HOIG_WINO_ATTR(2) HOIG_WINO_ATTR(3) HOIG_WINO_ATTR(4) HOIG_WINO_ATTR(7)
#undef HOIG_WINO_ATTR
        if (!ok) return HOIG_ELAUNCH;
        once.set();
    }
    hipStream_t st = (hipStream_t)stream;
    switch (ko_env) {
        case 1: conv_wino_kernel<1><<<a.nblk, W_NT, W_LDS, st>>>(a); break;
        case 2: conv_wino_kernel<2><<<a.nblk, W_NT, W_LDS, st>>>(a); break;
        case 3: conv_wino_kernel<3><<<a.nblk, W_NT, W_LDS, st>>>(a); break;
        case 4: conv_wino_kernel<4><<<a.nblk, W_NT, W_LDS, st>>>(a); break;
        case 7: conv_wino_kernel<7><<<a.nblk, W_NT, W_LDS, st>>>(a); break;
        default: conv_wino_kernel<0><<<a.nblk, W_NT, W_LDS, st>>>(a);
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
