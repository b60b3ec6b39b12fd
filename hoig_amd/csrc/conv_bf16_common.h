// Shared pieces of the 16-bit-operand convolution kernels (conv_igemm_bf16.hip, conv_halo16.hip): the split helpers, the
// blocked weight-plane layout and the argument block of the LDS-halo kernels.
#pragma once
#include "common.h"

namespace hoig_detail {

// Number of 16-bit MFMA terms per algorithmic product, as the launchers' runtime `ns` and the kernels' NSX template value:
//   1 : a*b ~= ah*bh                       one plane per operand                      (HOIG_PREC_BF16)
//   2 : a*b ~= ah*bh + al*bh + ah*bl       both operands split hi + lo   (3 MFMAs)    (HOIG_PREC_BF16X3)
//   3 : a*b ~= ah*bh + al*bh               A split, B = its rounded hi plane only (2 MFMAs: the B-side -- weight -- LDS image,
//                                          its L2 stream and the ah*bl pass are dropped)                (HOIG_PREC_F16X2)
#define HOIG_NS_SWITCH(ns, ...)                                                \
    do {                                                                       \
        if ((ns) == 2) { constexpr int NSX = 2; __VA_ARGS__; }                 \
        else if ((ns) == 3) { constexpr int NSX = 3; __VA_ARGS__; }            \
        else { constexpr int NSX = 1; __VA_ARGS__; }                           \
    } while (0)
__host__ __device__ constexpr int ns_a(int nsx) { return nsx == 1 ? 1 : 2; }
__host__ __device__ constexpr int ns_b(int nsx) { return nsx == 2 ? 2 : 1; }
inline int ns_of_precision(int precision) {
    return precision == HOIG_PREC_BF16X3 ? 2 : (precision == HOIG_PREC_F16X2 ? 3 : 1);
}
inline bool is_16bit_precision(int precision) {
    return precision == HOIG_PREC_BF16X3 || precision == HOIG_PREC_BF16 || precision == HOIG_PREC_F16X2;
}

typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt2(float a, float b) {
    f2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_t));
}
// FORWARD operands are split on fp16 instead (11-bit mantissas: x = hi + lo to 2^-22, the three products to ~2^-21 at the
// same MFMA rate): the error of the forward point, not the backward arithmetic, sets the gradient parity of the fast mode
// (precision).  Forward activations are O(1)-O(100) (post-norm / images / VGG features), far inside fp16's
// range; weights are scaled by 2^8 when they are split (exact) and the accumulator by 2^-8 in the epilogue.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float W_SCALE_F16 = 256.f;
__device__ __forceinline__ unsigned cvt2h(float a, float b) {
    f2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_t));
}
__device__ __forceinline__ void split4h(const float4 v, uint2 &hi, uint2 &lo) {
    hi.x = cvt2h(v.x, v.y);
    hi.y = cvt2h(v.z, v.w);
    const h2_t h0 = __builtin_bit_cast(h2_t, hi.x), h1 = __builtin_bit_cast(h2_t, hi.y);
    lo.x = cvt2h(v.x - (float)h0[0], v.y - (float)h0[1]);
    lo.y = cvt2h(v.z - (float)h1[0], v.w - (float)h1[1]);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// hi = bf16(x), lo = bf16(x - hi) for four values
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo) {
    hi.x = cvt2(v.x, v.y);
    hi.y = cvt2(v.z, v.w);
    const float r0 = v.x - __uint_as_float(hi.x << 16), r1 = v.y - __uint_as_float(hi.x & 0xFFFF0000u);
    const float r2 = v.z - __uint_as_float(hi.y << 16), r3 = v.w - __uint_as_float(hi.y & 0xFFFF0000u);
    lo.x = cvt2(r0, r1);
    lo.y = cvt2(r2, r3);
}
template <bool F16>
__device__ __forceinline__ void split4t(const float4 v, uint2 &hi, uint2 &lo) {
    if constexpr (F16) split4h(v, hi, lo);
    else split4(v, hi, lo);
}

// Layout of a packed bf16 weight plane in HBM: 32(n) x 32(k) blocks, each 2 KB contiguous, block (n/32, k/32) at
// ((n/32) * (K/32) + k/32) * 1024.  The weight tile of one k-step (BN rows x 32 k) is then BN/32 fully used 2-KB runs
// instead of BN half-used 128-B lines K*2 bytes apart; N and K are multiples of 32 on this path.  Inside a block, row n
// holds its four 16-B chunks at position chunk ^ ((n >> 2) & 3): the block is byte for byte the XOR-swizzled LDS image
// the kernels read (lds_swz<32>), so a tile can be copied to LDS linearly -- by global_load_lds, 1 KB per wave-instruction.
__host__ __device__ __forceinline__ size_t plane_index(int n, int k, int K) {
    return ((size_t)(n >> 5) * (K >> 5) + (k >> 5)) * 1024 + (n & 31) * 32 + (((((k & 31) >> 3) ^ ((n >> 2) & 3))) << 3) +
           (k & 7);
}

struct HaloArgs {
    const float *A;
    const unsigned short *Wh, *Wl;
    const float *bias;
    float *C;
    int Bn, H, W, Cg;       // input == output spatial size (stride 1, "same" padding), gathered channels
    int N, K;               // output channels, KS*KS*Cg
    int pad;                // halo origin = tile origin - pad   (dgrad: KS-1-pad)
    int flip;               // 1: weight tap index = KS*KS-1 - (r*KS+s)   (data gradient)
    int act;
    float slope;
    int nblk_n, nblk;
    int tiles_x, tiles_y;
    int nmajor;                // tile index = channel tile * pixel tiles + pixel tile (3x3 kernel)
    int f16;                   // forward launch: fp16-split operands, weights pre-scaled by 2^8
    float oscale;              // accumulator scale of the epilogue (2^-8 or 1)
    // channel concatenation without the copy (3x3 kernel): the gathered tensor is [A | A2] along channels (A holds the
    // first cg1 of the Cg channels), and / or the output is [C | C2] (C receives the first n1 of the N columns)
    const float *A2;
    int cg1;
    float *C2;
    int n1;
    const float *addend;       // same shape as C (single-output launches only): C = conv + addend -- the data gradient that lands in
                               // a tensor with a second consumer adds that consumer's gradient itself (ops.conv2d_fork)
    // grouped launch (3x3 stride-1 kernel of conv_halo16.hip only): Bn counts the images of BOTH problems; images b >= b_split read and
    // write the *_g2 tensors (indexed from their own image 0).  0: one problem.
    // forward of an INFERENCE chain conv -> instance norm (+ affine) -> ReLU -> THIS convolution (3x3 stride-1 kernel of conv_halo16.hip
    // only): the gathered tensor is the RAW output of the first convolution; the halo loader applies x * in_scale[b][c] + in_shift[b][c]
    // (the norm folded to one FMA per element, per image and channel of the Cg gathered channels) and ReLU on channels >= in_relu_c0
    // to every pixel inside the image (the zero padding stays zero).  nullptr: off.
    const float *in_scale = nullptr, *in_shift = nullptr;
    int in_relu_c0 = 0;
    int b_split;
    const float *A_g2;
    const unsigned short *Wh_g2, *Wl_g2;
    const float *bias_g2, *addend_g2;
    float *C_g2;
    int a_split;               // 1: A is a PRE-SPLIT tensor, per pixel [hi: Cg bf16][lo: Cg bf16] (hoig_split_planes_bf16): the 3x3
                               // stride-1 data gradient on conv_halo16.hip copies it to LDS without splitting (no A2 then)
    float *stats;              // (nullable) [Bn][2][N] fp32 accumulators: += per-image, per-channel sum and sum of squares of the
                               // values written to C -- the statistics of the instance norm that reads C next
};

// geometry and argument block of the generic implicit-GEMM kernels (igemm_bf16_kernel in conv_igemm_bf16.hip, conv_igemm16.hip)
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
    int ksplit, steps_per_split;   // split-K over blockIdx.y (atomic epilogue into a zeroed output)
    int f16;                       // forward launch: fp16-split operands, weights pre-scaled by 2^8
    float oscale;                  // accumulator scale of the epilogue (2^-8 or 1)
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

// wgrad_flat.hip: weight gradient of valid 5x5 stride-1 convolutions on the flattened pixel axis
struct WFlatArgs {
    const float *X, *DY;
    float *DW, *DB;
    int Ci, Co, Wc, HWc, Q, Ho, Wo, HPOS;
    int nblk_ci, nblk, n_mtiles, mt_per_split;
};
int launch_wgrad_flat5(const float *x, const float *dy, float *dw, float *dbias, int Bn, int Hi, int Wi, int Ci, int Co, int ns,
                       hipStream_t st);

// conv_igemm16.hip: the generic implicit GEMM on v_mfma_f32_16x16x32 (same tiles, split-K rule and geometry handling as
// launch<BM, BN, WM, WN> of conv_igemm_bf16.hip; `cfg` = 0: 128x128 on 4 waves, 1: 128x128 on 8 waves, 2: 64x128, 3: 128x64, 4: 64x64)
int launch_igemm_m16(Args a, int ns, int cfg, hipStream_t st);

// conv_flat16.hip: valid KS x KS convolutions (and their data gradients) on a flattened pixel axis
struct FlatArgs {
    const float *A;            // source tensor [Bn][Hs][Ws][Cg]
    const unsigned short *Wh, *Wl;
    const float *bias;
    float *C;                  // destination [Bn][Hd][Wd][N]
    const float *addend;       // (nullable) same shape as C: C = conv + addend
    int Bn, Hc, Wc;            // canvas: the grid the taps walk (the input grid of the convolution)
    int Hs, Ws, oy, ox;        // the source grid and its offset on the canvas (zero elsewhere)
    int Hd, Wd;                // canvas positions (y < Hd, x < Wd) are outputs
    int KS, Cg, N, K, flip, f16;
    int act;
    float slope, oscale;
    int HWc, Q, HP;            // filled in by the launcher: Hc * Wc, Bn * Hc * Wc, halo positions per tile
    int nblk, nblk_n, steps_per_split;
};
int launch_flat_m16(FlatArgs a, int ns, hipStream_t st);

// argument block of the weight-gradient halo kernels (wgrad_halo_bf16_kernel in conv_igemm_bf16.hip, wgrad_dma.hip)
struct WHaloArgs {
    const float *DY, *X, *X2;  // X2 (nullable): the input is [X | X2] along channels, X holding the first ci1
    int ci1;
    float *DW, *DB;            // DB (nullable): bias gradient = column sums of dy, taken from the dy tiles as they are staged
    int Bn, H, W, Co, Ci;     // H, W: output (= dy) size
    int Hin, Win, pad;         // input (= x) size and padding
    int tout;                  // 1: the accumulator tile is [DY channel][X channel] but DW is laid out [X channel][tap][DY channel]
                               // (ConvTranspose2d stride 2: the plain operand is x, the gathered one dy -- roles swapped)
    int tiles_x, tiles_y, n_mtiles, mt_per_split;
    int nblk_ci, nblk;
    // grouped launch (wgrad_dma.hip only): Bn counts the images of BOTH problems; a workgroup whose pixel tiles lie in images >= b_split
    // reads DY_g2 / X_g2 (indexed from their own image 0) and adds into DW_g2; a workgroup's tile range never straddles the two
    int b_split;
    const float *DY_g2, *X_g2;
    float *DW_g2;
};

// wgrad_dma.hip: the stride-1 3x3 weight gradient from PRE-SPLIT dy (a.DY points at [pixel][2][Co] bf16: hoig_split_planes_bf16 or a
// producer's epilogue), staged by LDS-DMA into double-buffered tiles; `a` carries the 4 x 32-pixel tiling; HOIG_EUNSUPPORTED otherwise
int launch_halo_s2_m16p(const HaloArgs &a, int ns, bool scatter, bool rows8, hipStream_t st);      // conv_s2_16.hip
// dgrad_k128.hip: dX[M][N] = dY[M][128] W[N][128]^T for the attention MLP's first layer (N = 25 C up to 12 800; M % 128 == 0, N % 64 == 0)
int launch_dgrad_thin(const float *dy, const unsigned short *wh, const unsigned short *wl, float *dx, int M, int N, int ns, hipStream_t st);
int launch_wgrad_dma(const WHaloArgs &a, int ns, dim3 grid, hipStream_t st);

// conv_halo16.hip: the 3x3 stride-1 halo kernel on v_mfma_f32_16x16x32 (8 rows x 32 pixels x bn channels per workgroup, bn = 128
// or 64; `a` carries that tiling's geometry); HOIG_EUNSUPPORTED for shapes it has no tiling for
int launch_halo3_m16(HaloArgs a, int ns, int bn, hipStream_t st);
int launch_halo_s2_m16(const HaloArgs &a, int ns, bool scatter, hipStream_t st);

}  // namespace hoig_detail
