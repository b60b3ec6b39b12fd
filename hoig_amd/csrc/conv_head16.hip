// Forward of the generator's image / mask heads (generator.py:124,219-235: 7x7 stride-1 "same" convolutions, 64 input channels,
// <= 5 output channels, at full resolution) on v_mfma_f32_16x16x32_f16 with fp16-split operands (three terms: the forward
// arithmetic of HOIG_PREC_BF16X3).  With Co <= 5 the implicit GEMM's N is almost all padding; here the HORIZONTAL taps take the
// place of the missing output channels:
//
//     Z[p][(s, co)] = sum_{r, ci} x[p + (r - 3) * W][ci] * w[co][r][s][ci]          N = 7 * Co (35 of 48 for Co = 5), K = 7 * Ci
//     y[p][co]      = sum_s Z[p + (s - 3)][(s, co)]
//
// i.e. a 7-tap VERTICAL convolution with 7 * Co outputs per pixel -- 126 MFMAs per 16 pixels of an input row instead of the
// 294 of the Co -> 16 padded form -- followed by a 7-term horizontal shifted sum of its result.
//
// A workgroup owns TH output rows of a strip of <= 144 computed pixels (9 waves, one 16-pixel tile each) and marches down the
// TH + 6 input rows.  The pixels are the MFMA's A operand and come STRAIGHT from global memory into the lanes that need them
// (lane = pixel l15, 8 channels lg: 32 B of the pixel's 256-B channel vector; no LDS image of x, every input row read once per
// workgroup and split once); the weights are the B operand, staged once per workgroup as fragments in lane order (84 KB: one
// conflict-free ds_read_b128 each).  An input row feeds the seven output rows it is a tap of: seven rows of accumulators are live,
// in a ring whose slot is static in the 7-times unrolled row loop.  When an output row is complete its Z tile goes to LDS
// (double-buffered, one barrier per row), and all threads do the shifted sum, bias and activation and store the row.
// Neighbouring strips overlap by the 6 pixels the shifted sum needs (W = 256: two strips of 144 computed / 128 stored pixels).
#include "conv_bf16_common.h"
#include "tuning.h"

namespace hoig_detail {
namespace {

constexpr int H7 = 7, H7_TH = 16, H7_MAXT = 9, H7_OUTW = 128, H7_THREADS = H7_MAXT * 64;

__host__ __device__ constexpr int h7_spt(int co) { return 16 / co > H7 ? H7 : 16 / co; }            // horizontal taps per 16-column tile
__host__ __device__ constexpr int h7_nt(int co) { return (H7 + h7_spt(co) - 1) / h7_spt(co); }      // column tiles
__host__ __device__ constexpr int h7_zs(int co) { return H7 * co; }                                // floats per pixel of a Z row

template <int CO, int KSTEPS>
__global__ __launch_bounds__(H7_THREADS) void conv_head7_m16_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                     const float *__restrict__ bias, float *__restrict__ y,
                                                                     int B, int H, int W, unsigned long long acts, float slope,
                                                                     int n_strips, int n_chunks) {
    constexpr int SPT = h7_spt(CO), NT = h7_nt(CO), ZS = h7_zs(CO), CI = 32 * KSTEPS;
    constexpr int NFRAG = H7 * KSTEPS * NT;                           // (r, k-step, column tile) fragment pairs (hi, lo)
    constexpr int ZROW = (H7_MAXT * 16 + 6) * ZS;                     // a Z row with 3 zero pixels either side
    extern __shared__ unsigned char lds[];
    uint4 *wf = reinterpret_cast<uint4 *>(lds);                        // [NFRAG][2 planes][64 lanes] x 16 B
    float *zr = reinterpret_cast<float *>(lds + (size_t)NFRAG * 2 * 64 * 16);      // [2][ZROW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    int t = blockIdx.x;
    const int strip = t % n_strips;
    t /= n_strips;
    const int chunk = t % n_chunks, img = t / n_chunks;
    const int ox0 = strip * H7_OUTW, ow = min(H7_OUTW, W - ox0);
    const int cx0 = max(0, min(ox0 - 8, W - H7_MAXT * 16));           // first computed pixel of the strip
    const int ntiles = min(H7_MAXT, (W - cx0 + 15) >> 4);
    const int h0 = chunk * H7_TH, th = min(H7_TH, H - h0), n_in = th + H7 - 1;

    // ---- weights -> fragments: lane (column l15, k-chunk lg) of fragment (r, ks, nt) holds w[co][r][s][ks*32 + lg*8 .. +8] of the
    //      column's (s, co), scaled by 2^8 and split to fp16 hi / lo; columns past the tile's taps are zero
    for (int i = tid; i < NFRAG * 64; i += H7_THREADS) {
        const int L = i & 63, f = i >> 6;
        const int nt = f % NT, ks = (f / NT) % KSTEPS, r = f / (NT * KSTEPS);
        const int col = L & 15, s = nt * SPT + col / CO, co = col % CO;
        uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
        if (col < SPT * CO && s < H7) {
            const float *src = w + ((size_t)co * H7 * H7 + r * H7 + s) * CI + ks * 32 + (L >> 4) * 8;
            float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
            a.x *= W_SCALE_F16; a.y *= W_SCALE_F16; a.z *= W_SCALE_F16; a.w *= W_SCALE_F16;
            b.x *= W_SCALE_F16; b.y *= W_SCALE_F16; b.z *= W_SCALE_F16; b.w *= W_SCALE_F16;
            uint2 h0v, l0v, h1v, l1v;
            split4h(a, h0v, l0v);
            split4h(b, h1v, l1v);
            hi = make_uint4(h0v.x, h0v.y, h1v.x, h1v.y);
            lo = make_uint4(l0v.x, l0v.y, l1v.x, l1v.y);
        }
        wf[(f * 2 + 0) * 64 + L] = hi;
        wf[(f * 2 + 1) * 64 + L] = lo;
    }
    for (int i = tid; i < 2 * ZROW; i += H7_THREADS) zr[i] = 0.f;

    const bool tile_on = wv < ntiles;
    const int px = cx0 + wv * 16 + l15;
    const bool px_on = tile_on && px < W;
    const float *xcol = x + ((size_t)img * H * W + px) * CI + lg * 8;
    float4 raw[KSTEPS][2];
    auto fetch = [&](int it) {
        const int g = h0 - 3 + it;
        if (g >= 0 && g < H && px_on) {
            const float *p = xcol + (size_t)g * W * CI;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                raw[ks][0] = *reinterpret_cast<const float4 *>(p + ks * 32);
                raw[ks][1] = *reinterpret_cast<const float4 *>(p + ks * 32 + 4);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) raw[ks][0] = raw[ks][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    f32x4 acc[H7][NT];
#pragma unroll
    for (int q = 0; q < H7; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[q][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int base = 0; base < n_in; base += H7) {
#pragma unroll
        for (int j = 0; j < H7; ++j) {
            const int it = base + j;
            if (it >= n_in) break;
            const int g = h0 - 3 + it;
            const bool row_on = g >= 0 && g < H;
            bf16x8 ah[KSTEPS], al[KSTEPS];
            if (row_on && tile_on) {
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    uint2 h0v, l0v, h1v, l1v;
                    split4h(raw[ks][0], h0v, l0v);
                    split4h(raw[ks][1], h1v, l1v);
                    ah[ks] = __builtin_bit_cast(bf16x8, make_uint4(h0v.x, h0v.y, h1v.x, h1v.y));
                    al[ks] = __builtin_bit_cast(bf16x8, make_uint4(l0v.x, l0v.y, l1v.x, l1v.y));
                }
            }
            if (it + 1 < n_in) fetch(it + 1);
            if (row_on && tile_on) {
#pragma unroll
                for (int r = 0; r < H7; ++r) {
                    const int oi = it - r;                             // the output row (of this chunk) this tap feeds
                    if (oi < 0 || oi >= th) continue;
                    const int q = (j - r + H7) % H7;
#pragma unroll
                    for (int ks = 0; ks < KSTEPS; ++ks) {
                        bf16x8 bh[NT], bl[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int f = (r * KSTEPS + ks) * NT + nt;
                            bh[nt] = __builtin_bit_cast(bf16x8, wf[(f * 2 + 0) * 64 + lane]);
                            bl[nt] = __builtin_bit_cast(bf16x8, wf[(f * 2 + 1) * 64 + lane]);
                        }
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[ks]), __builtin_bit_cast(f16x8, bh[nt]), acc[q][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al[ks]), __builtin_bit_cast(f16x8, bh[nt]), acc[q][nt], 0, 0, 0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[ks]), __builtin_bit_cast(f16x8, bl[nt]), acc[q][nt], 0, 0, 0);
                    }
                }
            }
            const int oi = it - (H7 - 1);                              // the output row that this input row completed
            if (oi >= 0) {
                const int q = (j + 1) % H7;
                float *zrow = zr + (oi & 1) * ZROW;
                if (tile_on) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool col_on = l15 < SPT * CO && nt * SPT + l15 / CO < H7;
                        if (col_on) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) zrow[(3 + wv * 16 + 4 * lg + i) * ZS + nt * SPT * CO + l15] = acc[q][nt][i];
                        }
                        acc[q][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
                __syncthreads();
                float *yrow = y + (((size_t)img * H + h0 + oi) * W + ox0) * CO;
                const int lx0 = ox0 - cx0;
                for (int idx = tid; idx < ow * CO; idx += H7_THREADS) {
                    const int pxo = idx / CO, co = idx - pxo * CO;
                    const float *zp = zrow + (lx0 + pxo) * ZS + co;
                    float v = 0.f;
#pragma unroll
                    for (int s = 0; s < H7; ++s) v += zp[s * (ZS + CO)];
                    v = v * (1.f / W_SCALE_F16) + (bias ? bias[co] : 0.f);
                    yrow[idx] = hoig_act(v, (int)((acts >> (4 * co)) & 15), slope);
                }
            }
        }
    }
}

template <int CO, int KSTEPS>
int launch_head7(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, unsigned long long acts,
                 hipStream_t st) {
    constexpr int NFRAG = H7 * KSTEPS * h7_nt(CO);
    constexpr size_t lds = (size_t)NFRAG * 2 * 64 * 16 + (size_t)2 * (H7_MAXT * 16 + 6) * h7_zs(CO) * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    static hoig_once attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_head7_m16_kernel<CO, KSTEPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HOIG_ELAUNCH;
        attr.set();
    }
    const int n_strips = (d->Wi + H7_OUTW - 1) / H7_OUTW, n_chunks = (d->Hi + H7_TH - 1) / H7_TH;
    conv_head7_m16_kernel<CO, KSTEPS><<<d->B * n_chunks * n_strips, H7_THREADS, lds, st>>>(x, w, bias, y, d->B, d->Hi, d->Wi, acts, d->slope,
                                                                                   n_strips, n_chunks);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace
}  // namespace hoig_detail

// returns HOIG_EUNSUPPORTED unless: three-term forward arithmetic, 7x7 stride-1 "same", 64 input channels, <= 5 outputs
int hoig_conv_head7_m16(const hoig_conv_desc *d, const float *x, const float *w, const float *bias, float *y, unsigned long long acts,
                        hipStream_t st) {
    using namespace hoig_detail;
    if (d->precision != HOIG_PREC_BF16X3 || !hoig_tuning(HOIG_TUNE_HEAD16)) return HOIG_EUNSUPPORTED;
    if (d->transposed || d->stride != 1 || d->R != H7 || d->S != H7 || d->pad != 3 || d->Ho != d->Hi || d->Wo != d->Wi) return HOIG_EUNSUPPORTED;
    if (d->Ci != 64 || d->Co < 3 || d->Co > 5 || d->Wi < 16) return HOIG_EUNSUPPORTED;      // (1-2 outputs: never alone in this network)
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return HOIG_EUNSUPPORTED;
    switch (d->Co) {
        case 3: return launch_head7<3, 2>(d, x, w, bias, y, acts, st);
        case 4: return launch_head7<4, 2>(d, x, w, bias, y, acts, st);
        default: return launch_head7<5, 2>(d, x, w, bias, y, acts, st);
    }
}
