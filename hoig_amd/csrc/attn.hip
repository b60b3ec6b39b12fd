// Local-attention feature warping (ExtractorAttn, extract_attn.py:23-29) as a composition around the MFMA convolution
// kernels -- WITHOUT any 25x-sized tensor.  Reference dataflow: K1 block-extracts source (with flow) and target (zero flow)
// into [B,C,5H,5W] tensors, concatenates, conv k5/s5 (2C->128) + LeakyReLU(0.01) + conv1x1 (128->25) + softmax, K3 reshape,
// multiply with the extracted source, average-pool 5x5.  Restated per output pixel m and tap q = ty*5+tx (offsets ty-2,
// tx-2, block_extractor_kernel.cu:57-60):
//     hidden[m] = b1 + sum_q W1t[q] . target[clamp(m + off_q)]  +  sum_q W1s[q] . S[m][q]
//     S[m][q]   = K1 bilinear sample of source at m + flow(m) + off_q  (border-clamped taps, un-renormalised weights)
//     a[m]      = softmax_25( W2 . leaky(hidden[m]) + b2 );     out[m] = (1/25) sum_q a[m][q] * S[m][q]
// The 25 taps of a pixel sit on a regular grid: P = floor(m + flow(m)) is shared and so are the bilinear fractions w_ab
// (K1's coordinate is (flow + off) + m, block_extractor_kernel.cu:62-76: an integer shift of one real number), and every
// corner is clamped on its own, i.e. reads the REPLICATE-padded source at P + off_q + (a,b).  The sampling therefore
// commutes with the linear map over the taps:
//     sum_q W1s[q] . S[m][q] = sum_ab w_ab(m) * Gs[P(m) + (a,b)],   Gs[p] = sum_q W1s[q] . srcpad[p + off_q]
// Gs is an ordinary 5x5 convolution of the replicate-padded source (128 output channels, needed for p in [-2, H+1]: beyond,
// every tap reads the border pixel), exactly like the target half, and the source half of `hidden` is a bilinear read of it.
//     * target half: 5x5 valid conv of replicate_pad(target, 2)            -> Gt [B,H,W,128]      (conv kernels)
//     * source half: 5x5 valid conv of replicate_pad(source, 4)            -> Gs [B,H+4,W+4,128]  (conv kernels)
//     * hoig_attn_pixel_fwd: hidden = Gt + bilinear(Gs), softmax, and out = the 6x6 footprint of the source weighted by
//       k[i][j] = (1/25) sum_ab w_ab a[i-a][j-b] (36 L2-resident loads per pixel and channel vector)
// and the backward likewise: da_q from 36 footprint dot products <dout[m], src[..]>, dGs = bilinear^T(dhidden), the two
// 5x5 convolutions' data / weight gradients on the conv kernels, and the source gradient of the weighted sum as a footprint
// scatter.  (Round 1 materialised S [B,H,W,25C] -- 419 MB per 32x32 layer -- and dS; the reference materialises three such
// tensors plus gradients.)  K1's per-tap fraction differs from the shared one by at most one ulp of the coordinate (~1e-6).
// The source gradient is a scatter through the bilinear taps (K2, block_extractor_kernel.cu:158-161): sample positions
// stay within a few pixels of m (the flow is a normalised-coordinate difference read as pixels, generator.py:484-488),
// so each workgroup accumulates its tile's contributions in an LDS patch and flushes the patch with one
// global atomic per patch cell; taps that fall outside the patch take the global-atomic path directly.
#include "common.h"

namespace {

constexpr int KS = 5, NTAP = 25, NH = 128;

// Sampling frame of pixel coordinate `pos` under flow `f` (K1's centre tap: d = (f + 0) + pos, block_extractor_kernel.cu:62-76):
// integer cell and the fraction towards the next cell.  Tap t reads cells base + (t - 2) and base + (t - 2) + 1.
__device__ __forceinline__ void k1_frame(float f, int pos, int &base, float &w1) {
    const float d = f + (float)pos;
    const float fl = floorf(d);
    w1 = d - fl;
    base = (int)fminf(fmaxf(fl, -1048576.f), 1048576.f);      // (any cell this far out clamps to the border anyway)
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return max(min(v, hi), lo); }

// ---------------------------------------------------------------------------------------------- replicate padding
__global__ void replicate_pad_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C,
                                         int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p, CV = C >> 2;
    const int64_t n = (int64_t)B * Hp * Wp * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int px = (int)(t % Wp);
        t /= Wp;
        const int py = (int)(t % Hp), b = (int)(t / Hp);
        const int sy = max(min(py - p, H - 1), 0), sx = max(min(px - p, W - 1), 0);
        reinterpret_cast<float4 *>(y)[i] = *reinterpret_cast<const float4 *>(x + (((size_t)b * H + sy) * W + sx) * C + cv * 4);
    }
}

// dx[y][x] = sum of dy over the padded cells that replicate (y,x)  (gather form: no atomics)
__global__ void replicate_pad_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int B, int H, int W, int C,
                                         int p, const float *__restrict__ addend) {
    const int Hp = H + 2 * p, Wp = W + 2 * p, CV = C >> 2;
    const int64_t n = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H), b = (int)(t / H);
        const int py0 = y == 0 ? 0 : y + p, py1 = y == H - 1 ? Hp - 1 : y + p;
        const int px0 = x == 0 ? 0 : x + p, px1 = x == W - 1 ? Wp - 1 : x + p;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int py = py0; py <= py1; ++py)
            for (int px = px0; px <= px1; ++px) {
                const float4 v = *reinterpret_cast<const float4 *>(dy + (((size_t)b * Hp + py) * Wp + px) * C + cv * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        if (addend) {                       // the gradient that reached the unpadded tensor through another consumer
            const float4 a = reinterpret_cast<const float4 *>(addend)[i];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        reinterpret_cast<float4 *>(dx)[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------- per-pixel softmax
// Gs lives on the extended grid p in [-2, H+1] x [-2, W+1] (array index p + 2); beyond it every tap reads the border pixel,
// so a clamped read is exact for any flow.
constexpr int GPAD = 2, FP = KS + 1, NFP = FP * FP;      // 6x6 footprint of a pixel's 25 taps x 4 corners

struct Frame {
    int by, bx;          // cell of the centre tap's upper-left corner
    float wy1, wx1;      // fractions towards the lower / right neighbour
};
__device__ __forceinline__ Frame pixel_frame(const float *__restrict__ flow, int b, int rem, int hw, int y, int x) {
    Frame f;
    k1_frame(flow[((size_t)b * 2 + 0) * hw + rem], x, f.bx, f.wx1);
    k1_frame(flow[((size_t)b * 2 + 1) * hw + rem], y, f.by, f.wy1);
    return f;
}

// 8 pixels per workgroup: hidden = Gt + bilinear(Gs); 32 lanes per pixel compute the 25 logits + softmax; then
// out = sum over the 6x6 footprint of k[i][j] * source
__global__ __launch_bounds__(256) void attn_pixel_fwd_kernel(const float *__restrict__ gt, const float *__restrict__ gs,
                                                             const float *__restrict__ flow, const float *__restrict__ w2,
                                                             const float *__restrict__ b2, const float *__restrict__ src,
                                                             float *__restrict__ hidden, float *__restrict__ attn,
                                                             float *__restrict__ out, float *__restrict__ kfout, int B, int H,
                                                             int W, int C) {
    constexpr int PIX = 8;
    __shared__ float hs[PIX][NH];
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float as[PIX][32];
    __shared__ float kf[PIX][NFP];
    __shared__ Frame fr[PIX];
    const int tid = threadIdx.x, hw = H * W, M = B * hw;
    const int m0 = blockIdx.x * PIX;
    for (int i = tid; i < NTAP * NH; i += 256) w2s[i / NH][i % NH] = w2[i];
    if (tid < PIX && m0 + tid < M) {
        const int m = m0 + tid, b = m / hw, rem = m - b * hw;
        fr[tid] = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
    }
    __syncthreads();
    const int Hg = H + 2 * GPAD, Wg = W + 2 * GPAD;
    for (int i = tid; i < PIX * NH; i += 256) {
        const int p = i / NH, j = i % NH, m = m0 + p;
        float v = 0.f;
        if (m < M) {
            const int b = m / hw;
            const Frame f = fr[p];
            const int y0 = clampi(f.by, -GPAD, H + GPAD - 1) + GPAD, y1 = clampi(f.by + 1, -GPAD, H + GPAD - 1) + GPAD;
            const int x0 = clampi(f.bx, -GPAD, W + GPAD - 1) + GPAD, x1 = clampi(f.bx + 1, -GPAD, W + GPAD - 1) + GPAD;
            const float *g = gs + (size_t)b * Hg * Wg * NH + j;
            const float wy0 = 1.f - f.wy1, wx0 = 1.f - f.wx1;
            v = gt[(size_t)m * NH + j];
            v += (wx0 * wy0) * g[((size_t)y0 * Wg + x0) * NH];
            v += (f.wx1 * wy0) * g[((size_t)y0 * Wg + x1) * NH];
            v += (wx0 * f.wy1) * g[((size_t)y1 * Wg + x0) * NH];
            v += (f.wx1 * f.wy1) * g[((size_t)y1 * Wg + x1) * NH];
            hidden[(size_t)m * NH + j] = v;
        }
        hs[p][j] = v > 0.f ? v : 0.01f * v;            // LeakyReLU(0.01): generator.py:344 / extract_attn.py:19
    }
    __syncthreads();
    {
        const int p = tid >> 5, q = tid & 31, m = m0 + p;
        float logit = -INFINITY;
        if (q < NTAP) {
            float s = b2[q];
#pragma unroll 8
            for (int j = 0; j < NH; ++j) s += w2s[q][j] * hs[p][j];
            logit = s;
        }
        float mx = logit;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
        const float e = q < NTAP ? expf(logit - mx) : 0.f;
        float sum = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 32);
        const float a = e / sum;
        as[p][q] = q < NTAP ? a * (1.f / NTAP) : 0.f;  // avg_pool2d(5,5) of the product (extract_attn.py:28)
        if (q < NTAP && m < M) attn[(size_t)m * NTAP + q] = a;
    }
    __syncthreads();
    // footprint weights: tap (ty,tx) corner (a,b) reads cell (ty+a, tx+b) with weight w_ab
    for (int i = tid; i < PIX * NFP; i += 256) {
        const int p = i / NFP, c = i % NFP, ci = c / FP, cj = c % FP;
        const Frame f = fr[p];
        float k = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int ty = ci - a, tx = cj - bb;
                if (ty >= 0 && ty < KS && tx >= 0 && tx < KS)
                    k += (a ? f.wy1 : 1.f - f.wy1) * (bb ? f.wx1 : 1.f - f.wx1) * as[p][ty * KS + tx];
            }
        kf[p][c] = k;
        if (kfout && m0 + p < M) kfout[(size_t)(m0 + p) * NFP + c] = k;      // saved for the source-gradient gather
    }
    __syncthreads();
    const int CV = C >> 2;
    for (int i = tid; i < PIX * CV; i += 256) {
        const int p = i / CV, cv = i - p * CV, m = m0 + p;
        if (m >= M) continue;
        const int b = m / hw;
        const Frame f = fr[p];
        const float *s = src + (size_t)b * hw * C + cv * 4;
        int xs[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) xs[j] = clampi(f.bx - KS / 2 + j, 0, W - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < FP; ++r) {
            const float *row = s + (size_t)clampi(f.by - KS / 2 + r, 0, H - 1) * W * C;
            float4 v[FP];
#pragma unroll
            for (int j = 0; j < FP; ++j) v[j] = *reinterpret_cast<const float4 *>(row + (size_t)xs[j] * C);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const float w = kf[p][r * FP + j];
                acc.x += w * v[j].x; acc.y += w * v[j].y; acc.z += w * v[j].z; acc.w += w * v[j].w;
            }
        }
        *reinterpret_cast<float4 *>(out + (size_t)m * C + cv * 4) = acc;
    }
}

// E[m][i*6+j] = <dout[m], source cell (i,j) of pixel m's footprint> (36 dot products of length C per pixel).  One wave per
// pixel, lane = footprint cell: every lane streams ITS cell's channel row (a whole 128-B line per step, so nothing relies on
// L1 keeping 36 x #waves lines alive) against the pixel's dout row (wave-uniform loads); no cross-lane reduction at all.
// (Round 1 read the 25 x C sampled tensor here; the first S-free version reduced each cell across the lanes of a wave, 216
// shuffles per pixel, two pixels per wave in sequence: 200 us per launch.)
__global__ __launch_bounds__(256) void attn_edots_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                                         const float *__restrict__ dout, float *__restrict__ E, int B, int H,
                                                         int W, int C) {
    const int hw = H * W, M = B * hw;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const int b = m / hw, rem = m - b * hw;
    const Frame f = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
    const int cell = lane < NFP ? lane : NFP - 1;
    const int yy = clampi(f.by - KS / 2 + cell / FP, 0, H - 1), xx = clampi(f.bx - KS / 2 + cell % FP, 0, W - 1);
    const float *sp = src + (((size_t)b * H + yy) * W + xx) * C;
    const float *gp = dout + (size_t)m * C;
    float acc = 0.f;
    for (int c = 0; c < C; c += 32) {
        float4 sv[8], gv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = c + k * 4 < C;
            sv[k] = ok ? *reinterpret_cast<const float4 *>(sp + c + k * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            gv[k] = ok ? *reinterpret_cast<const float4 *>(gp + c + k * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += gv[k].x * sv[k].x + gv[k].y * sv[k].y + gv[k].z * sv[k].z + gv[k].w * sv[k].w;
    }
    if (lane < NFP) E[(size_t)m * NFP + lane] = acc;
}

// The same dot products with the lanes along the CHANNELS (every load a contiguous 512 B - 1 KB run of one source row): a
// pixel is served by LPP = 64 (C = 256, 512) or 32 (C = 128) lanes that keep 36 partial dots in registers, and the 36 sums
// over the lanes are taken by a halving butterfly -- at each step a lane keeps one half of its values and hands the other
// half to its partner: 18 + 9 + 5 + 3 + 2 + 1 = 38 shuffles per wave instead of 36 x 6.
template <int LPP, int VPL>
__global__ __launch_bounds__(256) void attn_edots_ch_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                                            const float *__restrict__ dout, float *__restrict__ E, int B, int H,
                                                            int W) {
    constexpr int C = LPP * VPL * 4, PPW = 64 / LPP;               // pixels per wave
    const int hw = H * W, M = B * hw;
    const int lane = threadIdx.x & 63, sub = lane / LPP, cl = lane % LPP;
    const int m = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + sub;
    const bool live = m < M;
    const int mm = live ? m : M - 1;
    const int b = mm / hw, rem = mm - b * hw;
    const Frame f = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
    float4 go[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) go[k] = *reinterpret_cast<const float4 *>(dout + (size_t)mm * C + (cl + k * LPP) * 4);
    const float *sb = src + (size_t)b * hw * C + cl * 4;
    float v[NFP];
#pragma unroll
    for (int r = 0; r < FP; ++r) {
        const float *row = sb + (size_t)clampi(f.by - KS / 2 + r, 0, H - 1) * W * C;
        float4 sv[FP][VPL];
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const float *sp = row + (size_t)clampi(f.bx - KS / 2 + j, 0, W - 1) * C;
#pragma unroll
            for (int k = 0; k < VPL; ++k) sv[j][k] = *reinterpret_cast<const float4 *>(sp + k * LPP * 4);
        }
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < VPL; ++k) a += go[k].x * sv[j][k].x + go[k].y * sv[j][k].y + go[k].z * sv[j][k].z + go[k].w * sv[j][k].w;
            v[r * FP + j] = a;
        }
    }
    // halving butterfly over the LPP lanes of the pixel: after the step with partner distance o the lane holds the sums over
    // its 2o-aligned group... of the index interval [lo, lo + n)
    int lo = 0, cnt = NFP;                     // the lane's live values are v[0 .. cnt) = sums of the indices [lo, lo + cnt)
#define HOIG_BFLY(N, O)                                                                                   \
    {                                                                                                      \
        constexpr int HALF = ((N) + 1) / 2;                                                                \
        const bool up = (cl & (O)) != 0;                                                                   \
        _Pragma("unroll") for (int j = 0; j < HALF; ++j) {                                                 \
            const float hi = (HALF + j < (N)) ? v[HALF + j] : 0.f;                                         \
            const float keep = up ? hi : v[j], send = up ? v[j] : hi;                                      \
            v[j] = keep + __shfl_xor(send, (O), 64);                                                       \
        }                                                                                                  \
        lo += up ? HALF : 0;                                                                               \
        cnt = up ? max(cnt - HALF, 0) : min(cnt, HALF);                                                    \
    }
    if (LPP == 64) {
        HOIG_BFLY(36, 32) HOIG_BFLY(18, 16) HOIG_BFLY(9, 8) HOIG_BFLY(5, 4) HOIG_BFLY(3, 2) HOIG_BFLY(2, 1)
        if (live && cnt > 0) E[(size_t)m * NFP + lo] = v[0];
    } else {
        HOIG_BFLY(36, 16) HOIG_BFLY(18, 8) HOIG_BFLY(9, 4) HOIG_BFLY(5, 2) HOIG_BFLY(3, 1)
        if (live && cnt > 0) E[(size_t)m * NFP + lo] = v[0];
        if (live && cnt > 1) E[(size_t)m * NFP + lo + 1] = v[1];
    }
#undef HOIG_BFLY
}

// da_q = (1/25) <dout[m], S[m][q]> = (1/25) sum_ab w_ab E[ty+a][tx+b], E[i][j] = <dout[m], source cell (i,j) of the footprint>;
// dlogit = a*(da - <a,da>); dW2 += dlogit (x) leaky(h); db2 += dlogit; dhidden = (W2^T dlogit) * leaky'(h).
// (The source gradient a_q/25*dout is attn_sample_bwd; the source gradient through `hidden` runs through dGs.)
// One 1024-thread workgroup per 32 pixels (x nit consecutive groups): dW2 / db2 are accumulated in registers over the
// workgroup's pixels and added to global memory ONCE -- with 8-pixel workgroups those 3225 fp32 atomics per workgroup, all
// workgroups on the same addresses, were a third of the kernel.
constexpr int APB_NT = 1024, APB_PIX = APB_NT / 32, APB_ITEMS = (NTAP * NH + APB_NT - 1) / APB_NT;
__global__ __launch_bounds__(APB_NT) void attn_pixel_bwd_kernel(const float *__restrict__ hidden, const float *__restrict__ attn,
                                                                const float *__restrict__ w2, const float *__restrict__ E,
                                                                const float *__restrict__ flow,
                                                                float *__restrict__ dhidden, float *__restrict__ dw2,
                                                                float *__restrict__ db2, int B, int H, int W, int C, int nit) {
    constexpr int PIX = APB_PIX;
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float dl[PIX][32];
    __shared__ float da[PIX][32];
    __shared__ float ef[PIX][NFP + 4];                   // footprint dot products E[i][j]
    __shared__ Frame fr[PIX];
    __shared__ float hl[PIX][NH];                        // leaky(hidden) of the group's pixels (read 25 times each for dW2)
    const int tid = threadIdx.x, hw = H * W, M = B * hw;
    for (int i = tid; i < NTAP * NH; i += APB_NT) w2s[i / NH][i % NH] = w2[i];
    float w2acc[APB_ITEMS], b2acc = 0.f;
#pragma unroll
    for (int k = 0; k < APB_ITEMS; ++k) w2acc[k] = 0.f;
    (void)C;
    for (int it = 0; it < nit; ++it) {
        const int m0 = (blockIdx.x * nit + it) * PIX;
        if (m0 >= M) break;                              // (uniform)
        __syncthreads();                                 // w2s loaded / the previous group's dl, da consumed
        if (tid < PIX && m0 + tid < M) {
            const int m = m0 + tid, b = m / hw, rem = m - b * hw;
            fr[tid] = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
        }
        __syncthreads();
        for (int i = tid; i < PIX * NFP; i += APB_NT) {
            const int p = i / NFP, cell = i % NFP;
            ef[p][cell] = m0 + p < M ? E[(size_t)(m0 + p) * NFP + cell] : 0.f;
        }
        __syncthreads();
        {
            const int p = tid >> 5, q = tid & 31, m = m0 + p;
            float d = 0.f;
            if (q < NTAP && m < M) {
                const Frame f = fr[p];
                const int ty = q / KS, tx = q % KS;
                const float wy0 = 1.f - f.wy1, wx0 = 1.f - f.wx1;
                d = (wx0 * wy0) * ef[p][ty * FP + tx] + (f.wx1 * wy0) * ef[p][ty * FP + tx + 1] +
                    (wx0 * f.wy1) * ef[p][(ty + 1) * FP + tx] + (f.wx1 * f.wy1) * ef[p][(ty + 1) * FP + tx + 1];
                d *= (1.f / NTAP);
            }
            da[p][q] = d;
            const float a = (q < NTAP && m < M) ? attn[(size_t)m * NTAP + q] : 0.f;
            float dot = a * d;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 32);
            dl[p][q] = a * (d - dot);
        }
        __syncthreads();
        for (int i = tid; i < PIX * NH; i += APB_NT) {
            const int p = i / NH, j = i % NH, m = m0 + p;
            if (m >= M) {
                hl[p][j] = 0.f;
                continue;
            }
            float s = 0.f;
#pragma unroll 5
            for (int q = 0; q < NTAP; ++q) s += dl[p][q] * w2s[q][j];
            const float pre = hidden[(size_t)m * NH + j];
            dhidden[(size_t)m * NH + j] = pre > 0.f ? s : 0.01f * s;
            hl[p][j] = pre > 0.f ? pre : 0.01f * pre;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < APB_ITEMS; ++k) {
            const int i = tid + k * APB_NT;
            if (i < NTAP * NH) {
                const int q = i / NH, j = i % NH;
                float s = 0.f;
#pragma unroll 8
                for (int p = 0; p < PIX; ++p) s += dl[p][q] * hl[p][j];      // (dl of a pixel beyond M is 0: its attn was read as 0)
                w2acc[k] += s;
            }
        }
        if (tid < NTAP)
            for (int p = 0; p < PIX; ++p)
                if (m0 + p < M) b2acc += dl[p][tid];
    }
#pragma unroll
    for (int k = 0; k < APB_ITEMS; ++k) {
        const int i = tid + k * APB_NT;
        if (i < NTAP * NH) atomicAdd(&dw2[i], w2acc[k]);
    }
    if (tid < NTAP) atomicAdd(&db2[tid], b2acc);
}

// ---------------------------------------------------------------------------------------------- backward: gathers
// The two source-side gradients of the layer are transposes of gathers through the pixels' sampling frames:
//   dsource[cell] += sum_m kf(m)[cell - (P(m) - 2)] * dout[m]            (the weighted average; 6x6 footprint)
//   dGs[cell]      = sum_m w_ab(m) * dhidden[m],  cell = P(m) + (a,b)     (the bilinear read of Gs; 2x2 footprint)
// Round 1 scattered them with atomics (through LDS patches flushed with one global atomic per patch cell: the flush alone
// was 4x the tensor's size in atomics).  Here the pixels are first BUCKETED by their frame cell P(m), clamped to
// [-3, H+1] x [-3, W+1] (beyond that range every footprint cell clamps to the border anyway, so the clamped frame gives the
// same result) -- a counting sort: count, scan, fill; the flow of a resolution is shared by all its attention layers, so the
// host builds the index once per resolution and step -- and every output cell then GATHERS from the few buckets whose
// footprints reach it: no atomics, every cell written once, exact for any flow.
constexpr int BLO = 3;                       // bucket grid: P in [-BLO, H+1] -> (H + BLO + 2) rows
__device__ __forceinline__ int bucket_dim(int n) { return n + BLO + 2; }

__global__ void attn_bucket_count_kernel(const float *__restrict__ flow, int *__restrict__ counts, int *__restrict__ bucket_of,
                                         int B, int H, int W) {
    const int hw = H * W, M = B * hw;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int b = m / hw, rem = m - b * hw;
    const Frame f = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
    const int py = clampi(f.by, -BLO, H + 1) + BLO, px = clampi(f.bx, -BLO, W + 1) + BLO;
    const int bk = (b * bucket_dim(H) + py) * bucket_dim(W) + px;
    bucket_of[m] = bk;
    atomicAdd(&counts[bk], 1);
}

// exclusive scan of n ints in three small launches: per-1024 block scan, scan of the block totals (n <= 1M), offset add
__global__ __launch_bounds__(1024) void scan_block_kernel(const int *__restrict__ in, int *__restrict__ out,
                                                          int *__restrict__ totals, int n) {
    __shared__ int buf[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    if (i < n) out[i] = buf[threadIdx.x] - v;
    if (threadIdx.x == 1023 && totals) totals[blockIdx.x] = buf[1023];
}
__global__ void scan_add_kernel(int *__restrict__ out, const int *__restrict__ block_off, int n) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_off[blockIdx.x];
}
__global__ void attn_bucket_fill_kernel(const int *__restrict__ bucket_of, const int *__restrict__ offsets,
                                        int *__restrict__ cursor, int *__restrict__ items, int M, int bw) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int bk = bucket_of[m];
    items[offsets[bk] + atomicAdd(&cursor[bk], 1)] = m | ((bk % bw) << 20);        // pixel | bucket column
}

// footprint index range [lo, hi] of frame row `p` (cells p - 2 + i, i = 0..5, border-clamped) that lands on cell `y`
__device__ __forceinline__ void fp_range(int p, int y, int n, int span, int org, int &lo, int &hi) {
    // cells q = p - org + i, i in [0, span); clamp(q, 0, n-1) == y
    const int i0 = y - p + org;
    lo = (y == 0) ? 0 : i0;
    hi = (y == n - 1) ? span - 1 : i0;
    if (y == 0) hi = min(hi, i0);            // q <= 0
    if (y == n - 1) lo = max(lo, i0);        // q >= n-1
    if (y == 0 && y == n - 1) { lo = 0; hi = span - 1; }
    lo = max(lo, 0);
    hi = min(hi, span - 1);
}

// dsource[b][y][x][c4] += sum over the buckets whose 6x6 footprint reaches cell (y, x).
// The candidate frames of a cell are ALWAYS a 6 x 6 window of buckets (rows y-3 .. y+2; at the borders -3 .. 2 and
// H-4 .. H+1, where several footprint rows clamp onto the cell), and the buckets of one row are consecutive, so their
// pixels are ONE contiguous range of `items` (an item = pixel | bucket column << 20): per window row two offsets, then a
// short list (~6-9 pixels).  A workgroup = a 4 x 4 patch of cells x 16 channel vectors (256 B per pixel row; the patch's cells
// share most of their candidates, whose dout rows then come from L1).  The 16 lanes of a cell first build the cell's
// (pixel, weight) list in LDS TOGETHER -- the index arithmetic and the kf lookups are per (cell, pixel), not per channel --
// and then every lane walks the list: two LDS reads, one 16-B load and four FMAs per entry.
constexpr int SG_ROWCAP = 16;                 // list slots per window row (a row with more pixels takes the slow path)
__global__ __launch_bounds__(256) void attn_src_gather_kernel(const int *__restrict__ offsets, const int *__restrict__ items,
                                                              const float *__restrict__ kfbuf, const float *__restrict__ dout,
                                                              float *__restrict__ dsrc, int B, int H, int W, int C,
                                                              const float *__restrict__ init) {
    __shared__ int lm[16][FP * SG_ROWCAP];
    __shared__ float lw[16][FP * SG_ROWCAP];
    const int CV = C >> 2, CVC = CV / 16;
    const int bh = bucket_dim(H), bw = bucket_dim(W);
    const int tiles_x = (W + 3) >> 2, tiles_y = (H + 3) >> 2;
    int t = blockIdx.x;
    const int cc = t % CVC;
    t /= CVC;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int cell = threadIdx.x >> 4, l16 = threadIdx.x & 15, cv = cc * 16 + l16;
    const int y = ty * 4 + (cell >> 2), x = tx * 4 + (cell & 3);
    const bool live = y < H && x < W;
    const int py0 = y == 0 ? -BLO : (y == H - 1 ? H - 4 : y - 3);          // first of the six window rows / columns
    const int px0 = x == 0 ? -BLO : (x == W - 1 ? W - 4 : x - 3);
    int e0[FP], e1[FP];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < FP; ++r) {
        e0[r] = e1[r] = 0;
        if (live) {
            const int row = (b * bh + py0 + r + BLO) * bw + BLO;
            e0[r] = offsets[row + px0];
            e1[r] = offsets[row + px0 + FP];
        }
    }
    auto weight_of = [&](int item, int ilo, int ihi) -> float {
        const int m = item & 0xFFFFF, px = (item >> 20) - BLO;
        int jlo, jhi;
        fp_range(px, x, W, FP, KS / 2, jlo, jhi);
        float ws = 0.f;
        const float *kf = kfbuf + (size_t)m * NFP;
        for (int ii = ilo; ii <= ihi; ++ii)
            for (int jj = jlo; jj <= jhi; ++jj) ws += kf[ii * FP + jj];
        return ws;
    };
#pragma unroll
    for (int r = 0; r < FP; ++r) {
        int ilo, ihi;
        fp_range(py0 + r, y, H, FP, KS / 2, ilo, ihi);
        const int cnt = (live && ilo <= ihi) ? e1[r] - e0[r] : 0;
        // slots of this row: lane l fills slot l (SG_ROWCAP == 16 lanes)
        int mm = -1;
        float ww = 0.f;
        if (l16 < cnt) {
            const int item = items[e0[r] + l16];
            mm = item & 0xFFFFF;
            ww = weight_of(item, ilo, ihi);
        }
        lm[cell][r * SG_ROWCAP + l16] = mm;
        lw[cell][r * SG_ROWCAP + l16] = ww;
        // overflow (more than 16 pixels framed on one window row: strongly convergent flow): every lane walks the rest itself
        for (int e = e0[r] + SG_ROWCAP; e < e0[r] + cnt; ++e) {
            const int item = items[e];
            const float w = weight_of(item, ilo, ihi);
            const float4 g = *reinterpret_cast<const float4 *>(dout + (size_t)(item & 0xFFFFF) * C + cv * 4);
            acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
        }
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int r = 0; r < FP; ++r) {
        const int cnt = min(e1[r] - e0[r], SG_ROWCAP);
        for (int k0 = 0; k0 < cnt; k0 += 4) {
            float4 g[4];
            float w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int mm = k0 + k < cnt ? lm[cell][r * SG_ROWCAP + k0 + k] : -1;
                w[k] = mm >= 0 ? lw[cell][r * SG_ROWCAP + k0 + k] : 0.f;
                g[k] = mm >= 0 ? *reinterpret_cast<const float4 *>(dout + (size_t)mm * C + cv * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x += w[k] * g[k].x; acc.y += w[k] * g[k].y; acc.z += w[k] * g[k].z; acc.w += w[k] * g[k].w;
            }
        }
    }
    // every (pixel, channel) is written exactly once: dsource = init (another consumer's gradient; nothing if NULL) + the gather
    const size_t di = (((size_t)b * H + y) * W + x) * C + cv * 4;
    if (init) {
        const float4 o = *reinterpret_cast<const float4 *>(init + di);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *reinterpret_cast<float4 *>(dsrc + di) = acc;
}

// dGs[b][gy][gx][j4] = sum_m w_ab(m) * dhidden[m][j4] over the pixels whose frame corner (a,b) (clamped to [-2, H+1]) is the cell
__global__ __launch_bounds__(256) void attn_gs_gather_kernel(const int *__restrict__ offsets, const int *__restrict__ items,
                                                             const float *__restrict__ flow, const float *__restrict__ dhidden,
                                                             float *__restrict__ dgs, int B, int H, int W) {
    const int Hg = H + 2 * GPAD, Wg = W + 2 * GPAD, hw = H * W;
    const int64_t n = (int64_t)B * Hg * Wg * (NH / 4);
    const int bh = bucket_dim(H), bw = bucket_dim(W);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int jv = (int)(i % (NH / 4));
        int64_t t = i / (NH / 4);
        const int gx = (int)(t % Wg) - GPAD;
        t /= Wg;
        const int gy = (int)(t % Hg) - GPAD, b = (int)(t / Hg);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // frame rows p with clamp(p + a, -2, H+1) == gy for a in {0,1}: p in {gy-1, gy}, plus p = -3 for the first row
        for (int py = max(gy - 1, -BLO); py <= min(gy, H + 1); ++py)
            for (int px = max(gx - 1, -BLO); px <= min(gx, W + 1); ++px) {
                const int bk = (b * bh + py + BLO) * bw + px + BLO;
                const int e0 = offsets[bk], e1 = offsets[bk + 1];
                for (int e = e0; e < e1; ++e) {
                    const int m = items[e] & 0xFFFFF;
                    const int rem = m - b * hw;
                    const Frame f = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
                    float wy = 0.f, wx = 0.f;          // total weight of this pixel's corners that clamp onto the cell
                    if (clampi(f.by, -GPAD, H + GPAD - 1) == gy) wy += 1.f - f.wy1;
                    if (clampi(f.by + 1, -GPAD, H + GPAD - 1) == gy) wy += f.wy1;
                    if (clampi(f.bx, -GPAD, W + GPAD - 1) == gx) wx += 1.f - f.wx1;
                    if (clampi(f.bx + 1, -GPAD, W + GPAD - 1) == gx) wx += f.wx1;
                    const float w = wy * wx;
                    const float4 d = *reinterpret_cast<const float4 *>(dhidden + (size_t)m * NH + jv * 4);
                    acc.x += w * d.x; acc.y += w * d.y; acc.z += w * d.z; acc.w += w * d.w;
                }
            }
        *reinterpret_cast<float4 *>(dgs + (((size_t)b * Hg + gy + GPAD) * Wg + gx + GPAD) * NH + jv * 4) = acc;
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hoig_replicate_pad_fwd(const float *x, float *y, int B, int H, int W, int C, int pad, hoig_stream_t stream) {
    if (!x || !y || pad < 0 || (C & 3)) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * (H + 2 * pad) * (W + 2 * pad) * (C / 4);
    replicate_pad_fwd_kernel<<<hoig_stream_grid(n, 256), 256, 0, ST>>>(x, y, B, H, W, C, pad);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_replicate_pad_bwd_add(const float *dy, const float *addend, float *dx, int B, int H, int W, int C, int pad,
                                          hoig_stream_t stream) {
    if (!dy || !dx || pad < 0 || (C & 3)) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    replicate_pad_bwd_kernel<<<hoig_stream_grid(n, 256), 256, 0, ST>>>(dy, dx, B, H, W, C, pad, addend);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_replicate_pad_bwd(const float *dy, float *dx, int B, int H, int W, int C, int pad, hoig_stream_t stream) {
    return hoig_replicate_pad_bwd_add(dy, nullptr, dx, B, H, W, C, pad, stream);
}
// workspace of the pixel index: [counts / offsets: nb + 1][cursor: nb][bucket_of: M][items: M][block totals: 2048] ints
extern "C" int64_t hoig_attn_index_ints(int B, int H, int W) {
    const int64_t nb = (int64_t)B * (H + BLO + 2) * (W + BLO + 2);
    return 2 * (nb + 1) + 2 * (int64_t)B * H * W + 2048 + 8;
}
extern "C" int hoig_attn_build_index(const float *flow, int *index, int B, int H, int W, hoig_stream_t stream) {
    if (!flow || !index || B <= 0 || H <= 0 || W <= 0) return HOIG_EINVAL;
    const int64_t nb64 = (int64_t)B * (H + BLO + 2) * (W + BLO + 2);
    const int M = B * H * W;
    if (nb64 + 1 > 1024 * 1024 || M >= (1 << 20) || W + BLO + 2 >= 2048) return HOIG_EUNSUPPORTED;      // item = pixel | column << 20
    const int nb = (int)nb64, n = nb + 1;
    int *offsets = index, *cursor = index + n, *bucket_of = cursor + nb + 1, *items = bucket_of + M, *totals = items + M;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(index, 0, (size_t)(2 * n + 1) * sizeof(int), st) != hipSuccess) return HOIG_ELAUNCH;   // counts, cursor
    attn_bucket_count_kernel<<<(M + 255) / 256, 256, 0, st>>>(flow, offsets, bucket_of, B, H, W);
    const int nblk = (n + 1023) / 1024;
    scan_block_kernel<<<nblk, 1024, 0, st>>>(offsets, offsets, totals, n);
    scan_block_kernel<<<1, 1024, 0, st>>>(totals, totals + 1024, nullptr, nblk);
    scan_add_kernel<<<nblk, 1024, 0, st>>>(offsets, totals + 1024, n);
    attn_bucket_fill_kernel<<<(M + 255) / 256, 256, 0, st>>>(bucket_of, offsets, cursor, items, M, W + BLO + 2);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_src_gather(const int *index, const float *kf, const float *dout, const float *init, float *dsource,
                                    int B, int H, int W, int C, hoig_stream_t stream) {
    if (!index || !kf || !dout || !dsource || (C & 3) || B <= 0) return HOIG_EINVAL;
    const int n = B * (H + BLO + 2) * (W + BLO + 2) + 1, M = B * H * W;
    const int *items = index + 2 * n + M;
    if (C % 64) return HOIG_EUNSUPPORTED;
    const int64_t blocks = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 64);
    attn_src_gather_kernel<<<(unsigned)blocks, 256, 0, ST>>>(index, items, kf, dout, dsource, B, H, W, C, init);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_gs_gather(const int *index, const float *flow, const float *dhidden, float *dgs, int B, int H, int W,
                                   hoig_stream_t stream) {
    if (!index || !flow || !dhidden || !dgs || B <= 0) return HOIG_EINVAL;
    const int n = B * (H + BLO + 2) * (W + BLO + 2) + 1, M = B * H * W;
    const int *items = index + 2 * n + M;
    attn_gs_gather_kernel<<<hoig_stream_grid((int64_t)B * (H + 2 * GPAD) * (W + 2 * GPAD) * (NH / 4), 256), 256, 0, ST>>>(
        index, items, flow, dhidden, dgs, B, H, W);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_pixel_fwd(const float *gt, const float *gs, const float *flow, const float *w2, const float *b2,
                                   const float *source, float *hidden, float *attn, float *out, float *kf, int B, int H, int W,
                                   int C, hoig_stream_t stream) {
    if (!gt || !gs || !flow || !w2 || !b2 || !source || !hidden || !attn || !out || (C & 3) || B <= 0 || H <= 0 || W <= 0)
        return HOIG_EINVAL;
    const int M = B * H * W;
    attn_pixel_fwd_kernel<<<(M + 7) / 8, 256, 0, ST>>>(gt, gs, flow, w2, b2, source, hidden, attn, out, kf, B, H, W, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_pixel_bwd(const float *hidden, const float *attn, const float *w2, const float *source,
                                   const float *flow, const float *dout, float *dhidden, float *dw2, float *db2, float *e_ws,
                                   int B, int H, int W, int C, hoig_stream_t stream) {
    if (!hidden || !attn || !w2 || !source || !flow || !dout || !dhidden || !dw2 || !db2 || !e_ws || (C & 3) || B <= 0)
        return HOIG_EINVAL;
    const int M = B * H * W;
    if (C == 512) attn_edots_ch_kernel<64, 2><<<(M + 3) / 4, 256, 0, ST>>>(source, flow, dout, e_ws, B, H, W);
    else if (C == 256) attn_edots_ch_kernel<64, 1><<<(M + 3) / 4, 256, 0, ST>>>(source, flow, dout, e_ws, B, H, W);
    else if (C == 128) attn_edots_ch_kernel<32, 1><<<(M + 7) / 8, 256, 0, ST>>>(source, flow, dout, e_ws, B, H, W);
    else attn_edots_kernel<<<(M + 3) / 4, 256, 0, ST>>>(source, flow, dout, e_ws, B, H, W, C);
    const int groups = (M + APB_PIX - 1) / APB_PIX;
    // <= 256 workgroups: each closes with 3225 atomics into the SAME dW2 / db2 addresses, which retire at ~25 ns per address and
    // atomic (1024 workgroups on the 128x128 layer: a 25-us tail)
    const int nit = (groups + 255) / 256;
    attn_pixel_bwd_kernel<<<(groups + nit - 1) / nit, APB_NT, 0, ST>>>(hidden, attn, w2, e_ws, flow, dhidden, dw2, db2, B, H, W, C,
                                                                      nit);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
