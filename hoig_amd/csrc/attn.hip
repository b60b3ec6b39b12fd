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
                                         int p) {
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
        reinterpret_cast<float4 *>(dx)[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------- source gradient scatter
// dsource += bilinear^T( attn[m][q]/25 * dout[m][c] )   -- the source gradient of out[m] = (1/25) sum_q a_q S[m][q].
// The 25 taps of a pixel sample a regular 5x5 grid shifted by the pixel's flow, all with the SAME bilinear fractions, so
// their 100 (tap, corner) contributions collapse to a 6x6 footprint: out[i][j] = sum_{a,b in {0,1}} w_ab * v[i-a][j-b]
// (a 2x2 "full" correlation done in registers).
// LDS float atomics run at ~0.5 lane/clk/CU on gfx950 (measured: they were 80 % of this kernel), so the footprints are
// accumulated WITHOUT atomics: a workgroup is ONE wave that owns 32 channels of a TILE x TILE pixel tile and walks its
// pixels one after the other; lanes = 32 channels x 2 footprint halves (rows 0-2 / 3-5), i.e. the 64 lanes of one
// instruction always touch 64 distinct patch words, and LDS operations of one wave execute in order, so a plain
// read / add / write per cell is exact.  The patch is indexed by UNCLAMPED image coordinates (tile +- PRAD); K1's
// border clamp is applied when the patch is flushed to dsource with global atomics.  Pixels whose footprint leaves the
// patch (|flow| > ~PRAD-3) scatter straight to global memory.
constexpr int PCH = 32;
template <int TILE, int PRAD>
__global__ __launch_bounds__(64) void attn_sample_bwd_kernel(const float *__restrict__ flow,
                                                             const float *__restrict__ attn, const float *__restrict__ dout,
                                                             float *__restrict__ dsrc, int B, int H, int W, int C) {
    constexpr int PS = TILE + 2 * PRAD;
    __shared__ float patch[PS * PS * PCH];
    const int tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE;
    int t = blockIdx.x;
    const int bx = t % tiles_x;
    t /= tiles_x;
    const int by = t % tiles_y, b = t / tiles_y;
    const int c0 = blockIdx.y * PCH;
    const int hw = H * W;
    // patch origin (unclamped image coordinates), one cell further up-left than centred: the footprint of a pixel starts at
    // floor(flow) - 2, and off-hand pixels carry flow = -2 - identity in (-3, -1] (generator.py:484-488 on the -2 sentinel,
    // treated as PIXELS by K1) -- centred, the first row / column of every tile fell off the patch onto the atomic path
    constexpr int shift = 1;
    const int py0 = by * TILE - PRAD - shift, px0 = bx * TILE - PRAD - shift;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    for (int i = lane; i < PS * PS * PCH; i += 64) patch[i] = 0.f;
    __syncthreads();
    float *dimg = dsrc + (size_t)b * hw * C + c0 + c;
    const int rb = h ? 2 : 0;                                   // first tap row this half reads
    // (explicit software pipelining of the next pixels' loads -- 1 or 4 pixels ahead -- measured slower than letting five
    // one-wave workgroups per CU interleave)
    struct Pix {
        float v[3][KS];
        float fx, fy;
        int y, x;
    };
    auto load_pix = [&](int pl, Pix &P) {
        P.y = by * TILE + pl / TILE;
        P.x = bx * TILE + pl % TILE;
        if (pl >= TILE * TILE || P.y >= H || P.x >= W) {
            P.y = -1;
            return;
        }
        const int rem = P.y * W + P.x;
        const size_t m = (size_t)b * hw + rem;
        P.fx = flow[((size_t)b * 2 + 0) * hw + rem];
        P.fy = flow[((size_t)b * 2 + 1) * hw + rem];
        const float go = dout[m * C + c0 + c] * (1.f / NTAP);
        const float *am = attn + m * NTAP;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const int tap = (rb + k) * KS + q;
                P.v[k][q] = am[tap] * go;
            }
    };
    auto scatter = [&](const Pix &P) {
        if (P.y < 0) return;
        int by0, bx0;
        float wy1, wx1;
        k1_frame(P.fy, P.y, by0, wy1);
        k1_frame(P.fx, P.x, bx0, wx1);
        by0 -= KS / 2;                                              // tap (0,0)
        bx0 -= KS / 2;
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        // x pass: xr[k][j] = wx0 * v[k][j] + wx1 * v[k][j-1], j = 0..5
        float xr[3][KS + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xr[k][0] = wx0 * P.v[k][0];
#pragma unroll
            for (int j = 1; j < KS; ++j) xr[k][j] = wx0 * P.v[k][j] + wx1 * P.v[k][j - 1];
            xr[k][KS] = wx1 * P.v[k][KS - 1];
        }
        // y pass for this half's three footprint rows i = 3h + k:
        //   h = 0 (v rows 0,1,2): o_k = wy0 * xr[k]   + wy1 * xr[k-1]      h = 1 (v rows 2,3,4): o_k = wy0 * xr[k+1] + wy1 * xr[k]
        float o[3][KS + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j <= KS; ++j) {
                const float up = h ? (k < 2 ? xr[k + 1][j] : 0.f) : xr[k][j];
                const float dn = h ? xr[k][j] : (k > 0 ? xr[k - 1][j] : 0.f);
                o[k][j] = wy0 * up + wy1 * dn;
            }
        const int ly0 = by0 - py0, lx0 = bx0 - px0;
        if (ly0 >= 0 && ly0 + KS < PS && lx0 >= 0 && lx0 + KS < PS) {
            float *cell = patch + ((ly0 + 3 * h) * PS + lx0) * PCH + c;
            float old[3][KS + 1];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j <= KS; ++j) old[k][j] = cell[(k * PS + j) * PCH];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j <= KS; ++j) cell[(k * PS + j) * PCH] = old[k][j] + o[k][j];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int yy = max(min(by0 + 3 * h + k, H - 1), 0);
#pragma unroll
                for (int j = 0; j <= KS; ++j) {
                    const int xx = max(min(bx0 + j, W - 1), 0);
                    atomicAdd(dimg + ((size_t)yy * W + xx) * C, o[k][j]);
                }
            }
        }
    };
#pragma unroll 2
    for (int pl = 0; pl < TILE * TILE; ++pl) {
        Pix P;
        load_pix(pl, P);
        scatter(P);
    }
    __syncthreads();
    for (int cell = h; cell < PS * PS; cell += 2) {
        const float v = patch[cell * PCH + c];
        if (v != 0.f) {
            const int yy = max(min(py0 + cell / PS, H - 1), 0), xx = max(min(px0 + cell % PS, W - 1), 0);
            atomicAdd(dimg + ((size_t)yy * W + xx) * C, v);
        }
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
                                                             float *__restrict__ out, int B, int H, int W, int C) {
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

// da_q = (1/25) <dout[m], S[m][q]> = (1/25) sum_ab w_ab E[ty+a][tx+b], E[i][j] = <dout[m], source cell (i,j) of the footprint>;
// dlogit = a*(da - <a,da>); dW2 += dlogit (x) leaky(h); db2 += dlogit; dhidden = (W2^T dlogit) * leaky'(h).
// (The source gradient a_q/25*dout is attn_sample_bwd; the source gradient through `hidden` runs through dGs.)
// One 1024-thread workgroup per 32 pixels (x nit consecutive groups): dW2 / db2 are accumulated in registers over the
// workgroup's pixels and added to global memory ONCE -- with 8-pixel workgroups those 3225 fp32 atomics per workgroup, all
// workgroups on the same addresses, were a third of the kernel.
constexpr int APB_NT = 1024, APB_PIX = APB_NT / 32, APB_ITEMS = (NTAP * NH + APB_NT - 1) / APB_NT;
__global__ __launch_bounds__(APB_NT) void attn_pixel_bwd_kernel(const float *__restrict__ hidden, const float *__restrict__ attn,
                                                                const float *__restrict__ w2, const float *__restrict__ src,
                                                                const float *__restrict__ flow, const float *__restrict__ dout,
                                                                float *__restrict__ dhidden, float *__restrict__ dw2,
                                                                float *__restrict__ db2, int B, int H, int W, int C, int nit) {
    constexpr int PIX = APB_PIX;
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float dl[PIX][32];
    __shared__ float da[PIX][32];
    __shared__ float ef[PIX][NFP + 4];                   // footprint dot products E[i][j]
    __shared__ Frame fr[PIX];
    __shared__ float hl[PIX][NH];                        // leaky(hidden) of the group's pixels (read 25 times each for dW2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hw = H * W, M = B * hw;
    for (int i = tid; i < NTAP * NH; i += APB_NT) w2s[i / NH][i % NH] = w2[i];
    float w2acc[APB_ITEMS], b2acc = 0.f;
#pragma unroll
    for (int k = 0; k < APB_ITEMS; ++k) w2acc[k] = 0.f;
    const int CV = C >> 2;
    for (int it = 0; it < nit; ++it) {
        const int m0 = (blockIdx.x * nit + it) * PIX;
        if (m0 >= M) break;                              // (uniform)
        __syncthreads();                                 // w2s loaded / the previous group's dl, da consumed
        if (tid < PIX && m0 + tid < M) {
            const int m = m0 + tid, b = m / hw, rem = m - b * hw;
            fr[tid] = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
        }
        __syncthreads();
        // E: a wave takes two pixels; its lanes split into G groups of LPP lanes, a group reduces one footprint cell at a time
        for (int pp = 0; pp < 2; ++pp) {
            const int p = wave * 2 + pp, m = m0 + p;
            if (m >= M) continue;
            const int b = m / hw;
            const Frame f = fr[p];
            const int LPP = CV >= 64 ? 64 : (CV >= 32 ? 32 : 16), G = 64 / LPP, ts = lane / LPP, cl = lane % LPP;
            const float *sb = src + (size_t)b * hw * C;
            for (int c0 = 0; c0 < NFP; c0 += 6 * G) {
                float part[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int cell = c0 + u * G + ts;
                    float acc = 0.f;
                    if (cell < NFP) {
                        const int yy = clampi(f.by - KS / 2 + cell / FP, 0, H - 1), xx = clampi(f.bx - KS / 2 + cell % FP, 0, W - 1);
                        const float *sp = sb + ((size_t)yy * W + xx) * C;
                        for (int cv = cl; cv < CV; cv += LPP) {
                            const float4 go = *reinterpret_cast<const float4 *>(dout + (size_t)m * C + cv * 4);
                            const float4 sv = *reinterpret_cast<const float4 *>(sp + cv * 4);
                            acc += go.x * sv.x + go.y * sv.y + go.z * sv.z + go.w * sv.w;
                        }
                    }
                    part[u] = acc;
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    float v = part[u];
                    for (int o = LPP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    const int cell = c0 + u * G + ts;
                    if (cl == 0 && cell < NFP) ef[p][cell] = v;
                }
            }
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

// dGs[b][clamp(P + (a,b))][j] += w_ab(m) * dhidden[m][j]: the transpose of the bilinear read of Gs in attn_pixel_fwd.
// One thread per (pixel, 4 channels); fp32 atomics (fire-and-forget) into the caller-zeroed [B, H+4, W+4, 128] tensor.
__global__ __launch_bounds__(256) void attn_gs_scatter_kernel(const float *__restrict__ dhidden, const float *__restrict__ flow,
                                                              float *__restrict__ dgs, int B, int H, int W) {
    const int hw = H * W, Hg = H + 2 * GPAD, Wg = W + 2 * GPAD;
    const int64_t n = (int64_t)B * hw * (NH / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int jv = (int)(i % (NH / 4));
        const int m = (int)(i / (NH / 4));
        const int b = m / hw, rem = m - b * hw;
        const Frame f = pixel_frame(flow, b, rem, hw, rem / W, rem % W);
        const float4 d = *reinterpret_cast<const float4 *>(dhidden + (size_t)m * NH + jv * 4);
        const int ys[2] = {clampi(f.by, -GPAD, H + GPAD - 1) + GPAD, clampi(f.by + 1, -GPAD, H + GPAD - 1) + GPAD};
        const int xs[2] = {clampi(f.bx, -GPAD, W + GPAD - 1) + GPAD, clampi(f.bx + 1, -GPAD, W + GPAD - 1) + GPAD};
        const float wy[2] = {1.f - f.wy1, f.wy1}, wx[2] = {1.f - f.wx1, f.wx1};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float w = wx[bb] * wy[a];
                float *g = dgs + (((size_t)b * Hg + ys[a]) * Wg + xs[bb]) * NH + jv * 4;
                atomicAdd(g + 0, w * d.x);
                atomicAdd(g + 1, w * d.y);
                atomicAdd(g + 2, w * d.z);
                atomicAdd(g + 3, w * d.w);
            }
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
extern "C" int hoig_replicate_pad_bwd(const float *dy, float *dx, int B, int H, int W, int C, int pad, hoig_stream_t stream) {
    if (!dy || !dx || pad < 0 || (C & 3)) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    replicate_pad_bwd_kernel<<<hoig_stream_grid(n, 256), 256, 0, ST>>>(dy, dx, B, H, W, C, pad);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_sample_bwd(const float *flow, const float *attn, const float *dout, float *dsource, int B, int H,
                                    int W, int C, hoig_stream_t stream) {
    if (!flow || !attn || !dout || !dsource || (C % PCH)) return HOIG_EINVAL;
    // 8x8 tiles (32 KB patch, 4-5 one-wave workgroups per CU) unless that leaves too few pixels per flush: 16x16 tiles for
    // the large maps
    static const int force = getenv("HOIG_ASB_TILE") ? atoi(getenv("HOIG_ASB_TILE")) : 0;
    const bool big = force ? force == 16 : (int64_t)B * hoig_cdiv(H, 8) * hoig_cdiv(W, 8) * (C / PCH) > 8192;
    if (big) {
        const int tiles = B * (int)hoig_cdiv(H, 16) * (int)hoig_cdiv(W, 16);
        attn_sample_bwd_kernel<16, 4><<<dim3(tiles, C / PCH), 64, 0, ST>>>(flow, attn, dout, dsource, B, H, W, C);
    } else {
        const int tiles = B * (int)hoig_cdiv(H, 8) * (int)hoig_cdiv(W, 8);
        attn_sample_bwd_kernel<8, 4><<<dim3(tiles, C / PCH), 64, 0, ST>>>(flow, attn, dout, dsource, B, H, W, C);
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_pixel_fwd(const float *gt, const float *gs, const float *flow, const float *w2, const float *b2,
                                   const float *source, float *hidden, float *attn, float *out, int B, int H, int W, int C,
                                   hoig_stream_t stream) {
    if (!gt || !gs || !flow || !w2 || !b2 || !source || !hidden || !attn || !out || (C & 3) || B <= 0 || H <= 0 || W <= 0)
        return HOIG_EINVAL;
    const int M = B * H * W;
    attn_pixel_fwd_kernel<<<(M + 7) / 8, 256, 0, ST>>>(gt, gs, flow, w2, b2, source, hidden, attn, out, B, H, W, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_pixel_bwd(const float *hidden, const float *attn, const float *w2, const float *source,
                                   const float *flow, const float *dout, float *dhidden, float *dw2, float *db2, int B, int H,
                                   int W, int C, hoig_stream_t stream) {
    if (!hidden || !attn || !w2 || !source || !flow || !dout || !dhidden || !dw2 || !db2 || (C & 3) || B <= 0) return HOIG_EINVAL;
    const int M = B * H * W;
    const int groups = (M + APB_PIX - 1) / APB_PIX;
    const int nit = groups >= 2048 ? (groups / 1024 > 8 ? 8 : groups / 1024) : 1;      // ~1024 workgroups on the large maps
    attn_pixel_bwd_kernel<<<(groups + nit - 1) / nit, APB_NT, 0, ST>>>(hidden, attn, w2, source, flow, dout, dhidden, dw2, db2,
                                                                      B, H, W, C, nit);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_gs_scatter(const float *dhidden, const float *flow, float *dgs, int B, int H, int W,
                                    hoig_stream_t stream) {
    if (!dhidden || !flow || !dgs || B <= 0 || H <= 0 || W <= 0) return HOIG_EINVAL;
    attn_gs_scatter_kernel<<<hoig_stream_grid((int64_t)B * H * W * (NH / 4), 256), 256, 0, ST>>>(dhidden, flow, dgs, B, H, W);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
