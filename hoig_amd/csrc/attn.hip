// Local-attention feature warping (ExtractorAttn, extract_attn.py:23-29) as a composition around the MFMA convolution
// kernels.  Reference dataflow: K1 block-extracts source (with flow) and target (zero flow) into [B,C,5H,5W] tensors,
// concatenates, conv k5/s5 (2C->128) + LeakyReLU(0.01) + conv1x1 (128->25) + softmax, K3 reshape, multiply with the
// extracted source, average-pool 5x5.  Restated per output pixel m and tap q = ty*5+tx (offsets ty-2, tx-2,
// block_extractor_kernel.cu:57-60):
//     hidden[m] = b1 + sum_q W1t[q] . target[clamp(m + off_q)]  +  sum_q W1s[q] . S[m][q]
//     S[m][q]   = K1 bilinear sample of source at m + flow(m) + off_q  (border-clamped taps, un-renormalised weights)
//     a[m]      = softmax_25( W2 . leaky(hidden[m]) + b2 );     out[m] = (1/25) sum_q a[m][q] * S[m][q]
// so  * the target half is a 5x5 convolution of the REPLICATE-padded target     -> hoig_replicate_pad + conv kernels
//     * the source half is a 1x1 convolution over the sampled tensor S [M][25*C] -> hoig_attn_sample + conv kernels
// (both GEMMs, their data and weight gradients run on the tuned implicit-GEMM kernels, bf16x3 or fp32), and only the
// per-pixel softmax / weighted average and the bilinear gather / scatter live here.  S is the one 25x-sized tensor that
// is materialised (the reference materialises three, all fp32, plus their gradients).
// The source gradient is a scatter through the bilinear taps (K2, block_extractor_kernel.cu:158-161): sample positions
// stay within a few pixels of m (the flow is a normalised-coordinate difference read as pixels, generator.py:484-488),
// so each workgroup accumulates its tile's contributions in an LDS patch with ds_add_f32 and flushes the patch with one
// global atomic per patch cell; taps that fall outside the patch take the global-atomic path directly.
#include "common.h"

namespace {

constexpr int KS = 5, NTAP = 25, NH = 128;

struct Corner {
    int y0, y1, x0, x1;          // clamped tap coordinates
    float w00, w01, w10, w11;    // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
};

// K1 sampling taps for pixel (y,x), flow (fx,fy) in pixel units, tap q (block_extractor_kernel.cu:57-76)
__device__ __forceinline__ Corner k1_corner(int y, int x, float fx, float fy, int q, int H, int W) {
    const int oy = q / KS - KS / 2, ox = q % KS - KS / 2;
    const float flow_y = fy + oy, flow_x = fx + ox;
    const float dy = flow_y + (float)y, dx = flow_x + (float)x;
    const float fly = floorf(dy), flx = floorf(dx);
    Corner c;
    c.x0 = max(min((int)flx, W - 1), 0);
    c.x1 = max(min((int)flx + 1, W - 1), 0);
    c.y0 = max(min((int)fly, H - 1), 0);
    c.y1 = max(min((int)fly + 1, H - 1), 0);
    const float xR_P = dx - flx, xL_P = 1.f - xR_P, yB_P = dy - fly, yT_P = 1.f - yB_P;
    c.w00 = xL_P * yT_P; c.w01 = xR_P * yT_P; c.w10 = xL_P * yB_P; c.w11 = xR_P * yB_P;
    return c;
}

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

// ---------------------------------------------------------------------------------------------- source sampling
// S[m][q][c]: one thread per (m, 4 channels) walks the 25 taps.  K1's sampling coordinate is separable -- dy depends
// only on (flow_y, tap row, y), dx only on (flow_x, tap column, x) (block_extractor_kernel.cu:62-76) -- so the five row
// and five column coordinate sets are computed once per pixel with exactly K1's arithmetic and shared by the 25 taps.
struct Axis {
    int i0[KS], i1[KS];
    float w0[KS], w1[KS];
};
__device__ __forceinline__ void k1_axis(Axis &a, float f, int pos, int lim) {
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        const float fl_ = f + (float)(t - KS / 2);          // flow + offset   (:62-63)
        const float d = fl_ + (float)pos;                   // + pixel index   (:66-67)
        const float fl = floorf(d);
        a.i0[t] = max(min((int)fl, lim - 1), 0);
        a.i1[t] = max(min((int)fl + 1, lim - 1), 0);
        a.w1[t] = d - fl;
        a.w0[t] = 1.f - a.w1[t];
    }
}
__global__ __launch_bounds__(256) void attn_sample_fwd_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                                              float *__restrict__ S, int B, int H, int W, int C) {
    const int CV = C >> 2, hw = H * W;
    const int64_t n = (int64_t)B * hw * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const int64_t m = i / CV;
        const int b = (int)(m / hw), rem = (int)(m - (int64_t)b * hw);
        const int y = rem / W, x = rem - y * W;
        Axis ay, ax;
        k1_axis(ax, flow[((size_t)b * 2 + 0) * hw + rem], x, W);
        k1_axis(ay, flow[((size_t)b * 2 + 1) * hw + rem], y, H);
        const float *s = src + (size_t)b * hw * C + cv * 4;
        float *o = S + (size_t)m * NTAP * C + cv * 4;
        // The 25 taps sample a 5x5 grid with a common shift: their 100 corner reads hit a 6x6 footprint whenever every
        // tap's right / lower neighbour is the next tap's left / upper one (always, up to a rounding coincidence of the
        // per-tap floor at an exact integer coordinate).  Then 36 loads serve all taps, with K1's arithmetic unchanged.
        bool chained = true;
#pragma unroll
        for (int t = 0; t + 1 < KS; ++t) chained = chained && ax.i1[t] == ax.i0[t + 1] && ay.i1[t] == ay.i0[t + 1];
        if (chained) {
            float4 R0[KS + 1], R1[KS + 1];
            const float *row = s + (size_t)ay.i0[0] * W * C;
#pragma unroll
            for (int t = 0; t <= KS; ++t) R0[t] = *reinterpret_cast<const float4 *>(row + (size_t)(t < KS ? ax.i0[t] : ax.i1[KS - 1]) * C);
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                row = s + (size_t)ay.i1[r] * W * C;
#pragma unroll
                for (int t = 0; t <= KS; ++t)
                    R1[t] = *reinterpret_cast<const float4 *>(row + (size_t)(t < KS ? ax.i0[t] : ax.i1[KS - 1]) * C);
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    const float4 a = R0[t], bb = R0[t + 1], d = R1[t], e = R1[t + 1];
                    const float w00 = ax.w0[t] * ay.w0[r], w01 = ax.w1[t] * ay.w0[r];
                    const float w10 = ax.w0[t] * ay.w1[r], w11 = ax.w1[t] * ay.w1[r];
                    float4 v;
                    v.x = w00 * a.x; v.x += w01 * bb.x; v.x += w10 * d.x; v.x += w11 * e.x;
                    v.y = w00 * a.y; v.y += w01 * bb.y; v.y += w10 * d.y; v.y += w11 * e.y;
                    v.z = w00 * a.z; v.z += w01 * bb.z; v.z += w10 * d.z; v.z += w11 * e.z;
                    v.w = w00 * a.w; v.w += w01 * bb.w; v.w += w10 * d.w; v.w += w11 * e.w;
                    *reinterpret_cast<float4 *>(o + (size_t)(r * KS + t) * C) = v;
                }
#pragma unroll
                for (int t = 0; t <= KS; ++t) R0[t] = R1[t];
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const float *row0 = s + (size_t)ay.i0[r] * W * C, *row1 = s + (size_t)ay.i1[r] * W * C;
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const float4 a = *reinterpret_cast<const float4 *>(row0 + (size_t)ax.i0[t] * C);
                const float4 bb = *reinterpret_cast<const float4 *>(row0 + (size_t)ax.i1[t] * C);
                const float4 d = *reinterpret_cast<const float4 *>(row1 + (size_t)ax.i0[t] * C);
                const float4 e = *reinterpret_cast<const float4 *>(row1 + (size_t)ax.i1[t] * C);
                const float w00 = ax.w0[t] * ay.w0[r], w01 = ax.w1[t] * ay.w0[r];     // xL_P*yT_P, xR_P*yT_P (:79-82)
                const float w10 = ax.w0[t] * ay.w1[r], w11 = ax.w1[t] * ay.w1[r];
                float4 v;
                v.x = w00 * a.x; v.x += w01 * bb.x; v.x += w10 * d.x; v.x += w11 * e.x;
                v.y = w00 * a.y; v.y += w01 * bb.y; v.y += w10 * d.y; v.y += w11 * e.y;
                v.z = w00 * a.z; v.z += w01 * bb.z; v.z += w10 * d.z; v.z += w11 * e.z;
                v.w = w00 * a.w; v.w += w01 * bb.w; v.w += w10 * d.w; v.w += w11 * e.w;
                *reinterpret_cast<float4 *>(o + (size_t)(r * KS + t) * C) = v;
            }
        }
    }
}

// dsource += bilinear^T( dS[m][q][c] + attn[m][q]/25 * dout[m][c] ).
// The 25 taps of a pixel sample a regular 5x5 grid shifted by the pixel's flow, all with the SAME bilinear fractions, so
// their 100 (tap, corner) contributions collapse to a 6x6 footprint: out[i][j] = sum_{a,b in {0,1}} w_ab * v[i-a][j-b]
// (a 2x2 "full" correlation done in registers).  (Per-tap K1 arithmetic can differ from the shared fraction by one ulp of
// the sampling coordinate; for this gradient scatter that is a ~1e-7 relative effect.)
// LDS float atomics run at ~0.5 lane/clk/CU on gfx950 (measured: they were 80 % of this kernel), so the footprints are
// accumulated WITHOUT atomics: a workgroup is ONE wave that owns 32 channels of a TILE x TILE pixel tile and walks its
// pixels one after the other; lanes = 32 channels x 2 footprint halves (rows 0-2 / 3-5), i.e. the 64 lanes of one
// instruction always touch 64 distinct patch words, and LDS operations of one wave execute in order, so a plain
// read / add / write per cell is exact.  The patch is indexed by UNCLAMPED image coordinates (tile +- PRAD); K1's
// border clamp is applied when the patch is flushed to dsource with global atomics.  Pixels whose footprint leaves the
// patch (|flow| > ~PRAD-3) scatter straight to global memory.
constexpr int PCH = 32;
template <int TILE, int PRAD>
__global__ __launch_bounds__(64) void attn_sample_bwd_kernel(const float *__restrict__ flow, const float *__restrict__ dS,
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
    const int rb = h ? 2 : 0;                                   // first dS tap row this half reads
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
        const float *dSm = dS ? dS + m * NTAP * C + c0 + c : nullptr;
        const float *am = attn + m * NTAP;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const int tap = (rb + k) * KS + q;
                P.v[k][q] = (dSm ? dSm[(size_t)tap * C] : 0.f) + am[tap] * go;
            }
    };
    auto scatter = [&](const Pix &P) {
        if (P.y < 0) return;
        const float dyc = (P.fy + (float)(-KS / 2)) + (float)P.y, dxc = (P.fx + (float)(-KS / 2)) + (float)P.x;   // tap (0,0)
        const float fly = floorf(dyc), flx = floorf(dxc);
        const float wy1 = dyc - fly, wy0 = 1.f - wy1, wx1 = dxc - flx, wx0 = 1.f - wx1;
        const int by0 = (int)fly, bx0 = (int)flx;
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
// 8 pixels per workgroup: 32 lanes per pixel compute the 25 logits + softmax, then out = (1/25) sum_q a_q S[m][q]
__global__ __launch_bounds__(256) void attn_pixel_fwd_kernel(const float *__restrict__ hidden, const float *__restrict__ w2,
                                                             const float *__restrict__ b2, const float *__restrict__ S,
                                                             float *__restrict__ attn, float *__restrict__ out, int M,
                                                             int C) {
    constexpr int PIX = 8;
    __shared__ float hs[PIX][NH];
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float as[PIX][32];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * PIX;
    for (int i = tid; i < NTAP * NH; i += 256) w2s[i / NH][i % NH] = w2[i];
    for (int i = tid; i < PIX * NH; i += 256) {
        const int p = i / NH, j = i % NH, m = m0 + p;
        const float v = m < M ? hidden[(size_t)m * NH + j] : 0.f;
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
        as[p][q] = a * (1.f / NTAP);                   // avg_pool2d(5,5) of the product (extract_attn.py:28)
        if (q < NTAP && m < M) attn[(size_t)m * NTAP + q] = a;
    }
    __syncthreads();
    const int CV = C >> 2;
    for (int i = tid; i < PIX * CV; i += 256) {
        const int p = i / CV, cv = i - p * CV, m = m0 + p;
        if (m >= M) continue;
        const float *s = S + (size_t)m * NTAP * C + cv * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 5
        for (int q = 0; q < NTAP; ++q) {
            const float w = as[p][q];
            const float4 v = *reinterpret_cast<const float4 *>(s + (size_t)q * C);
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        *reinterpret_cast<float4 *>(out + (size_t)m * C + cv * 4) = acc;
    }
}

// da_q = (1/25) <dout[m], S[m][q]>; dlogit = a*(da - <a,da>); dW2 += dlogit (x) leaky(h); db2 += dlogit;
// dhidden = (W2^T dlogit) * leaky'(h).   (The source gradient a_q/25*dout is folded into attn_sample_bwd.)
// One 1024-thread workgroup per 32 pixels (x nit consecutive groups): dW2 / db2 are accumulated in registers over the
// workgroup's pixels and added to global memory ONCE -- with 8-pixel workgroups those 3225 fp32 atomics per workgroup, all
// workgroups on the same addresses, were a third of the kernel (106 vs 170 us on the 32x32 layers with them switched off).
constexpr int APB_NT = 1024, APB_PIX = APB_NT / 32, APB_ITEMS = (NTAP * NH + APB_NT - 1) / APB_NT;
__global__ __launch_bounds__(APB_NT) void attn_pixel_bwd_kernel(const float *__restrict__ hidden, const float *__restrict__ attn,
                                                                const float *__restrict__ w2, const float *__restrict__ S,
                                                                const float *__restrict__ dout, float *__restrict__ dhidden,
                                                                float *__restrict__ dw2, float *__restrict__ db2, int M,
                                                                int C, int nit) {
    constexpr int PIX = APB_PIX;
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float dl[PIX][32];
    __shared__ float da[PIX][32];
    __shared__ float hl[PIX][NH];                        // leaky(hidden) of the group's pixels (read 25 times each for dW2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NTAP * NH; i += APB_NT) w2s[i / NH][i % NH] = w2[i];
    float w2acc[APB_ITEMS], b2acc = 0.f;
#pragma unroll
    for (int k = 0; k < APB_ITEMS; ++k) w2acc[k] = 0.f;
    const int CV = C >> 2;
    // The S rows (25 x C floats per pixel) are the kernel's HBM stream.  Fast form for C = 128, 256, 512: dout stays in
    // registers, five taps' loads are in flight before their reductions, and with C = 128 the two wave halves take two taps.
    const bool fast = C == 128 || C == 256 || C == 512;
    for (int it = 0; it < nit; ++it) {
        const int m0 = (blockIdx.x * nit + it) * PIX;
        if (m0 >= M) break;                              // (uniform)
        __syncthreads();                                 // w2s loaded / the previous group's dl, da consumed
        da[tid >> 5][tid & 31] = 0.f;
        __syncthreads();
        for (int pp = 0; fast && pp < 2; ++pp) {
            const int p = wave * 2 + pp, m = m0 + p;
            if (m >= M) continue;
            const int LPP = CV >= 64 ? 64 : 32, G = 64 / LPP, ts = lane / LPP, cl = lane % LPP;
            const int nv = CV / LPP;                     // 1 or 2
            float4 go[2];
            go[0] = *reinterpret_cast<const float4 *>(dout + (size_t)m * C + cl * 4);
            go[1] = nv > 1 ? *reinterpret_cast<const float4 *>(dout + (size_t)m * C + (cl + LPP) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float *Sm = S + (size_t)m * NTAP * C + cl * 4;
            for (int q0 = 0; q0 < NTAP; q0 += 5 * G) {
                float part[5];
                float4 sv[5][2];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int q = q0 + u * G + ts;
                    const bool ok = q < NTAP;
                    sv[u][0] = ok ? *reinterpret_cast<const float4 *>(Sm + (size_t)q * C) : make_float4(0.f, 0.f, 0.f, 0.f);
                    sv[u][1] = (ok && nv > 1) ? *reinterpret_cast<const float4 *>(Sm + (size_t)q * C + LPP * 4)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    part[u] = go[0].x * sv[u][0].x + go[0].y * sv[u][0].y + go[0].z * sv[u][0].z + go[0].w * sv[u][0].w;
                    part[u] += go[1].x * sv[u][1].x + go[1].y * sv[u][1].y + go[1].z * sv[u][1].z + go[1].w * sv[u][1].w;
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    float v = part[u];
                    for (int o = LPP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    const int q = q0 + u * G + ts;
                    if (cl == 0 && q < NTAP) da[p][q] = v * (1.f / NTAP);
                }
            }
        }
        for (int pp = 0; !fast && pp < 2; ++pp) {       // any other C: lanes along channels, 25 reductions per pixel
            const int p = wave * 2 + pp, m = m0 + p;
            if (m >= M) continue;
            for (int q = 0; q < NTAP; ++q) {
                float part = 0.f;
                for (int cv = lane; cv < CV; cv += 64) {
                    const float4 go = *reinterpret_cast<const float4 *>(dout + (size_t)m * C + cv * 4);
                    const float4 sv = *reinterpret_cast<const float4 *>(S + ((size_t)m * NTAP + q) * C + cv * 4);
                    part += go.x * sv.x + go.y * sv.y + go.z * sv.z + go.w * sv.w;
                }
                part = hoig_wave_sum(part);
                if (lane == 0) da[p][q] = part * (1.f / NTAP);
            }
        }
        __syncthreads();
        {
            const int p = tid >> 5, q = tid & 31, m = m0 + p;
            const float a = (q < NTAP && m < M) ? attn[(size_t)m * NTAP + q] : 0.f;
            float dot = a * da[p][q];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 32);
            dl[p][q] = a * (da[p][q] - dot);
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
extern "C" int hoig_attn_sample_fwd(const float *source, const float *flow, float *sampled, int B, int H, int W, int C,
                                    hoig_stream_t stream) {
    if (!source || !flow || !sampled || (C & 3)) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    attn_sample_fwd_kernel<<<(unsigned)hoig_cdiv(n, 256) > 65535u * 16u ? 65535u * 16u : (unsigned)hoig_cdiv(n, 256), 256, 0, ST>>>(source, flow, sampled, B, H, W, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_sample_bwd(const float *flow, const float *dsampled, const float *attn, const float *dout,
                                    float *dsource, int B, int H, int W, int C, hoig_stream_t stream) {
    if (!flow || !attn || !dout || !dsource || (C % PCH)) return HOIG_EINVAL;
    // 8x8 tiles (32 KB patch, 4-5 one-wave workgroups per CU) unless that leaves too few pixels per flush: 16x16 tiles for
    // the large maps
    static const int force = getenv("HOIG_ASB_TILE") ? atoi(getenv("HOIG_ASB_TILE")) : 0;
    const bool big = force ? force == 16 : (int64_t)B * hoig_cdiv(H, 8) * hoig_cdiv(W, 8) * (C / PCH) > 8192;
    if (big) {
        const int tiles = B * (int)hoig_cdiv(H, 16) * (int)hoig_cdiv(W, 16);
        attn_sample_bwd_kernel<16, 4><<<dim3(tiles, C / PCH), 64, 0, ST>>>(flow, dsampled, attn, dout, dsource, B, H, W, C);
    } else {
        const int tiles = B * (int)hoig_cdiv(H, 8) * (int)hoig_cdiv(W, 8);
        attn_sample_bwd_kernel<8, 4><<<dim3(tiles, C / PCH), 64, 0, ST>>>(flow, dsampled, attn, dout, dsource, B, H, W, C);
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_pixel_fwd(const float *hidden, const float *w2, const float *b2, const float *sampled, float *attn,
                                   float *out, int M, int C, hoig_stream_t stream) {
    if (!hidden || !w2 || !b2 || !sampled || !attn || !out || (C & 3)) return HOIG_EINVAL;
    attn_pixel_fwd_kernel<<<(M + 7) / 8, 256, 0, ST>>>(hidden, w2, b2, sampled, attn, out, M, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_pixel_bwd(const float *hidden, const float *attn, const float *w2, const float *sampled,
                                   const float *dout, float *dhidden, float *dw2, float *db2, int M, int C,
                                   hoig_stream_t stream) {
    if (!hidden || !attn || !w2 || !sampled || !dout || !dhidden || !dw2 || !db2 || (C & 3)) return HOIG_EINVAL;
    const int groups = (M + APB_PIX - 1) / APB_PIX;
    const int nit = groups >= 2048 ? (groups / 1024 > 8 ? 8 : groups / 1024) : 1;      // ~1024 workgroups on the large maps
    attn_pixel_bwd_kernel<<<(groups + nit - 1) / nit, APB_NT, 0, ST>>>(hidden, attn, w2, sampled, dout, dhidden, dw2, db2, M, C, nit);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
