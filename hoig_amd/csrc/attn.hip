// Fused local-attention feature warping (ExtractorAttn, extract_attn.py:23-29) for gfx950.
//
// Reference dataflow (per layer): K1 block-extracts source (with flow) and target (zero flow) into two
// [B,C,5H,5W] tensors, concatenates them, runs conv k5/s5 (2C->128) + LeakyReLU(0.01) + conv1x1 (128->25) +
// softmax, K3-reshapes the 25 weights back to 5Hx5W, multiplies with the extracted source and average-pools 5x5.
// Here nothing 25x-sized is materialised:
//   fc1      : hidden[m][128] = b1 + sum_{tap,ch} A[m][(tap,ch)] * W1[n][tap][ch]   (implicit GEMM, MFMA; A is
//              gathered on the fly: ch<C -> target at the border-clamped neighbour, ch>=C -> K1's bilinear sample of
//              source at p + flow(p) + tap offset, block_extractor_kernel.cu:57-84)
//   pixel    : logits = W2 * leaky(hidden) + b2 ; a = softmax_25 ; out[m][c] = (1/25) sum_q a_q * s_q[c]
// Backward mirrors it: pixel-bwd (d a, softmax, dW2/db2, d hidden, source-gradient of the weighted average),
// fc1-wgrad (MFMA, reduction over pixels) and fc1-dgrad (MFMA, scattered into dtarget / dsource with fp32 atomics,
// exactly what K2 does at block_extractor_kernel.cu:158-161).
// Pixel index m = (b*H + y)*W + x; taps q = ty*5 + tx with offsets (ty-2, tx-2) (y%k - k/2, :59-60).
#include "common.h"

namespace {

constexpr int KS = 5, NTAP = 25, NH = 128;

struct AttnGeom {
    int B, H, W, C;
};

struct Corner {
    int i00, i01, i10, i11;   // pixel offsets (y*W + x) of the four clamped taps
    float w00, w01, w10, w11; // un-renormalised bilinear weights
};

// K1 sampling position for pixel (y,x), flow (fx,fy) in pixel units and tap q
__device__ __forceinline__ Corner k1_corner(int y, int x, float fx, float fy, int q, int H, int W) {
    const int oy = q / KS - KS / 2, ox = q % KS - KS / 2;
    const float flow_y = fy + oy, flow_x = fx + ox;
    const float dy = flow_y + (float)y, dx = flow_x + (float)x;
    const float fly = floorf(dy), flx = floorf(dx);
    const int xL = max(min((int)flx, W - 1), 0), xR = max(min((int)flx + 1, W - 1), 0);
    const int yT = max(min((int)fly, H - 1), 0), yB = max(min((int)fly + 1, H - 1), 0);
    const float xR_P = dx - flx, xL_P = 1.f - xR_P, yB_P = dy - fly, yT_P = 1.f - yB_P;
    Corner c;
    c.i00 = yT * W + xL; c.i01 = yT * W + xR; c.i10 = yB * W + xL; c.i11 = yB * W + xR;
    c.w00 = xL_P * yT_P; c.w01 = xR_P * yT_P; c.w10 = xL_P * yB_P; c.w11 = xR_P * yB_P;
    return c;
}

__device__ __forceinline__ int target_index(int y, int x, int q, int H, int W) {
    // zero flow: floor(dy) = y+oy exactly, the second tap has weight 0 (block_extractor_kernel.cu:69-76)
    const int oy = q / KS - KS / 2, ox = q % KS - KS / 2;
    const int yy = max(min(y + oy, H - 1), 0), xx = max(min(x + ox, W - 1), 0);
    return yy * W + xx;
}

struct Row {
    int b, y, x;
    float fx, fy;
};

__device__ __forceinline__ Row load_row(const AttnGeom &g, const float *__restrict__ flow, int m) {
    Row r;
    const int hw = g.H * g.W;
    r.b = m / hw;
    const int rem = m - r.b * hw;
    r.y = rem / g.W;
    r.x = rem - r.y * g.W;
    r.fx = flow[((size_t)r.b * 2 + 0) * hw + rem];
    r.fy = flow[((size_t)r.b * 2 + 1) * hw + rem];
    return r;
}

// A[m][k..k+3] of the virtual [M][25*2C] matrix
__device__ __forceinline__ float4 attn_gather(const AttnGeom &g, const float *__restrict__ src,
                                              const float *__restrict__ tgt, const Row &r, int k) {
    const int C2 = 2 * g.C;
    const int q = k / C2, ch = k - q * C2;
    const size_t img = (size_t)r.b * g.H * g.W;
    if (ch < g.C) {
        const int idx = target_index(r.y, r.x, q, g.H, g.W);
        return *reinterpret_cast<const float4 *>(tgt + (img + idx) * g.C + ch);
    }
    const Corner c = k1_corner(r.y, r.x, r.fx, r.fy, q, g.H, g.W);
    const float *s = src + img * g.C + (ch - g.C);
    const float4 a = *reinterpret_cast<const float4 *>(s + (size_t)c.i00 * g.C);
    const float4 b = *reinterpret_cast<const float4 *>(s + (size_t)c.i01 * g.C);
    const float4 d = *reinterpret_cast<const float4 *>(s + (size_t)c.i10 * g.C);
    const float4 e = *reinterpret_cast<const float4 *>(s + (size_t)c.i11 * g.C);
    float4 v;
    v.x = c.w00 * a.x; v.x += c.w01 * b.x; v.x += c.w10 * d.x; v.x += c.w11 * e.x;
    v.y = c.w00 * a.y; v.y += c.w01 * b.y; v.y += c.w10 * d.y; v.y += c.w11 * e.y;
    v.z = c.w00 * a.z; v.z += c.w01 * b.z; v.z += c.w10 * d.z; v.z += c.w11 * e.z;
    v.w = c.w00 * a.w; v.w += c.w01 * b.w; v.w += c.w10 * d.w; v.w += c.w11 * e.w;
    return v;
}

// ------------------------------------------------------------------------------------------ fc1 forward
// hidden[m][n] = b1[n] + sum_k A[m][k] W1[n][k];   tile 32 pixels x 128 hidden, 4 waves along n
__global__ __launch_bounds__(256) void attn_fc1_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                       const float *__restrict__ flow, const float *__restrict__ w1,
                                                       const float *__restrict__ b1, float *__restrict__ hidden,
                                                       AttnGeom g, int M, int K) {
    constexpr int BM = 32, BN = 128, BK = 32, LDA = BM + 1, LDB = BN + 1;
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int kc = (tid & 7) * 4, lrow = tid >> 3;
    const int m = m0 + lrow;
    Row row = load_row(g, flow, m < M ? m : M - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra, rb[4];
    const int nkb = K / BK;
    auto load_tiles = [&](int kb) {
        const int k = kb * BK + kc;
        ra = (m < M) ? attn_gather(g, src, tgt, row, k) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const float4 *>(w1 + (size_t)(lrow + 32 * i) * K + k);
    };
    auto store_tiles = [&]() {
        As[(kc + 0) * LDA + lrow] = ra.x; As[(kc + 1) * LDA + lrow] = ra.y;
        As[(kc + 2) * LDA + lrow] = ra.z; As[(kc + 3) * LDA + lrow] = ra.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i;
            Bs[(kc + 0) * LDB + r] = rb[i].x; Bs[(kc + 1) * LDB + r] = rb[i].y;
            Bs[(kc + 2) * LDB + r] = rb[i].z; Bs[(kc + 3) * LDB + r] = rb[i].w;
        }
    };
    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        if (kb + 1 < nkb) load_tiles(kb + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const float a = As[(2 * ks + lh) * LDA + l31];
            const float b = Bs[(2 * ks + lh) * LDB + wave * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
        if (kb + 1 < nkb) store_tiles();
        __syncthreads();
    }
    const int n = wave * 32 + l31;
    const float bias = b1[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mm < M) hidden[(size_t)mm * NH + n] = acc[r] + bias;
    }
}

// ------------------------------------------------------------------------------------------ pixel forward
// 8 pixels per workgroup.  Phase A: 32 lanes per pixel compute the 25 logits + softmax.  Phase B: weighted average.
__global__ __launch_bounds__(256) void attn_pixel_fwd_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                                             const float *__restrict__ hidden,
                                                             const float *__restrict__ w2, const float *__restrict__ b2,
                                                             float *__restrict__ attn, float *__restrict__ out,
                                                             AttnGeom g, int M) {
    constexpr int PIX = 8;
    __shared__ float hs[PIX][NH];
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float cw[PIX][NTAP][4];
    __shared__ int ci[PIX][NTAP][4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * PIX;
    for (int i = tid; i < NTAP * NH; i += 256) w2s[i / NH][i % NH] = w2[i];
    for (int i = tid; i < PIX * NH; i += 256) {
        const int p = i / NH, j = i % NH, m = m0 + p;
        float v = m < M ? hidden[(size_t)m * NH + j] : 0.f;
        hs[p][j] = v > 0.f ? v : 0.01f * v;   // LeakyReLU(0.01), extract_attn.py:19 / generator.py:344
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
        float e = q < NTAP ? expf(logit - mx) : 0.f;
        float sum = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 32);
        const float a = e / sum;
        if (q < NTAP && m < M) {
            attn[(size_t)m * NTAP + q] = a;
            const Row r = load_row(g, flow, m);
            const Corner c = k1_corner(r.y, r.x, r.fx, r.fy, q, g.H, g.W);
            const float s25 = a * (1.f / NTAP);   // avg_pool2d(5,5) of the product, extract_attn.py:28
            cw[p][q][0] = c.w00 * s25; cw[p][q][1] = c.w01 * s25; cw[p][q][2] = c.w10 * s25; cw[p][q][3] = c.w11 * s25;
            ci[p][q][0] = c.i00; ci[p][q][1] = c.i01; ci[p][q][2] = c.i10; ci[p][q][3] = c.i11;
        }
    }
    __syncthreads();
    const int CV = g.C >> 2;
    for (int i = tid; i < PIX * CV; i += 256) {
        const int p = i / CV, cv = i - p * CV, m = m0 + p;
        if (m >= M) continue;
        const int b = m / (g.H * g.W);
        const float *s = src + (size_t)b * g.H * g.W * g.C + cv * 4;
        float4 acc = make_float4(0, 0, 0, 0);
        for (int q = 0; q < NTAP; ++q) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float w = cw[p][q][t];
                const float4 v = *reinterpret_cast<const float4 *>(s + (size_t)ci[p][q][t] * g.C);
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        *reinterpret_cast<float4 *>(out + (size_t)m * g.C + cv * 4) = acc;
    }
}

// ------------------------------------------------------------------------------------------ pixel backward
// per pixel: da_q = (1/25) <dout, s_q>; dlogit = a*(da - <a,da>); dW2 += dlogit (x) h; db2 += dlogit;
// dpre = (W2^T dlogit) * leaky'(pre)   (the source gradient of the weighted average is folded into fc1-dgrad)
__global__ __launch_bounds__(256) void attn_pixel_bwd_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                                             const float *__restrict__ hidden,
                                                             const float *__restrict__ attn, const float *__restrict__ w2,
                                                             const float *__restrict__ dout,
                                                             float *__restrict__ dhidden, float *__restrict__ dw2,
                                                             float *__restrict__ db2, AttnGeom g, int M) {
    constexpr int PIX = 8;
    __shared__ float w2s[NTAP][NH + 1];
    __shared__ float dl[PIX][32];
    __shared__ float da[PIX][32];
    __shared__ float cw[PIX][NTAP][4];
    __shared__ int ci[PIX][NTAP][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * PIX;
    for (int i = tid; i < NTAP * NH; i += 256) w2s[i / NH][i % NH] = w2[i];
    {
        const int p = tid >> 5, q = tid & 31, m = m0 + p;
        if (q < NTAP && m < M) {
            const Row r = load_row(g, flow, m);
            const Corner c = k1_corner(r.y, r.x, r.fx, r.fy, q, g.H, g.W);
            cw[p][q][0] = c.w00; cw[p][q][1] = c.w01; cw[p][q][2] = c.w10; cw[p][q][3] = c.w11;
            ci[p][q][0] = c.i00; ci[p][q][1] = c.i01; ci[p][q][2] = c.i10; ci[p][q][3] = c.i11;
        }
        da[p][q] = 0.f;
    }
    __syncthreads();
    // da: each wave takes 2 pixels; lanes stride channels; 25 wave reductions per pixel
    const int CV = g.C >> 2;
    for (int pp = 0; pp < 2; ++pp) {
        const int p = wave * 2 + pp, m = m0 + p;
        if (m >= M) continue;
        const int b = m / (g.H * g.W);
        const float *s = src + (size_t)b * g.H * g.W * g.C;
        for (int q = 0; q < NTAP; ++q) {
            float part = 0.f;
            for (int cv = lane; cv < CV; cv += 64) {
                const float4 go = *reinterpret_cast<const float4 *>(dout + (size_t)m * g.C + cv * 4);
                float4 sv = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float w = cw[p][q][t];
                    const size_t off = (size_t)ci[p][q][t] * g.C + cv * 4;
                    const float4 v = *reinterpret_cast<const float4 *>(s + off);
                    sv.x += w * v.x; sv.y += w * v.y; sv.z += w * v.z; sv.w += w * v.w;
                }
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
    // dpre and dW2/db2
    for (int i = tid; i < PIX * NH; i += 256) {
        const int p = i / NH, j = i % NH, m = m0 + p;
        if (m >= M) continue;
        float s = 0.f;
#pragma unroll 5
        for (int q = 0; q < NTAP; ++q) s += dl[p][q] * w2s[q][j];
        const float pre = hidden[(size_t)m * NH + j];
        dhidden[(size_t)m * NH + j] = pre > 0.f ? s : 0.01f * s;
    }
    for (int i = tid; i < NTAP * NH; i += 256) {
        const int q = i / NH, j = i % NH;
        float s = 0.f;
        for (int p = 0; p < PIX; ++p) {
            const int m = m0 + p;
            if (m < M) {
                const float pre = hidden[(size_t)m * NH + j];
                s += dl[p][q] * (pre > 0.f ? pre : 0.01f * pre);
            }
        }
        atomicAdd(&dw2[i], s);
    }
    if (tid < NTAP) {
        float s = 0.f;
        for (int p = 0; p < PIX; ++p)
            if (m0 + p < M) s += dl[p][tid];
        atomicAdd(&db2[tid], s);
    }
}

// ------------------------------------------------------------------------------------------ fc1 wgrad
// dW1[n][k] += sum_m dpre[m][n] * A[m][k]; rows n (128), cols k tile of 128, reduction over pixels (32 per step)
__global__ __launch_bounds__(256) void attn_wgrad_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                         const float *__restrict__ flow, const float *__restrict__ dpre,
                                                         float *__restrict__ dw1, AttnGeom g, int M, int K,
                                                         int m_per_split) {
    constexpr int BM = 128, BN = 128, BK = 32, LDA = BM + 4, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int j0 = blockIdx.x * BN;
    const int m_begin = blockIdx.y * m_per_split, m_end = min(M, m_begin + m_per_split);
    const int col4 = tid & 31, krow0 = tid >> 5;
    const int kcol = j0 + col4 * 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    auto load_tiles = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb + krow0 + 8 * i;
            if (m < m_end) {
                ra[i] = *reinterpret_cast<const float4 *>(dpre + (size_t)m * NH + col4 * 4);
                const Row r = load_row(g, flow, m);
                rb[i] = attn_gather(g, src, tgt, r, kcol);
            } else {
                ra[i] = make_float4(0, 0, 0, 0);
                rb[i] = make_float4(0, 0, 0, 0);
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&As[(krow0 + 8 * i) * LDA + col4 * 4]) = ra[i];
            *reinterpret_cast<float4 *>(&Bs[(krow0 + 8 * i) * LDB + col4 * 4]) = rb[i];
        }
    };
    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles();
    }
    __syncthreads();
    for (int mb = m_begin; mb < m_end; mb += BK) {
        if (mb + BK < m_end) load_tiles(mb + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(2 * ks + lh) * LDA + wm * 64 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(2 * ks + lh) * LDB + wn * 64 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (mb + BK < m_end) store_tiles();
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = j0 + wn * 64 + j * 32 + l31;
                if (k < K) atomicAdd(&dw1[(size_t)n * K + k], acc[i][j][r]);
            }
        }
}

// ------------------------------------------------------------------------------------------ fc1 dgrad
// dA[m][k] = sum_n dpre[m][n] W1[n][k], scattered: target half -> dtarget[clamped neighbour], source half -> the
// four bilinear taps of dsource (what K2 does, block_extractor_kernel.cu:158-161)
__global__ __launch_bounds__(256) void attn_dgrad_kernel(const float *__restrict__ flow, const float *__restrict__ dpre,
                                                         const float *__restrict__ w1, const float *__restrict__ attn,
                                                         const float *__restrict__ dout, float *__restrict__ dsrc,
                                                         float *__restrict__ dtgt, AttnGeom g, int M, int K) {
    constexpr int BM = 64, BN = 128, BK = 32, LDA = BM + 1, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbn = K / BN;
    const int tile = hoig_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nbn) * BM, j0 = (tile % nbn) * BN;
    const int kc = (tid & 7) * 4, lrow = tid >> 3;
    const int col4 = tid & 31, krow0 = tid >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int nb = 0; nb < NH; nb += BK) {
        // A' = dpre[m][nb..nb+31] -> As[n][m] (transposed write); B' = W1[nb+n][j0..] -> Bs[n][col]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + lrow + 32 * i;
            float4 v = make_float4(0, 0, 0, 0);
            if (m < M) v = *reinterpret_cast<const float4 *>(dpre + (size_t)m * NH + nb + kc);
            const int r = lrow + 32 * i;
            As[(kc + 0) * LDA + r] = v.x; As[(kc + 1) * LDA + r] = v.y;
            As[(kc + 2) * LDA + r] = v.z; As[(kc + 3) * LDA + r] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = krow0 + 8 * i;
            *reinterpret_cast<float4 *>(&Bs[n * LDB + col4 * 4]) =
                *reinterpret_cast<const float4 *>(w1 + (size_t)(nb + n) * K + j0 + col4 * 4);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const float a = As[(2 * ks + lh) * LDA + wm * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float b = Bs[(2 * ks + lh) * LDB + wn * 64 + j * 32 + l31];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int C2 = 2 * g.C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        const Row row = load_row(g, flow, m);
        const size_t img = (size_t)row.b * g.H * g.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = j0 + wn * 64 + j * 32 + l31;
            const int q = k / C2, ch = k - q * C2;
            const float vv = acc[j][r];
            if (ch < g.C) {
                const int idx = target_index(row.y, row.x, q, g.H, g.W);
                atomicAdd(dtgt + (img + idx) * g.C + ch, vv);
            } else {
                const Corner c = k1_corner(row.y, row.x, row.fx, row.fy, q, g.H, g.W);
                float *d = dsrc + img * g.C + (ch - g.C);
                // + gradient of the (1/25) sum_q a_q s_q output path: same taps, same weights
                const float v = vv + attn[(size_t)m * NTAP + q] * (1.f / NTAP) * dout[(size_t)m * g.C + (ch - g.C)];
                atomicAdd(d + (size_t)c.i00 * g.C, c.w00 * v);
                atomicAdd(d + (size_t)c.i01 * g.C, c.w01 * v);
                atomicAdd(d + (size_t)c.i10 * g.C, c.w10 * v);
                atomicAdd(d + (size_t)c.i11 * g.C, c.w11 * v);
            }
        }
    }
}

}  // namespace

extern "C" int hoig_local_attn_fwd(const float *source, const float *target, const float *flow, const float *w1,
                                   const float *b1, const float *w2, const float *b2, float *hidden, float *attn,
                                   float *out, int B, int H, int W, int C, int precision, hoig_stream_t stream) {
    (void)precision;
    if (!source || !target || !flow || !w1 || !b1 || !w2 || !b2 || !hidden || !attn || !out) return HOIG_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return HOIG_EINVAL;
    if ((2 * C) % 128 != 0) return HOIG_EUNSUPPORTED;   // a 128-wide k tile must stay inside one (tap, half)
    hipStream_t st = (hipStream_t)stream;
    AttnGeom g{B, H, W, C};
    const int M = B * H * W, K = NTAP * 2 * C;
    attn_fc1_kernel<<<(M + 31) / 32, 256, 0, st>>>(source, target, flow, w1, b1, hidden, g, M, K);
    HOIG_LAUNCH_CHECK();
    attn_pixel_fwd_kernel<<<(M + 7) / 8, 256, 0, st>>>(source, flow, hidden, w2, b2, attn, out, g, M);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_local_attn_bwd(const float *source, const float *target, const float *flow, const float *w1,
                                   const float *w2, const float *hidden, const float *attn, const float *dout,
                                   float *dsource, float *dtarget, float *dw1, float *db1, float *dw2, float *db2,
                                   float *dhidden, int B, int H, int W, int C, int precision, hoig_stream_t stream) {
    (void)precision;
    if (!source || !target || !flow || !w1 || !w2 || !hidden || !attn || !dout || !dsource || !dtarget || !dw1 || !db1 ||
        !dw2 || !db2 || !dhidden)
        return HOIG_EINVAL;
    if ((2 * C) % 128 != 0) return HOIG_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    AttnGeom g{B, H, W, C};
    const int M = B * H * W, K = NTAP * 2 * C;
    attn_pixel_bwd_kernel<<<(M + 7) / 8, 256, 0, st>>>(source, flow, hidden, attn, w2, dout, dhidden, dw2, db2, g, M);
    HOIG_LAUNCH_CHECK();
    int rc = hoig_colsum_accum(dhidden, db1, M, NH, stream);
    if (rc) return rc;
    const int ncol = K / 128;
    int splits = (int)hoig_cdiv(1024, ncol);
    const int max_splits = (int)hoig_cdiv(M, 256);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int mps = (int)hoig_cdiv(hoig_cdiv(M, splits), 32) * 32;
    splits = (int)hoig_cdiv(M, mps);
    attn_wgrad_kernel<<<dim3(ncol, splits), 256, 0, st>>>(source, target, flow, dhidden, dw1, g, M, K, mps);
    HOIG_LAUNCH_CHECK();
    const int nblk = (int)hoig_cdiv(M, 64) * ncol;
    attn_dgrad_kernel<<<nblk, 256, 0, st>>>(flow, dhidden, w1, attn, dout, dsource, dtarget, g, M, K);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
