// HBM-bound pointwise / reduction kernels of the HOGAN step: layout conversion, channel concat, bias gradients,
// alpha compositing (trainer.py:400-401), the loss terms of trainer.py:436-481, VGG max-pooling, fused Adam
// (trainer.py:275-278,425-434) and the eval.py uint8 output stage (utils/util.py:249-264).
#include "common.h"
#include <cstdio>
#include <map>
#include <mutex>

namespace {
constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v) {
    __shared__ float part[NT / 64];
    v = hoig_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) part[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += part[i];
    return s;
}

// ---- layout: 32x32 LDS-tiled transpose between [C][HW] and [HW][C] planes of one image
__global__ void transpose_kernel(const float *__restrict__ x, float *__restrict__ y, int rows, int cols) {
    // x: [batch][rows][cols] -> y: [batch][cols][rows]
    __shared__ float t[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        t[i][threadIdx.x] = (r < rows && c < cols) ? x[base + (size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < rows && c < cols) y[base + (size_t)c * rows + r] = t[threadIdx.x][i];
    }
}

__global__ void copy_channels_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t npix, int Cx, int x_off,
                                     int Cy, int y_off, int Cc, int accumulate) {
    const int64_t n = npix * Cc;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int64_t p = i / Cc;
        const int c = (int)(i - p * Cc);
        const float v = x[p * Cx + x_off + c];
        float *dst = y + p * Cy + y_off + c;
        *dst = accumulate ? *dst + v : v;
    }
}

// y[p][0:C1] = x1[p][:], y[p][C1:C1+C2] = x2[p][:] in ONE pass that walks y linearly: the destination is written as fully
// coalesced runs even when its row length is odd (the 19-channel discriminator input: 3 image + 16 condition channels --
// two strided partial-row copies into 76-B rows took 390 us each)
// Odd row lengths (the 19-channel discriminator input) without per-element index division: a workgroup owns 256 consecutive
// pixels, whose output rows are ONE contiguous run of 256 * (C1 + C2) floats.  Each thread gathers its pixel's channels into
// LDS (16-B loads where a source row is a multiple of 4 floats), then the run leaves as coalesced 16-B stores.
__global__ __launch_bounds__(256) void cat2_rows_kernel(const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2,
                                                        float *__restrict__ y, int64_t npix) {
    extern __shared__ float rows[];                       // [256][Cy]
    const int Cy = C1 + C2;
    const int64_t p0 = (int64_t)blockIdx.x * 256, p = p0 + threadIdx.x;
    if (p < npix) {
        float *r = rows + threadIdx.x * Cy;
        const float *a = x1 + p * C1, *b = x2 + p * C2;
        if ((C1 & 3) == 0)
            for (int c = 0; c < C1; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(a + c);
                r[c] = v.x; r[c + 1] = v.y; r[c + 2] = v.z; r[c + 3] = v.w;
            }
        else
            for (int c = 0; c < C1; ++c) r[c] = a[c];
        r += C1;
        if ((C2 & 3) == 0)
            for (int c = 0; c < C2; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(b + c);
                r[c] = v.x; r[c + 1] = v.y; r[c + 2] = v.z; r[c + 3] = v.w;
            }
        else
            for (int c = 0; c < C2; ++c) r[c] = b[c];
    }
    __syncthreads();
    const int64_t left = npix - p0;
    const int n = (int)(left < 256 ? left : 256) * Cy;        // floats of this workgroup's run (starts 16-B aligned: 256 * Cy * 4)
    float *dst = y + p0 * Cy;
    for (int i = threadIdx.x * 4; i + 3 < n; i += 256 * 4)
        *reinterpret_cast<float4 *>(dst + i) = make_float4(rows[i], rows[i + 1], rows[i + 2], rows[i + 3]);
    for (int i = (n & ~3) + threadIdx.x; i < n; i += 256) dst[i] = rows[i];
}

// The same without LDS: a thread produces four consecutive floats of the output stream (one integer division per thread, then a
// running (pixel, channel) pair); the scalar source reads of a wave cover a contiguous span of x1 and of x2.
__global__ __launch_bounds__(NT) void cat2_flat_kernel(const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2,
                                                       float *__restrict__ y, int64_t ntotal) {
    const int Cy = C1 + C2;
    const int64_t n4 = ntotal >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const int64_t e0 = i << 2;
        int64_t pix = e0 / Cy;
        int c = (int)(e0 - pix * Cy);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = c < C1 ? x1[pix * C1 + c] : x2[pix * C2 + (c - C1)];
            if (++c == Cy) {
                c = 0;
                ++pix;
            }
        }
        reinterpret_cast<float4 *>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (ntotal & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        const int64_t pix = e / Cy;
        const int c = (int)(e - pix * Cy);
        y[e] = c < C1 ? x1[pix * C1 + c] : x2[pix * C2 + (c - C1)];
    }
}

template <int VEC>
__global__ void cat2_kernel(const float *__restrict__ x1, int C1, const float *__restrict__ x2, int C2,
                            float *__restrict__ y, int64_t npix) {
    const int Cy = C1 + C2;
    if (VEC == 4) {
        const int Q1 = C1 >> 2, Qy = Cy >> 2;
        const int64_t n = npix * Qy;
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
            const int64_t p = i / Qy;
            const int q = (int)(i - p * Qy);
            const float4 v = q < Q1 ? reinterpret_cast<const float4 *>(x1)[p * Q1 + q]
                                    : reinterpret_cast<const float4 *>(x2)[p * (C2 >> 2) + (q - Q1)];
            reinterpret_cast<float4 *>(y)[i] = v;
        }
    } else {
        const int64_t n = npix * Cy;
        if (n < (int64_t)1 << 31) {            // 32-bit index arithmetic: the 64-bit division per element was the whole cost
            // four consecutive outputs per thread: four independent gathers in flight, one 16-B store (with one 4-B
            // element per thread the kernel ran at 0.25 TB/s: too few bytes in flight per CU)
            const unsigned n32 = (unsigned)n, n4 = n32 >> 2, step = gridDim.x * NT, cy = (unsigned)Cy;
            const unsigned c1 = (unsigned)C1, c2 = (unsigned)C2;
            for (unsigned q = blockIdx.x * NT + threadIdx.x; q < n4; q += step) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned i = q * 4 + k, p = i / cy, c = i - p * cy;
                    v[k] = c < c1 ? x1[p * c1 + c] : x2[p * c2 + (c - c1)];
                }
                reinterpret_cast<float4 *>(y)[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
            for (unsigned i = (n4 << 2) + blockIdx.x * NT + threadIdx.x; i < n32; i += step) {
                const unsigned p = i / cy, c = i - p * cy;
                y[i] = c < c1 ? x1[p * c1 + c] : x2[p * c2 + (c - c1)];
            }
            return;
        }
        for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
            const int64_t p = i / Cy;
            const int c = (int)(i - p * Cy);
            y[i] = c < C1 ? x1[p * C1 + c] : x2[p * C2 + (c - C1)];
        }
    }
}

__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const float4 u = reinterpret_cast<const float4 *>(a)[i], v = reinterpret_cast<const float4 *>(b)[i];
        reinterpret_cast<float4 *>(y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        y[i] = a[i] + b[i];
}

__global__ void add_act_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ y, int act,
                               float slope, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        y[i] = hoig_act(a[i] + b[i], act, slope);
}

__global__ void act_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy, float *__restrict__ dx,
                               int act, float slope, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        dx[i] = dy[i] * hoig_act_grad_from_y(y[i], act, slope);
}

// activation backward fused with the bias gradient of the convolution that produced y: g = dy * act'(y) is written AND its
// column sums accumulate into dbias (same slab / lane layout as colsum_kernel) -- one pass over dy instead of two
__global__ __launch_bounds__(NT) void act_bwd_colsum_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                                                            float *__restrict__ g, float *__restrict__ dbias, int act,
                                                            float slope, int64_t rows, int C, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const int lanes = C < NT ? C : NT;
    const int rl_n = NT / lanes;
    const int c0 = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    extern __shared__ float red[];
    if (threadIdx.x < lanes * rl_n) {
        for (int c = c0; c < C; c += lanes) {
            float s = 0.f;
            for (int64_t r = r0 + rl; r < r1; r += rl_n) {
                const float v = dy[r * C + c] * hoig_act_grad_from_y(y[r * C + c], act, slope);
                g[r * C + c] = v;
                s += v;
            }
            red[rl * C + c] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
        float s = 0.f;
        for (int k = 0; k < rl_n; ++k) s += red[k * C + c];
        atomicAdd(&dbias[c], s);
    }
}

// float4 form of the two kernels below for C % 4 == 0 (16-B accesses; FUSED: also writes g = dy * act'(y))
template <bool FUSED>
__global__ __launch_bounds__(NT) void colsum4_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                                                     float *__restrict__ g, float *__restrict__ out, int act, float slope,
                                                     int64_t rows, int C, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const int CV = C >> 2;
    const int lanes = CV < NT ? CV : NT;
    const int rl_n = NT / lanes;
    const int c0 = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    extern __shared__ float red[];                                     // [rl_n][C]
    if (threadIdx.x < lanes * rl_n) {
        for (int cv = c0; cv < CV; cv += lanes) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t r = r0 + rl; r < r1; r += rl_n) {
                float4 v = reinterpret_cast<const float4 *>(dy)[r * CV + cv];
                if (FUSED) {
                    const float4 yy = reinterpret_cast<const float4 *>(y)[r * CV + cv];
                    v.x *= hoig_act_grad_from_y(yy.x, act, slope);
                    v.y *= hoig_act_grad_from_y(yy.y, act, slope);
                    v.z *= hoig_act_grad_from_y(yy.z, act, slope);
                    v.w *= hoig_act_grad_from_y(yy.w, act, slope);
                    reinterpret_cast<float4 *>(g)[r * CV + cv] = v;
                }
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4 *>(&red[rl * C + cv * 4]) = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
        float s = 0.f;
        for (int k = 0; k < rl_n; ++k) s += red[k * C + c];
        atomicAdd(&out[c], s);
    }
}

// out[c] += sum_rows x[row][c]; one workgroup per row slab, lanes along channels (coalesced), LDS combine, one
// atomic per (workgroup, channel)
__global__ __launch_bounds__(NT) void colsum_kernel(const float *__restrict__ x, float *__restrict__ out, int64_t rows,
                                                    int C, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const int lanes = C < NT ? C : NT;
    const int rl_n = NT / lanes;
    const int c0 = threadIdx.x % lanes, rl = threadIdx.x / lanes;
    extern __shared__ float red[];
    if (threadIdx.x < lanes * rl_n) {
        for (int c = c0; c < C; c += lanes) {
            float s = 0.f;
            for (int64_t r = r0 + rl; r < r1; r += rl_n) s += x[r * C + c];
            red[rl * C + c] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
        float s = 0.f;
        for (int k = 0; k < rl_n; ++k) s += red[k * C + c];
        atomicAdd(&out[c], s);
    }
}

__global__ void compose_fwd_kernel(const float *__restrict__ bg, const float *__restrict__ obj,
                                   const float *__restrict__ hand, const float *__restrict__ mbg,
                                   const float *__restrict__ mh, float *__restrict__ img, int64_t npix, int C) {
    const int64_t n = npix * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int64_t p = i / C;
        const float a = mbg[p], h = mh[p];
        img[i] = a * bg[i] + (1.f - a) * (obj[i] * h + hand[i] * (1.f - h));
    }
}

__global__ void compose_bwd_kernel(const float *__restrict__ bg, const float *__restrict__ obj,
                                   const float *__restrict__ hand, const float *__restrict__ mbg,
                                   const float *__restrict__ mh, const float *__restrict__ dimg, float *__restrict__ dbg,
                                   float *__restrict__ dobj, float *__restrict__ dhand, float *__restrict__ dmbg,
                                   float *__restrict__ dmh, int64_t npix, int C) {
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < npix; p += (int64_t)gridDim.x * NT) {
        const float a = mbg[p], h = mh[p];
        float da = 0.f, dh = 0.f;
        for (int c = 0; c < C; ++c) {
            const int64_t i = p * C + c;
            const float g = dimg[i];
            const float fg = obj[i] * h + hand[i] * (1.f - h);
            dbg[i] = g * a;
            dobj[i] = g * (1.f - a) * h;
            dhand[i] = g * (1.f - a) * (1.f - h);
            da += g * (bg[i] - fg);
            dh += g * (1.f - a) * (obj[i] - hand[i]);
        }
        dmbg[p] = da;
        dmh[p] = dh;
    }
}

__global__ __launch_bounds__(NT) void loss_kernel(int kind, const float *__restrict__ pred,
                                                  const float *__restrict__ target, float tconst, float gscale,
                                                  float *__restrict__ out, float *__restrict__ dpred, int64_t n, float oscale,
                                                  bool vec) {
    float acc = 0.f;
    auto term = [&](float p, float t, float &g) -> float {
        float l;
        if (kind == HOIG_LOSS_L1) {
            const float d = p - t;
            l = fabsf(d);
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else if (kind == HOIG_LOSS_MSE) {
            const float d = p - t;
            l = d * d;
            g = 2.f * d;
        } else {  // BCE with torch's clamps: log >= -100 (forward), denominator >= 1e-12 (backward)
            const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
            l = -(t * lp + (1.f - t) * lq);
            g = (p - t) / fmaxf((1.f - p) * p, 1e-12f);
        }
        return l;
    };
    // 16-B accesses over the bulk (the perceptual loss streams 134-MB feature maps through here), scalars over the last n % 4
    const int64_t n4 = vec ? n >> 2 : 0;               // (vec: every pointer is 16-B aligned)
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const float4 p = reinterpret_cast<const float4 *>(pred)[i];
        const float4 t = target ? reinterpret_cast<const float4 *>(target)[i] : make_float4(tconst, tconst, tconst, tconst);
        float4 g;
        acc += term(p.x, t.x, g.x);
        acc += term(p.y, t.y, g.y);
        acc += term(p.z, t.z, g.z);
        acc += term(p.w, t.w, g.w);
        if (dpred) reinterpret_cast<float4 *>(dpred)[i] = make_float4(g.x * gscale, g.y * gscale, g.z * gscale, g.w * gscale);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        float g;
        acc += term(pred[i], target ? target[i] : tconst, g);
        if (dpred) dpred[i] = g * gscale;
    }
    const float s = block_sum(acc);
    if (threadIdx.x == 0) atomicAdd(out, s * oscale);
}

// `one`: both sums go to out[0], scaled by gx / gy (a term of an objective: hoig_tv_accumulate)
__global__ __launch_bounds__(NT) void tv_kernel(const float *__restrict__ m, float gx, float gy, float *__restrict__ out,
                                                float *__restrict__ dm, int B, int H, int W, bool one) {
    const int64_t n = (int64_t)B * H * W;
    float ax = 0.f, ay = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float v = m[i];
        float g = 0.f;
        if (x + 1 < W) {
            const float d = v - m[i + 1];
            ax += fabsf(d);
            g += gx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (x > 0) {
            const float d = m[i - 1] - v;
            g -= gx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (y + 1 < H) {
            const float d = v - m[i + W];
            ay += fabsf(d);
            g += gy * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (y > 0) {
            const float d = m[i - W] - v;
            g -= gy * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        if (dm) dm[i] = g;
    }
    const float sx = block_sum(ax);
    const float sy = block_sum(ay);
    if (threadIdx.x == 0) {
        if (one) {
            atomicAdd(&out[0], sx * gx + sy * gy);
        } else {
            atomicAdd(&out[0], sx);
            atomicAdd(&out[1], sy);
        }
    }
}

__global__ __launch_bounds__(NT) void sum_kernel(const float *__restrict__ x, float *__restrict__ out, int64_t n, float scale) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) acc += x[i];
    const float s = block_sum(acc);
    if (threadIdx.x == 0) atomicAdd(out, s * scale);
}

__global__ void maxpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), b = (int)(p / Ho);
        const float *s = x + (((size_t)b * H + 2 * ho) * W + 2 * wo) * C + c;
        const float a = s[0], bb = s[C], cc = s[(size_t)W * C], d = s[(size_t)W * C + C];
        y[i] = fmaxf(fmaxf(a, bb), fmaxf(cc, d));
    }
}

__global__ void maxpool_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx,
                                   int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), b = (int)(p / Ho);
        const size_t o = (((size_t)b * H + 2 * ho) * W + 2 * wo) * C + c;
        const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
        int best = 0;
        float bv = x[o];
#pragma unroll
        for (int k = 1; k < 4; ++k) {   // first maximum in scan order wins, as ATen's CPU kernel does
            const float v = x[o + offs[k]];
            if (v > bv) { bv = v; best = k; }
        }
        const float g = dy[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) dx[o + offs[k]] = (k == best) ? g : 0.f;
    }
}

// torch.optim.Adam single-tensor update (no amsgrad, no weight decay), op order as ATen applies it:
//   m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g; denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
__global__ __launch_bounds__(NT) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                  float *__restrict__ v, int64_t n, float step_size, float omb1, float b2,
                                                  float omb2, float eps, float bc2_sqrt, float gscale) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *pe = &pp.x, *me = &mm.x, *ve = &vv.x;
        const float *ge = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] = me[k] + (gk - me[k]) * omb1;
            ve[k] = ve[k] * b2 + omb2 * (gk * gk);
            const float denom = sqrtf(ve[k]) / bc2_sqrt + eps;
            pe[k] = pe[k] - step_size * (me[k] / denom);
        }
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gk = g[i] * gscale;
        const float mk = m[i] + (gk - m[i]) * omb1;
        const float vk = v[i] * b2 + omb2 * (gk * gk);
        m[i] = mk;
        v[i] = vk;
        p[i] = p[i] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
    }
}

// Adam with its schedule in DEVICE memory, so that an optimiser step is a pure function of device state and can sit inside a
// captured hipGraph: `state` = {lr, beta1, beta2, eps, step} (doubles; the host only rewrites lr), `derived` = the six fp32
// scalars adam_kernel takes by value.  One thread: advance the step count and evaluate the bias corrections in double, as
// hoig_adam_step does on the host.
__global__ void adam_tick_kernel(double *__restrict__ state, float *__restrict__ derived) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double lr = state[0], b1 = state[1], b2 = state[2], eps = state[3];
    const double step = state[4] + 1.0;
    state[4] = step;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    derived[0] = (float)(lr / bc1);
    derived[1] = (float)(1.0 - b1);
    derived[2] = (float)b2;
    derived[3] = (float)(1.0 - b2);
    derived[4] = (float)eps;
    derived[5] = (float)sqrt(bc2);
}

__global__ __launch_bounds__(NT) void adam_dev_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                      float *__restrict__ v, int64_t n, const float *__restrict__ derived,
                                                      float gscale) {
    const float step_size = derived[0], omb1 = derived[1], b2 = derived[2], omb2 = derived[3], eps = derived[4],
                bc2_sqrt = derived[5];
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *pe = &pp.x, *me = &mm.x, *ve = &vv.x;
        const float *ge = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] = me[k] + (gk - me[k]) * omb1;
            ve[k] = ve[k] * b2 + omb2 * (gk * gk);
            const float denom = sqrtf(ve[k]) / bc2_sqrt + eps;
            pe[k] = pe[k] - step_size * (me[k] / denom);
        }
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gk = g[i] * gscale;
        const float mk = m[i] + (gk - m[i]) * omb1;
        const float vk = v[i] * b2 + omb2 * (gk * gk);
        m[i] = mk;
        v[i] = vk;
        p[i] = p[i] - step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
    }
}

__global__ void tensor2im_kernel(const float *__restrict__ x, uint8_t *__restrict__ out, int B, int H, int W, int C,
                                 int ncol, int nrw, int unnorm) {
    // out: [C][nrw*H][ncol*W] uint8 ; x NHWC
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int w = (int)(p % W);
        p /= W;
        const int h = (int)(p % H), b = (int)(p / H);
        float v = x[i];
        if (unnorm) { v += 1.0f; v /= 2.0f; }
        v *= 255.0f;
        const int gr = b / ncol, gc = b % ncol;
        // numpy astype(uint8) of a float: C-style truncation, wrap on out-of-range (values here are in range)
        const int iv = (int)v;
        out[((size_t)c * (nrw * H) + gr * H + h) * ((size_t)ncol * W) + gc * W + w] = (uint8_t)(iv & 0xFF);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hoig_nchw_to_nhwc(const float *x, float *y, int B, int C, int H, int W, hoig_stream_t stream) {
    if (!x || !y) return HOIG_EINVAL;
    const int HW = H * W;  // x: [B][C][HW] -> y: [B][HW][C]
    transpose_kernel<<<dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(32, 8), 0, ST>>>(x, y, C, HW);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_nhwc_to_nchw(const float *x, float *y, int B, int C, int H, int W, hoig_stream_t stream) {
    if (!x || !y) return HOIG_EINVAL;
    const int HW = H * W;  // x: [B][HW][C] -> y: [B][C][HW]
    transpose_kernel<<<dim3((C + 31) / 32, (HW + 31) / 32, B), dim3(32, 8), 0, ST>>>(x, y, HW, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_copy_channels(const float *x, float *y, int64_t npix, int Cx, int x_off, int Cy, int y_off, int Cc,
                                  int accumulate, hoig_stream_t stream) {
    if (!x || !y || x_off + Cc > Cx || y_off + Cc > Cy) return HOIG_EINVAL;
    copy_channels_kernel<<<hoig_stream_grid(npix * Cc, NT), NT, 0, ST>>>(x, y, npix, Cx, x_off, Cy, y_off, Cc, accumulate);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_cat2_channels(const float *x1, int C1, const float *x2, int C2, float *y, int64_t npix,
                                  hoig_stream_t stream) {
    if (!x1 || !x2 || !y || C1 <= 0 || C2 <= 0) return HOIG_EINVAL;
    if (((C1 | C2) & 3) == 0)
        cat2_kernel<4><<<hoig_stream_grid(npix * ((C1 + C2) >> 2), NT), NT, 0, ST>>>(x1, C1, x2, C2, y, npix);
    else
        cat2_flat_kernel<<<hoig_stream_grid(npix * (C1 + C2) / 4 + 1, NT), NT, 0, ST>>>(x1, C1, x2, C2, y, npix * (C1 + C2));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_add(const float *a, const float *b, float *y, int64_t n, hoig_stream_t stream) {
    if (!a || !b || !y) return HOIG_EINVAL;
    add_kernel<<<hoig_stream_grid(n / 4 + 1, NT), NT, 0, ST>>>(a, b, y, n);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_add_act(const float *a, const float *b, float *y, int act, float slope, int64_t n, hoig_stream_t stream) {
    if (!a || !b || !y) return HOIG_EINVAL;
    add_act_kernel<<<hoig_stream_grid(n, NT), NT, 0, ST>>>(a, b, y, act, slope, n);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_act_bwd(const float *y, const float *dy, float *dx, int act, float slope, int64_t n,
                            hoig_stream_t stream) {
    if (!y || !dy || !dx) return HOIG_EINVAL;
    act_bwd_kernel<<<hoig_stream_grid(n, NT), NT, 0, ST>>>(y, dy, dx, act, slope, n);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_act_bwd_colsum(const float *y, const float *dy, float *g, float *dbias, int act, float slope,
                                   int64_t rows, int C, hoig_stream_t stream) {
    if (!y || !dy || !g || !dbias || C <= 0) return HOIG_EINVAL;
    // every workgroup ends with one atomic per channel into the SAME C addresses, and same-address atomics retire at ~25 ns each
    // (measured: 4096 workgroups over a [8,128,128,128] tensor took 108 us, 100 of them that queue): at most 512 workgroups
    int64_t nblk = hoig_cdiv(rows, 32);
    if (nblk > 512) nblk = 512;
    const int64_t rpb = hoig_cdiv(rows, nblk);
    nblk = hoig_cdiv(rows, rpb);
    if ((C & 3) == 0) {
        const int lanes4 = (C >> 2) < NT ? (C >> 2) : NT;
        colsum4_kernel<true><<<(int)nblk, NT, (size_t)(NT / lanes4) * C * sizeof(float), ST>>>(y, dy, g, dbias, act, slope,
                                                                                               rows, C, rpb);
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    const int lanes = C < NT ? C : NT;
    const size_t shm = (size_t)(NT / lanes) * C * sizeof(float);
    act_bwd_colsum_kernel<<<(int)nblk, NT, shm, ST>>>(y, dy, g, dbias, act, slope, rows, C, rpb);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_colsum_accum(const float *x, float *out, int64_t rows, int C, hoig_stream_t stream) {
    if (!x || !out || C <= 0) return HOIG_EINVAL;
    // >= 32 rows per workgroup, at most 512 workgroups (see hoig_act_bwd_colsum: the closing same-address atomics); streams x once
    int64_t nblk = hoig_cdiv(rows, 32);
    if (nblk > 512) nblk = 512;
    const int64_t rpb = hoig_cdiv(rows, nblk);
    nblk = hoig_cdiv(rows, rpb);
    if ((C & 3) == 0) {
        const int lanes4 = (C >> 2) < NT ? (C >> 2) : NT;
        colsum4_kernel<false><<<(int)nblk, NT, (size_t)(NT / lanes4) * C * sizeof(float), ST>>>(nullptr, x, nullptr, out, 0, 0.f,
                                                                                                rows, C, rpb);
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    const int lanes = C < NT ? C : NT;
    const size_t shm = (size_t)(NT / lanes) * C * sizeof(float);
    colsum_kernel<<<(int)nblk, NT, shm, ST>>>(x, out, rows, C, rpb);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_compose_fwd(const float *bg, const float *obj, const float *hand, const float *mbg, const float *mh,
                                float *img, int64_t npix, int C, hoig_stream_t stream) {
    if (!bg || !obj || !hand || !mbg || !mh || !img) return HOIG_EINVAL;
    compose_fwd_kernel<<<hoig_stream_grid(npix * C, NT), NT, 0, ST>>>(bg, obj, hand, mbg, mh, img, npix, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_compose_bwd(const float *bg, const float *obj, const float *hand, const float *mbg, const float *mh,
                                const float *dimg, float *dbg, float *dobj, float *dhand, float *dmbg, float *dmh,
                                int64_t npix, int C, hoig_stream_t stream) {
    if (!bg || !obj || !hand || !mbg || !mh || !dimg || !dbg || !dobj || !dhand || !dmbg || !dmh) return HOIG_EINVAL;
    compose_bwd_kernel<<<hoig_stream_grid(npix, NT), NT, 0, ST>>>(bg, obj, hand, mbg, mh, dimg, dbg, dobj, dhand, dmbg,
                                                                 dmh, npix, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
static bool loss_vec_ok(const void *a, const void *b, const void *c) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}
extern "C" int hoig_loss_fwd_bwd(int kind, const float *pred, const float *target, float target_const, float gscale,
                                 float *out, float *dpred, int64_t n, hoig_stream_t stream) {
    if (!pred || !out || kind < 0 || kind > 2) return HOIG_EINVAL;
    int g = hoig_stream_grid(n / 4 + 1, NT);
    if (g > 256) g = 256;            // (each workgroup closes with an atomic into the same address: ~25 ns apiece)
    loss_kernel<<<g, NT, 0, ST>>>(kind, pred, target, target_const, gscale, out, dpred, n, 1.f, loss_vec_ok(pred, target, dpred));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_loss_accumulate(int kind, const float *pred, const float *target, float target_const, float gscale,
                                    float *term, float *dpred, int64_t n, hoig_stream_t stream) {
    if (!pred || !term || kind < 0 || kind > 2) return HOIG_EINVAL;
    int g = hoig_stream_grid(n / 4 + 1, NT);
    if (g > 256) g = 256;
    loss_kernel<<<g, NT, 0, ST>>>(kind, pred, target, target_const, gscale, term, dpred, n, gscale, loss_vec_ok(pred, target, dpred));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_tv_fwd_bwd(const float *m, float gx, float gy, float *out, float *dm, int B, int H, int W,
                               hoig_stream_t stream) {
    if (!m || !out) return HOIG_EINVAL;
    int g = hoig_stream_grid((int64_t)B * H * W, NT);
    if (g > 512) g = 512;
    tv_kernel<<<g, NT, 0, ST>>>(m, gx, gy, out, dm, B, H, W, false);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_tv_accumulate(const float *m, float gx, float gy, float *term, float *dm, int B, int H, int W,
                                  hoig_stream_t stream) {
    if (!m || !term) return HOIG_EINVAL;
    int g = hoig_stream_grid((int64_t)B * H * W, NT);
    if (g > 512) g = 512;
    tv_kernel<<<g, NT, 0, ST>>>(m, gx, gy, term, dm, B, H, W, true);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_sum(const float *x, float *out, int64_t n, hoig_stream_t stream) {
    if (!x || !out) return HOIG_EINVAL;
    int g = hoig_stream_grid(n, NT);
    if (g > 512) g = 512;
    sum_kernel<<<g, NT, 0, ST>>>(x, out, n, 1.f);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_sum_scaled(const float *x, float scale, float *out, int64_t n, hoig_stream_t stream) {
    if (!x || !out) return HOIG_EINVAL;
    int g = hoig_stream_grid(n, NT);
    if (g > 512) g = 512;
    sum_kernel<<<g, NT, 0, ST>>>(x, out, n, scale);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_maxpool2_fwd(const float *x, float *y, int B, int H, int W, int C, hoig_stream_t stream) {
    if (!x || !y || (H & 1) || (W & 1)) return HOIG_EINVAL;
    maxpool_fwd_kernel<<<hoig_stream_grid((int64_t)B * H * W * C / 4, NT), NT, 0, ST>>>(x, y, B, H, W, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_maxpool2_bwd(const float *x, const float *y, const float *dy, float *dx, int B, int H, int W, int C,
                                 hoig_stream_t stream) {
    (void)y;
    if (!x || !dy || !dx || (H & 1) || (W & 1)) return HOIG_EINVAL;
    maxpool_bwd_kernel<<<hoig_stream_grid((int64_t)B * H * W * C / 4, NT), NT, 0, ST>>>(x, dy, dx, B, H, W, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                              double beta1, double beta2, double eps, int step, float grad_scale,
                              hoig_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return HOIG_EINVAL;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    // 1-beta evaluated in double like torch does (python floats), then rounded once
    adam_kernel<<<hoig_stream_grid(n / 4 + 1, NT), NT, 0, ST>>>(param, grad, exp_avg, exp_avg_sq, n, step_size,
                                                               (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                                                               (float)eps, bc2_sqrt, grad_scale);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_adam_tick(double *state, float *derived, hoig_stream_t stream) {
    if (!state || !derived) return HOIG_EINVAL;
    adam_tick_kernel<<<1, 64, 0, ST>>>(state, derived);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                  const float *derived, float grad_scale, hoig_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !derived) return HOIG_EINVAL;
    adam_dev_kernel<<<hoig_stream_grid(n / 4 + 1, NT), NT, 0, ST>>>(param, grad, exp_avg, exp_avg_sq, n, derived, grad_scale);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_tensor2im_u8(const float *x, uint8_t *out, int B, int H, int W, int C, int nrow, int unnormalize,
                                 hoig_stream_t stream) {
    if (!x || !out || nrow <= 0) return HOIG_EINVAL;
    const int ncol = nrow < B ? nrow : B;   // torchvision.utils.make_grid: xmaps = min(nrow, B)
    const int nrw = (B + ncol - 1) / ncol;
    tensor2im_kernel<<<hoig_stream_grid((int64_t)B * H * W * C, NT), NT, 0, ST>>>(x, out, B, H, W, C, ncol, nrw,
                                                                                 unnormalize);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
// A HIP stream of the library's own (non-blocking).  PyTorch hands out its side streams round-robin from a pool of 32 per
// device, so the 33rd torch.cuda.Stream() of a process ALIASES the first: two roles of the step then share one HIP stream, and a
// stream that forks from / joins itself inside a capture sends hip::Stream::EndCapture() into unbounded recursion (ROCm 7.2,
// seen as a segfault after ~10 Trainer objects in one process).  The step's long-lived streams are therefore created here and
// wrapped with torch.cuda.ExternalStream (hoig_amd/ops.py new_stream).
extern "C" int hoig_stream_create(hoig_stream_t *out) {
    if (!out) return HOIG_EINVAL;
    hipStream_t s = nullptr;
    // the same call PyTorch's stream pool makes (non-blocking, default priority 0)
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0) != hipSuccess) return HOIG_ELAUNCH;
    *out = (hoig_stream_t)s;
    return HOIG_OK;
}
// Per-stream scratch memory, owned by the CALLER (no allocation happens behind this ABI): a registry stream -> (pointer, bytes).
// Kernels that reduce per-workgroup partials through memory (the thin-channel weight gradients, conv_thin.hip) look their stream's
// block up here; launches of one stream are ordered, so they share it.  Without a registered block such a kernel takes its
// atomic path and says so once.
namespace {
struct Scratch { void *ptr; int64_t bytes; };
std::mutex g_scratch_mu;
// keyed by (device ordinal, stream): the null stream has the same handle on every device of a multi-device process
std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;
std::pair<int, hipStream_t> scratch_key(hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return std::make_pair(dev, st);
}
bool g_scratch_warned = false;
}  // namespace
namespace hoig_detail {
void *stream_scratch(hipStream_t st, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    auto it = g_scratch.find(scratch_key(st));
    if (it != g_scratch.end() && (size_t)it->second.bytes >= bytes) return it->second.ptr;
    if (!g_scratch_warned) {
        g_scratch_warned = true;
        fprintf(stderr, "hoig: no scratch block of %zu bytes registered for stream %p (hoig_stream_scratch_set): partial sums fall back to "
                        "atomics\n", bytes, (void *)st);
    }
    return nullptr;
}
}  // namespace hoig_detail
extern "C" int64_t hoig_stream_scratch_bytes(void) { return (int64_t)16 << 20; }
extern "C" int hoig_stream_scratch_set(hoig_stream_t stream, void *ptr, int64_t bytes) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!ptr || bytes <= 0) g_scratch.erase(scratch_key((hipStream_t)stream));
    else g_scratch[scratch_key((hipStream_t)stream)] = Scratch{ptr, bytes};
    return HOIG_OK;
}
extern "C" int hoig_stream_destroy(hoig_stream_t stream) {
    if (!stream) return HOIG_EINVAL;
    (void)hoig_stream_scratch_set(stream, nullptr, 0);
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? HOIG_OK : HOIG_ELAUNCH;
}
extern "C" const char *hoig_version(void) { return "hoig-hip 0.1 (gfx950)"; }
