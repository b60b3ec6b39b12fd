// Instance normalisation over NHWC fp32 (HBM-bound streaming kernels, 16-B accesses per lane).
//   reference: nn.InstanceNorm2d (eps 1e-5, biased variance, no running stats) at generator.py:16-22,101-120,154-208,
//   spade.py:13 (param-free, then SPADE modulate spade.py:36), discriminator.py:37,45 via base_network.py:31.
// stats  : per (b,c) shifted sums over H*W split across workgroups, one atomic per (workgroup, channel) -> finalise
// apply  : y = act((x-mean)*rstd*scale+shift) (+ residual)       scale/shift = 1/0 | weight/bias[c] | 1+gamma/beta
// bwd    : dx = rstd*(g' - mean(g') - xhat*mean(g'*xhat)),  g' = dy*act'(y)*scale ; affine / SPADE parameter grads
#include "conv_bf16_common.h"
#include <cstdlib>

namespace {

constexpr int NT = 256;

inline int inorm_chunks(int HW) {
    // >= 16 pixel rows per workgroup (a 32x32 map already yields 64 workgroups per image), at most 128 per
    // image: every workgroup closes with one atomic per (channel, moment) into the image's accumulators, the workgroups of a
    // launch finish together, and same-address atomics retire at ~25 ns each -- 512 chunks were a 13-us tail on every launch
    constexpr int cap = 128;
    int n = (HW + 15) / 16;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    return n;
}

// partial[b][chunk][0/1][c] = sum (v - pivot), sum (v - pivot)^2 over the chunk's rows; pivot = x[b][0][c]
// MODE_BWD: v = g' (sum) and g'*xhat (second sum), no pivot.
template <bool BWD>
__global__ __launch_bounds__(NT) void inorm_partial_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                           const float *__restrict__ rstd, int mode,
                                                           const float *__restrict__ p0, const float *__restrict__ y,
                                                           const float *__restrict__ dy, int act, float slope, int HW,
                                                           int C, int nchunks, float *__restrict__ partial, int ldp,
                                                           const float *__restrict__ p1) {
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int CV = C >> 2;                 // float4 columns
    const int rows_per = (HW + nchunks - 1) / nchunks;
    const int r0 = chunk * rows_per, r1 = min(HW, r0 + rows_per);
    const int lanes_per_row = CV < NT ? CV : NT;
    const int row_lanes = NT / lanes_per_row;
    const int cv0 = threadIdx.x % lanes_per_row, rl = threadIdx.x / lanes_per_row;
    extern __shared__ float red[];         // [row_lanes][2][C]
    for (int cv = cv0; cv < CV; cv += lanes_per_row) {
        const int c = cv * 4;
        float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
        float4 pv = make_float4(0, 0, 0, 0), mu = pv, rs = pv, sc = make_float4(1, 1, 1, 1), aw = sc, ab = pv;
        if (!BWD) {
            pv = *reinterpret_cast<const float4 *>(x + (size_t)b * HW * C + c);
        } else {
            mu = *reinterpret_cast<const float4 *>(mean + (size_t)b * C + c);
            rs = *reinterpret_cast<const float4 *>(rstd + (size_t)b * C + c);
            if (mode == 1 && p1) {
                aw = *reinterpret_cast<const float4 *>(p0 + c);
                ab = *reinterpret_cast<const float4 *>(p1 + c);
            }
        }
        for (int r = r0 + rl; r < r1; r += row_lanes) {
            const size_t off = ((size_t)b * HW + r) * C + c;
            const float4 v = *reinterpret_cast<const float4 *>(x + off);
            if (!BWD) {
                const float dx = v.x - pv.x, dy_ = v.y - pv.y, dz = v.z - pv.z, dw = v.w - pv.w;
                s1.x += dx; s1.y += dy_; s1.z += dz; s1.w += dw;
                s2.x += dx * dx; s2.y += dy_ * dy_; s2.z += dz * dz; s2.w += dw * dw;
            } else {
                float4 g = *reinterpret_cast<const float4 *>(dy + off);
                const float hx = (v.x - mu.x) * rs.x, hy = (v.y - mu.y) * rs.y, hz = (v.z - mu.z) * rs.z,
                            hw = (v.w - mu.w) * rs.w;
                if (act != HOIG_ACT_NONE) {
                    if (y) {
                        const float4 yy = *reinterpret_cast<const float4 *>(y + off);
                        g.x *= hoig_act_grad_from_y(yy.x, act, slope);
                        g.y *= hoig_act_grad_from_y(yy.y, act, slope);
                        g.z *= hoig_act_grad_from_y(yy.z, act, slope);
                        g.w *= hoig_act_grad_from_y(yy.w, act, slope);
                    } else {        // (Leaky)ReLU: the sign of y is the sign of xhat * weight + bias, recomputed bit for bit
                        g.x *= hoig_act_grad_from_y(fmaf(hx, aw.x, ab.x), act, slope);
                        g.y *= hoig_act_grad_from_y(fmaf(hy, aw.y, ab.y), act, slope);
                        g.z *= hoig_act_grad_from_y(fmaf(hz, aw.z, ab.z), act, slope);
                        g.w *= hoig_act_grad_from_y(fmaf(hw, aw.w, ab.w), act, slope);
                    }
                }
                if (mode == 2) {
                    const float4 ga = *reinterpret_cast<const float4 *>(p0 + ((size_t)b * HW + r) * ldp + c);
                    g.x *= 1.f + ga.x; g.y *= 1.f + ga.y; g.z *= 1.f + ga.z; g.w *= 1.f + ga.w;
                }
                s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                s2.x += g.x * hx; s2.y += g.y * hy; s2.z += g.z * hz; s2.w += g.w * hw;
            }
        }
        (void)sc;
        float *r_ = red + (size_t)rl * 2 * C;
        *reinterpret_cast<float4 *>(r_ + c) = s1;
        *reinterpret_cast<float4 *>(r_ + C + c) = s2;
    }
    __syncthreads();
    // one fp32 atomic per (workgroup, channel, moment) into sums[b][2][C] (zeroed by the launcher): the finalise step then
    // reads two values per channel instead of walking up to 128 chunk partials (it was pure latency: ~14 us per launch)
    float *out = partial + (size_t)b * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += NT) {
        float s = 0.f;
        for (int k = 0; k < row_lanes; ++k) s += red[(size_t)k * 2 * C + i];
        atomicAdd(&out[i], s);
    }
    (void)chunk;
}

// x == NULL: the sums are plain (pivot 0: they come from a convolution's epilogue, hoig_conv2d_fwd_packed_stats)
__global__ void inorm_finalize_kernel(const float *__restrict__ x, float *__restrict__ partial, int HW, int C,
                                      int nchunks, float eps, float *__restrict__ mean, float *__restrict__ rstd,
                                      int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i - b * C;
    const float s1 = partial[(size_t)b * 2 * C + c], s2 = partial[(size_t)b * 2 * C + C + c];
    (void)nchunks;
    const float inv = 1.f / (float)HW;
    const float d = s1 * inv;
    float var = s2 * inv - d * d;
    var = var > 0.f ? var : 0.f;
    mean[i] = (x ? x[(size_t)b * HW * C + c] : 0.f) + d;
    rstd[i] = 1.f / sqrtf(var + eps);
    partial[(size_t)b * 2 * C + c] = 0.f;          // leave the accumulators zeroed for the next call (no memset launches)
    partial[(size_t)b * 2 * C + C + c] = 0.f;
}

// sums[b][0/1][c] for the backward; affine grads accumulate atomically
__global__ void inorm_bwd_finalize_kernel(float *__restrict__ partial, int C, int nchunks, int mode,
                                          float *__restrict__ sums, float *__restrict__ dweight,
                                          float *__restrict__ dbias, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i - b * C;
    const float s1 = partial[(size_t)b * 2 * C + c], s2 = partial[(size_t)b * 2 * C + C + c];
    (void)nchunks;
    sums[(size_t)b * 2 * C + c] = s1;
    sums[(size_t)b * 2 * C + C + c] = s2;
    partial[(size_t)b * 2 * C + c] = 0.f;
    partial[(size_t)b * 2 * C + C + c] = 0.f;
    if (mode == 1) {
        if (dbias) atomicAdd(&dbias[c], s1);
        if (dweight) atomicAdd(&dweight[c], s2);
    }
}

// dx as a SPLIT tensor (include/hoig_kernels.h 'PRE-SPLIT gradients'): pixel `pix` holds [hi: C bf16][lo: C bf16] in the 4C bytes an
// fp32 row would take; this lane's four channels go out as two 8-B stores.  The values are exactly those the convolution kernels'
// own split makes of the fp32 value (split4).
__device__ __forceinline__ void store_split(float *dx, int64_t pix, int C, int c, const float4 o) {
    uint2 hi, lo;
    hoig_detail::split4(o, hi, lo);
    unsigned short *row = reinterpret_cast<unsigned short *>(dx) + pix * (2 * (int64_t)C) + c;
    *reinterpret_cast<uint2 *>(row) = hi;
    *reinterpret_cast<uint2 *>(row + C) = lo;
}

__global__ __launch_bounds__(NT) void inorm_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, int mode,
                                                         const float *__restrict__ p0, const float *__restrict__ p1,
                                                         int act, float slope, const float *__restrict__ residual,
                                                         float *__restrict__ y, int HW, int C, int64_t n4, int ldp) {
    const int CV = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const int cv = (int)(i % CV);
        const int64_t pix = i / CV;
        const int b = (int)(pix / HW);
        const int c = cv * 4;
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float4 mu = *reinterpret_cast<const float4 *>(mean + (size_t)b * C + c);
        const float4 rs = *reinterpret_cast<const float4 *>(rstd + (size_t)b * C + c);
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (mode == 1) {
            sc = *reinterpret_cast<const float4 *>(p0 + c);
            sh = *reinterpret_cast<const float4 *>(p1 + c);
        } else if (mode == 2) {
            sc = *reinterpret_cast<const float4 *>(p0 + (size_t)pix * ldp + c);
            sh = *reinterpret_cast<const float4 *>(p1 + (size_t)pix * ldp + c);
            sc.x += 1.f; sc.y += 1.f; sc.z += 1.f; sc.w += 1.f;
        }
        float4 o;
        o.x = hoig_act(fmaf((v.x - mu.x) * rs.x, sc.x, sh.x), act, slope);
        o.y = hoig_act(fmaf((v.y - mu.y) * rs.y, sc.y, sh.y), act, slope);
        o.z = hoig_act(fmaf((v.z - mu.z) * rs.z, sc.z, sh.z), act, slope);
        o.w = hoig_act(fmaf((v.w - mu.w) * rs.w, sc.w, sh.w), act, slope);
        if (residual) {
            const float4 r = reinterpret_cast<const float4 *>(residual)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

__global__ __launch_bounds__(NT) void inorm_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                             const float *__restrict__ rstd, int mode,
                                                             const float *__restrict__ p0, const float *__restrict__ y,
                                                             const float *__restrict__ dy, int act, float slope,
                                                             const float *__restrict__ sums, float *__restrict__ dx,
                                                             float *__restrict__ dp0, float *__restrict__ dp1, int HW,
                                                             int C, int64_t n4, int ldp, const float *__restrict__ p1,
                                                             const float *__restrict__ addend, int split) {
    const int CV = C >> 2;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NT) {
        const int cv = (int)(i % CV);
        const int64_t pix = i / CV;
        const int b = (int)(pix / HW);
        const int c = cv * 4;
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float4 g = reinterpret_cast<const float4 *>(dy)[i];
        const float4 mu = *reinterpret_cast<const float4 *>(mean + (size_t)b * C + c);
        const float4 rs = *reinterpret_cast<const float4 *>(rstd + (size_t)b * C + c);
        float4 h;
        h.x = (v.x - mu.x) * rs.x; h.y = (v.y - mu.y) * rs.y; h.z = (v.z - mu.z) * rs.z; h.w = (v.w - mu.w) * rs.w;
        if (act != HOIG_ACT_NONE) {
            if (y) {
                const float4 yy = reinterpret_cast<const float4 *>(y)[i];
                g.x *= hoig_act_grad_from_y(yy.x, act, slope);
                g.y *= hoig_act_grad_from_y(yy.y, act, slope);
                g.z *= hoig_act_grad_from_y(yy.z, act, slope);
                g.w *= hoig_act_grad_from_y(yy.w, act, slope);
            } else {
                float4 aw = make_float4(1, 1, 1, 1), ab = make_float4(0, 0, 0, 0);
                if (mode == 1) {
                    aw = *reinterpret_cast<const float4 *>(p0 + c);
                    ab = *reinterpret_cast<const float4 *>(p1 + c);
                }
                g.x *= hoig_act_grad_from_y(fmaf(h.x, aw.x, ab.x), act, slope);
                g.y *= hoig_act_grad_from_y(fmaf(h.y, aw.y, ab.y), act, slope);
                g.z *= hoig_act_grad_from_y(fmaf(h.z, aw.z, ab.z), act, slope);
                g.w *= hoig_act_grad_from_y(fmaf(h.w, aw.w, ab.w), act, slope);
            }
        }
        float4 s1 = *reinterpret_cast<const float4 *>(sums + (size_t)b * 2 * C + c);
        float4 s2 = *reinterpret_cast<const float4 *>(sums + (size_t)b * 2 * C + C + c);
        float4 sc = make_float4(1, 1, 1, 1);
        if (mode == 1) {
            sc = *reinterpret_cast<const float4 *>(p0 + c);
            s1.x *= sc.x; s1.y *= sc.y; s1.z *= sc.z; s1.w *= sc.w;
            s2.x *= sc.x; s2.y *= sc.y; s2.z *= sc.z; s2.w *= sc.w;
        } else if (mode == 2) {
            const float4 ga = *reinterpret_cast<const float4 *>(p0 + (size_t)pix * ldp + c);
            sc = make_float4(1.f + ga.x, 1.f + ga.y, 1.f + ga.z, 1.f + ga.w);
            *reinterpret_cast<float4 *>(dp0 + (size_t)pix * ldp + c) = make_float4(g.x * h.x, g.y * h.y, g.z * h.z, g.w * h.w);
            *reinterpret_cast<float4 *>(dp1 + (size_t)pix * ldp + c) = g;
        }
        float4 o;
        o.x = rs.x * (g.x * sc.x - s1.x * inv - h.x * s2.x * inv);
        o.y = rs.y * (g.y * sc.y - s1.y * inv - h.y * s2.y * inv);
        o.z = rs.z * (g.z * sc.z - s1.z * inv - h.z * s2.z * inv);
        o.w = rs.w * (g.w * sc.w - s1.w * inv - h.w * s2.w * inv);
        if (addend) {                           // the gradient that reached x through its other consumer (a residual add)
            const float4 a = reinterpret_cast<const float4 *>(addend)[i];
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        if (split & 1) store_split(dx, pix, C, c, o);
        else reinterpret_cast<float4 *>(dx)[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small maps (H*W <= 1024: the 32x32 bottleneck, where 2/3 of the norm launches of a step live, and everything below it):
// ONE launch per norm.  A 512-thread workgroup owns a (sample, 32-channel group) slab -- at most 1024 x 32 fp32 = 128 KB --
// and holds it in REGISTERS (<= 16 float4 per thread; lanes = 8 channel vectors x 64 pixel rows, so every access is a 128-B
// row segment): statistics by a two-pass (mean, then centred squares) reduction over the registers, then normalise /
// modulate / activate (+ residual) straight from them.  x is read once and y written once (the three-kernel path reads x
// twice and pays three dependent launches of 10-15 us each on these 17-33 MB tensors).  The backward keeps x and dy slabs
// in registers the same way: one read of each instead of two.
constexpr int TNT = 512, TCG = 32, TEPT = 16, TPL = TNT / (TCG / 4);      // 64 pixel lanes
__device__ __forceinline__ float4 tile_reduce(float4 v, float *red, int cq, int pl) {
    // sum over the 64 pixel lanes of each channel vector: red is [TPL][TCG] floats
    *reinterpret_cast<float4 *>(red + pl * TCG + cq * 4) = v;
    __syncthreads();
    float s = 0.f;
    const int c = threadIdx.x % TCG, part = threadIdx.x / TCG;           // 16 partial sums of 4 lanes each per channel
#pragma unroll
    for (int k = 0; k < TPL / (TNT / TCG); ++k) s += red[(part * (TPL / (TNT / TCG)) + k) * TCG + c];
    __syncthreads();
    red[part * TCG + c] = s;
    __syncthreads();
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < TNT / TCG; ++k) {
        const float4 t = *reinterpret_cast<const float4 *>(red + k * TCG + cq * 4);
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    }
    __syncthreads();
    return o;
}

__global__ __launch_bounds__(TNT) void inorm_tile_fwd_kernel(const float *__restrict__ x, int mode, const float *__restrict__ p0,
                                                             const float *__restrict__ p1, int ldp, int act, float slope,
                                                             const float *__restrict__ residual, float eps,
                                                             float *__restrict__ y, float *__restrict__ mean,
                                                             float *__restrict__ rstd, int HW, int C) {
    __shared__ float red[TPL * TCG];
    const int b = blockIdx.y, c0 = blockIdx.x * TCG;
    const int cq = threadIdx.x % (TCG / 4), pl = threadIdx.x / (TCG / 4), c = c0 + cq * 4;
    const size_t base = (size_t)b * HW * C + c;
    float4 v[TEPT];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < TEPT; ++k) {
        const int r = pl + k * TPL;
        v[k] = r < HW ? *reinterpret_cast<const float4 *>(x + base + (size_t)r * C) : make_float4(0.f, 0.f, 0.f, 0.f);
        s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w;
    }
    s = tile_reduce(s, red, cq, pl);
    const float inv = 1.f / (float)HW;
    const float4 mu = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < TEPT; ++k)
        if (pl + k * TPL < HW) {
            const float dx = v[k].x - mu.x, dy = v[k].y - mu.y, dz = v[k].z - mu.z, dw = v[k].w - mu.w;
            q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
        }
    q = tile_reduce(q, red, cq, pl);
    const float4 rs = make_float4(1.f / sqrtf(q.x * inv + eps), 1.f / sqrtf(q.y * inv + eps), 1.f / sqrtf(q.z * inv + eps),
                                  1.f / sqrtf(q.w * inv + eps));
    if (pl == 0) {
        *reinterpret_cast<float4 *>(mean + (size_t)b * C + c) = mu;
        *reinterpret_cast<float4 *>(rstd + (size_t)b * C + c) = rs;
    }
    float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
    if (mode == 1) {
        sc = *reinterpret_cast<const float4 *>(p0 + c);
        sh = *reinterpret_cast<const float4 *>(p1 + c);
    }
#pragma unroll
    for (int k = 0; k < TEPT; ++k) {
        const int r = pl + k * TPL;
        if (r >= HW) continue;
        if (mode == 2) {
            const size_t po = ((size_t)b * HW + r) * ldp + c;
            sc = *reinterpret_cast<const float4 *>(p0 + po);
            sh = *reinterpret_cast<const float4 *>(p1 + po);
            sc.x += 1.f; sc.y += 1.f; sc.z += 1.f; sc.w += 1.f;
        }
        float4 o;
        o.x = hoig_act(fmaf((v[k].x - mu.x) * rs.x, sc.x, sh.x), act, slope);
        o.y = hoig_act(fmaf((v[k].y - mu.y) * rs.y, sc.y, sh.y), act, slope);
        o.z = hoig_act(fmaf((v[k].z - mu.z) * rs.z, sc.z, sh.z), act, slope);
        o.w = hoig_act(fmaf((v[k].w - mu.w) * rs.w, sc.w, sh.w), act, slope);
        if (residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(residual + base + (size_t)r * C);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        *reinterpret_cast<float4 *>(y + base + (size_t)r * C) = o;
    }
}

__global__ __launch_bounds__(TNT) void inorm_tile_bwd_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                             const float *__restrict__ rstd, int mode,
                                                             const float *__restrict__ p0, const float *__restrict__ p1,
                                                             int ldp, const float *__restrict__ y, const float *__restrict__ dy,
                                                             int act, float slope, float *__restrict__ dx,
                                                             float *__restrict__ dp0, float *__restrict__ dp1, int HW, int C,
                                                             const float *__restrict__ addend, int split) {
    __shared__ float red[TPL * TCG];
    const int b = blockIdx.y, c0 = blockIdx.x * TCG;
    const int cq = threadIdx.x % (TCG / 4), pl = threadIdx.x / (TCG / 4), c = c0 + cq * 4;
    const size_t base = (size_t)b * HW * C + c;
    const float4 mu = *reinterpret_cast<const float4 *>(mean + (size_t)b * C + c);
    const float4 rs = *reinterpret_cast<const float4 *>(rstd + (size_t)b * C + c);
    float4 aw = make_float4(1, 1, 1, 1), ab = make_float4(0, 0, 0, 0);
    if (mode == 1) {
        aw = *reinterpret_cast<const float4 *>(p0 + c);
        if (p1) ab = *reinterpret_cast<const float4 *>(p1 + c);
    }
    float4 h[TEPT], g[TEPT];                    // xhat and g' = dy * act'(y) (* (1 + gamma) for SPADE, applied below)
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int k = 0; k < TEPT; ++k) {
        const int r = pl + k * TPL;
        h[k] = g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= HW) continue;
        const size_t off = base + (size_t)r * C;
        const float4 v = *reinterpret_cast<const float4 *>(x + off);
        float4 gg = *reinterpret_cast<const float4 *>(dy + off);
        float4 hh;
        hh.x = (v.x - mu.x) * rs.x; hh.y = (v.y - mu.y) * rs.y; hh.z = (v.z - mu.z) * rs.z; hh.w = (v.w - mu.w) * rs.w;
        if (act != HOIG_ACT_NONE) {
            if (y) {
                const float4 yy = *reinterpret_cast<const float4 *>(y + off);
                gg.x *= hoig_act_grad_from_y(yy.x, act, slope);
                gg.y *= hoig_act_grad_from_y(yy.y, act, slope);
                gg.z *= hoig_act_grad_from_y(yy.z, act, slope);
                gg.w *= hoig_act_grad_from_y(yy.w, act, slope);
            } else {                        // (Leaky)ReLU after a plain / affine norm: the sign of y, recomputed as the forward did
                gg.x *= hoig_act_grad_from_y(fmaf(hh.x, aw.x, ab.x), act, slope);
                gg.y *= hoig_act_grad_from_y(fmaf(hh.y, aw.y, ab.y), act, slope);
                gg.z *= hoig_act_grad_from_y(fmaf(hh.z, aw.z, ab.z), act, slope);
                gg.w *= hoig_act_grad_from_y(fmaf(hh.w, aw.w, ab.w), act, slope);
            }
        }
        if (mode == 2) {                    // dgamma = g*xhat, dbeta = g (per pixel); then g' = g * (1 + gamma)
            const size_t po = ((size_t)b * HW + r) * ldp + c;
            const float4 dga = make_float4(gg.x * hh.x, gg.y * hh.y, gg.z * hh.z, gg.w * hh.w);
            *reinterpret_cast<float4 *>(dp0 + po) = dga;
            *reinterpret_cast<float4 *>(dp1 + po) = gg;
            const float4 ga = *reinterpret_cast<const float4 *>(p0 + po);
            gg.x *= 1.f + ga.x; gg.y *= 1.f + ga.y; gg.z *= 1.f + ga.z; gg.w *= 1.f + ga.w;
        }
        h[k] = hh;
        g[k] = gg;
        s1.x += gg.x; s1.y += gg.y; s1.z += gg.z; s1.w += gg.w;
        s2.x += gg.x * hh.x; s2.y += gg.y * hh.y; s2.z += gg.z * hh.z; s2.w += gg.w * hh.w;
    }
    s1 = tile_reduce(s1, red, cq, pl);
    s2 = tile_reduce(s2, red, cq, pl);
    if (mode == 1 && pl == 0) {             // affine parameter gradients: dbias += sum g, dweight += sum g*xhat (g before * weight)
        if (dp1) { atomicAdd(dp1 + c, s1.x); atomicAdd(dp1 + c + 1, s1.y); atomicAdd(dp1 + c + 2, s1.z); atomicAdd(dp1 + c + 3, s1.w); }
        if (dp0) { atomicAdd(dp0 + c, s2.x); atomicAdd(dp0 + c + 1, s2.y); atomicAdd(dp0 + c + 2, s2.z); atomicAdd(dp0 + c + 3, s2.w); }
    }
    const float inv = 1.f / (float)HW;
    float4 sc = make_float4(1, 1, 1, 1);
    if (mode == 1) {
        sc = aw;
        s1.x *= sc.x; s1.y *= sc.y; s1.z *= sc.z; s1.w *= sc.w;
        s2.x *= sc.x; s2.y *= sc.y; s2.z *= sc.z; s2.w *= sc.w;
    }
#pragma unroll
    for (int k = 0; k < TEPT; ++k) {
        const int r = pl + k * TPL;
        if (r >= HW) continue;
        float4 o;
        o.x = rs.x * (g[k].x * sc.x - s1.x * inv - h[k].x * s2.x * inv);
        o.y = rs.y * (g[k].y * sc.y - s1.y * inv - h[k].y * s2.y * inv);
        o.z = rs.z * (g[k].z * sc.z - s1.z * inv - h[k].z * s2.z * inv);
        o.w = rs.w * (g[k].w * sc.w - s1.w * inv - h[k].w * s2.w * inv);
        if (addend) {
            const float4 a = *reinterpret_cast<const float4 *>(addend + base + (size_t)r * C);
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        if (split & 1) store_split(dx, (int64_t)b * HW + r, C, c, o);
        else *reinterpret_cast<float4 *>(dx + base + (size_t)r * C) = o;
    }
}

bool tile_ok(int B, int HW, int C) {
    constexpr bool off = false;
    return !off && B > 0 && HW > 0 && HW <= TEPT * TPL && C > 0 && C % TCG == 0;
}

bool shape_ok(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return false;
    const int CV = C / 4;
    if (CV < NT && (NT % CV) != 0) return false;
    return true;
}

size_t red_bytes(int C) {
    const int CV = C / 4;
    const int lanes_per_row = CV < NT ? CV : NT;
    return (size_t)(NT / lanes_per_row) * 2 * C * sizeof(float);
}

}  // namespace

// workspace = [accumulators: B*2*C floats inside a fixed pool of ACC_POOL floats | backward sums: B*2*C floats].  The pool
// only ever holds accumulators, which every call leaves zeroed; the sums live outside it, so a later call with a larger
// B*2*C never finds them inside its accumulator range.
constexpr int64_t ACC_POOL = 1 << 18;
extern "C" int64_t hoig_inorm_workspace_bytes(int B, int HW, int C) {
    (void)HW;
    return (ACC_POOL + (int64_t)B * 2 * C) * (int64_t)sizeof(float);
}

extern "C" int hoig_inorm_stats_from_sums(int B, int HW, int C, float eps, float *mean, float *rstd, void *workspace,
                                          hoig_stream_t stream) {
    if (!mean || !rstd || !workspace) return HOIG_EINVAL;
    if (B <= 0 || HW <= 0 || C <= 0 || (int64_t)B * 2 * C > ACC_POOL) return HOIG_EUNSUPPORTED;
    const int total = B * C;
    inorm_finalize_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(nullptr, (float *)workspace, HW, C, 0, eps, mean,
                                                                                rstd, total);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_inorm_stats(const float *x, int B, int HW, int C, float eps, float *mean, float *rstd,
                                void *workspace, hoig_stream_t stream) {
    if (!x || !mean || !rstd || !workspace) return HOIG_EINVAL;
    if (!shape_ok(B, HW, C) || (int64_t)B * 2 * C > ACC_POOL) return HOIG_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nch = inorm_chunks(HW);
    float *partial = (float *)workspace;
    inorm_partial_kernel<false><<<dim3(nch, B), NT, red_bytes(C), st>>>(x, nullptr, nullptr, 0, nullptr, nullptr,
                                                                        nullptr, 0, 0.f, HW, C, nch, partial, C, nullptr);
    HOIG_LAUNCH_CHECK();
    const int total = B * C;
    inorm_finalize_kernel<<<(total + 255) / 256, 256, 0, st>>>(x, partial, HW, C, nch, eps, mean, rstd, total);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_inorm_apply_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                   const float *p1, int ld_p, int act, float slope, const float *residual, float *y, int B,
                                   int HW, int C, hoig_stream_t stream);
extern "C" int hoig_inorm_apply(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                const float *p1, int act, float slope, const float *residual, float *y, int B, int HW,
                                int C, hoig_stream_t stream) {
    return hoig_inorm_apply_ld(x, mean, rstd, mode, p0, p1, C, act, slope, residual, y, B, HW, C, stream);
}
extern "C" int hoig_inorm_apply_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                   const float *p1, int ld_p, int act, float slope, const float *residual, float *y, int B,
                                   int HW, int C, hoig_stream_t stream) {
    if (mode == 2 && (ld_p < C || (ld_p & 3))) return HOIG_EINVAL;
    if (!x || !mean || !rstd || !y || mode < 0 || mode > 2) return HOIG_EINVAL;
    if (mode != 0 && (!p0 || !p1)) return HOIG_EINVAL;
    if (C & 3) return HOIG_EUNSUPPORTED;
    const int64_t n4 = (int64_t)B * HW * C / 4;
    inorm_apply_kernel<<<hoig_stream_grid(n4, NT), NT, 0, (hipStream_t)stream>>>(x, mean, rstd, mode, p0, p1, act, slope,
                                                                               residual, y, HW, C, n4, ld_p);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_inorm_bwd_add_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                     const float *p1, int ld_p, const float *y, const float *dy, int act, float slope,
                                     const float *addend, float *dx, float *dp0, float *dp1, int B, int HW, int C,
                                     void *workspace, hoig_stream_t stream);
extern "C" int hoig_inorm_bwd_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                 const float *p1, int ld_p, const float *y, const float *dy, int act, float slope, float *dx,
                                 float *dp0, float *dp1, int B, int HW, int C, void *workspace, hoig_stream_t stream) {
    return hoig_inorm_bwd_add_ld(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, nullptr, dx, dp0, dp1, B, HW, C, workspace,
                                 stream);
}
// The norm as ONE FMA per element (for a consumer that applies it while it loads: hoig_conv2d_fwd_packed_normin):
// scale[b][c] = rstd * gamma, shift[b][c] = beta - mean * scale, written with row stride ld_out (the consumer's gathered channel count)
__global__ void inorm_fold_kernel(const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                  const float *__restrict__ beta, int C, int total, float *__restrict__ scale, float *__restrict__ shift,
                                  int ld_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i - b * C;
    const float sc = rstd[i] * (gamma ? gamma[c] : 1.f);
    scale[(size_t)b * ld_out + c] = sc;
    shift[(size_t)b * ld_out + c] = (beta ? beta[c] : 0.f) - mean[i] * sc;
}
extern "C" int hoig_inorm_fold(const float *mean, const float *rstd, const float *gamma, const float *beta, int B, int C, float *scale,
                               float *shift, int ld_out, hoig_stream_t stream) {
    if (!mean || !rstd || !scale || !shift || B <= 0 || C <= 0 || ld_out < C) return HOIG_EINVAL;
    const int total = B * C;
    inorm_fold_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(mean, rstd, gamma, beta, C, total, scale, shift, ld_out);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_inorm_bwd(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                              const float *p1, const float *y, const float *dy, int act, float slope, float *dx, float *dp0,
                              float *dp1, int B, int HW, int C, void *workspace, hoig_stream_t stream) {
    return hoig_inorm_bwd_ld(x, mean, rstd, mode, p0, p1, C, y, dy, act, slope, dx, dp0, dp1, B, HW, C, workspace, stream);
}
static int inorm_bwd_any(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1, int ld_p,
                         const float *y, const float *dy, int act, float slope, const float *addend, float *dx, float *dp0, float *dp1,
                         int B, int HW, int C, void *workspace, hoig_stream_t stream, int split);
extern "C" int hoig_inorm_bwd_add_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                     const float *p1, int ld_p, const float *y, const float *dy, int act, float slope,
                                     const float *addend, float *dx, float *dp0, float *dp1, int B, int HW, int C,
                                     void *workspace, hoig_stream_t stream) {
    return inorm_bwd_any(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, addend, dx, dp0, dp1, B, HW, C, workspace, stream, 0);
}
// the same with dx written as a SPLIT tensor (uint16 planes in the fp32 tensor's bytes): the gradient of a convolution output whose
// backward reads pre-split dy (hoig_conv2d_bwd_weight_split / hoig_conv2d_bwd_data_packed_split)
extern "C" int hoig_inorm_bwd_add_ld_split(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                           const float *p1, int ld_p, const float *y, const float *dy, int act, float slope,
                                           const float *addend, uint16_t *dx_split, float *dp0, float *dp1, int B, int HW, int C,
                                           void *workspace, hoig_stream_t stream) {
    return inorm_bwd_any(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, addend, reinterpret_cast<float *>(dx_split), dp0, dp1, B,
                         HW, C, workspace, stream, 1);
}
static int inorm_bwd_any(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1, int ld_p,
                         const float *y, const float *dy, int act, float slope, const float *addend, float *dx, float *dp0, float *dp1,
                         int B, int HW, int C, void *workspace, hoig_stream_t stream, int split) {
    if (mode == 2 && (ld_p < C || (ld_p & 3))) return HOIG_EINVAL;
    if (!x || !mean || !rstd || !dy || !dx || !workspace || mode < 0 || mode > 2) return HOIG_EINVAL;
    // y may be NULL for (Leaky)ReLU after a plain or affine instance norm: the sign of y is recomputed from x (for the
    // affine form this needs its bias p1) -- the backward then reads two tensors per pass instead of three
    if (act != HOIG_ACT_NONE && !y &&
        !((act == HOIG_ACT_RELU || act == HOIG_ACT_LRELU) && (mode == 0 || (mode == 1 && p1))))
        return HOIG_EINVAL;
    if (mode != 0 && !p0) return HOIG_EINVAL;
    if (mode == 2 && (!dp0 || !dp1)) return HOIG_EINVAL;
    if (!shape_ok(B, HW, C) || (int64_t)B * 2 * C > ACC_POOL) return HOIG_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nch = inorm_chunks(HW);
    float *partial = (float *)workspace;
    float *sums = partial + ACC_POOL;
    inorm_partial_kernel<true><<<dim3(nch, B), NT, red_bytes(C), st>>>(x, mean, rstd, mode, p0, y, dy, act, slope, HW, C,
                                                                       nch, partial, ld_p, p1);
    HOIG_LAUNCH_CHECK();
    const int total = B * C;
    inorm_bwd_finalize_kernel<<<(total + 255) / 256, 256, 0, st>>>(partial, C, nch, mode, sums, dp0, dp1, total);
    HOIG_LAUNCH_CHECK();
    const int64_t n4 = (int64_t)B * HW * C / 4;
    inorm_bwd_apply_kernel<<<hoig_stream_grid(n4, NT), NT, 0, st>>>(x, mean, rstd, mode, p0, y, dy, act, slope, sums, dx,
                                                                   dp0, dp1, HW, C, n4, ld_p, p1, addend, split);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

/* Single-launch forms for maps of at most 1024 pixels and C a multiple of 32 (HOIG_EUNSUPPORTED otherwise: the caller then
 * uses stats + apply / the three-kernel backward).  fwd: y, and mean / rstd for the backward. */
extern "C" int hoig_inorm_fwd_fused(const float *x, int mode, const float *p0, const float *p1, int ld_p, int act, float slope,
                                    const float *residual, float eps, float *y, float *mean, float *rstd, int B, int HW, int C,
                                    hoig_stream_t stream) {
    if (!x || !y || !mean || !rstd || mode < 0 || mode > 2) return HOIG_EINVAL;
    if (mode != 0 && (!p0 || !p1)) return HOIG_EINVAL;
    if (mode == 2 && (ld_p < C || (ld_p & 3))) return HOIG_EINVAL;
    if (!tile_ok(B, HW, C)) return HOIG_EUNSUPPORTED;
    inorm_tile_fwd_kernel<<<dim3(C / TCG, B), TNT, 0, (hipStream_t)stream>>>(x, mode, p0, p1, ld_p, act, slope, residual, eps, y,
                                                                             mean, rstd, HW, C);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_inorm_bwd_fused_add(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                        const float *p1, int ld_p, const float *y, const float *dy, int act, float slope,
                                        const float *addend, float *dx, float *dp0, float *dp1, int B, int HW, int C,
                                        hoig_stream_t stream);
extern "C" int hoig_inorm_bwd_fused(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                    const float *p1, int ld_p, const float *y, const float *dy, int act, float slope, float *dx,
                                    float *dp0, float *dp1, int B, int HW, int C, hoig_stream_t stream) {
    return hoig_inorm_bwd_fused_add(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, nullptr, dx, dp0, dp1, B, HW, C, stream);
}
static int inorm_bwd_fused_any(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                               int ld_p, const float *y, const float *dy, int act, float slope, const float *addend, float *dx,
                               float *dp0, float *dp1, int B, int HW, int C, hoig_stream_t stream, int split);
extern "C" int hoig_inorm_bwd_fused_add(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                        const float *p1, int ld_p, const float *y, const float *dy, int act, float slope,
                                        const float *addend, float *dx, float *dp0, float *dp1, int B, int HW, int C,
                                        hoig_stream_t stream) {
    return inorm_bwd_fused_any(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, addend, dx, dp0, dp1, B, HW, C, stream, 0);
}
extern "C" int hoig_inorm_bwd_fused_add_split(const float *x, const float *mean, const float *rstd, int mode, const float *p0,
                                              const float *p1, int ld_p, const float *y, const float *dy, int act, float slope,
                                              const float *addend, uint16_t *dx_split, float *dp0, float *dp1, int B, int HW, int C,
                                              hoig_stream_t stream) {
    return inorm_bwd_fused_any(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, addend, reinterpret_cast<float *>(dx_split), dp0,
                               dp1, B, HW, C, stream, 1);
}
static int inorm_bwd_fused_any(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                               int ld_p, const float *y, const float *dy, int act, float slope, const float *addend, float *dx,
                               float *dp0, float *dp1, int B, int HW, int C, hoig_stream_t stream, int split) {
    if (!x || !mean || !rstd || !dy || !dx || mode < 0 || mode > 2) return HOIG_EINVAL;
    if (mode == 2 && (ld_p < C || (ld_p & 3) || !dp0 || !dp1)) return HOIG_EINVAL;
    if (act != HOIG_ACT_NONE && !y &&
        !((act == HOIG_ACT_RELU || act == HOIG_ACT_LRELU) && (mode == 0 || (mode == 1 && p1))))
        return HOIG_EINVAL;
    if (mode != 0 && !p0) return HOIG_EINVAL;
    if (!tile_ok(B, HW, C)) return HOIG_EUNSUPPORTED;
    inorm_tile_bwd_kernel<<<dim3(C / TCG, B), TNT, 0, (hipStream_t)stream>>>(x, mean, rstd, mode, p0, p1, ld_p, y, dy, act, slope, dx,
                                                                             dp0, dp1, HW, C, addend, split);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
