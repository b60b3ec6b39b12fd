// Sampling / resizing kernels (NHWC fp32, lanes along channels for coalesced gathers) and the NCHW drop-ins for the
// reference's two pybind CUDA ops.
//   grid_sample  : Generator.stn generator.py:475-478 (bilinear, zeros padding, align_corners=False)
//   resize (ac)  : Generator.resize_trans generator.py:466-473 (bilinear, align_corners=True)
//   nearest      : spade.py:30
//   attn_flow    : generator.py:484-488
//   block_extractor / local_attn_reshape: thirdparty/*/..._kernel.cu (K1-K4)
#include "common.h"

namespace {
constexpr int NT = 256;

__global__ void grid_sample_fwd_kernel(const float *__restrict__ x, const float *__restrict__ grid, float *__restrict__ y,
                                       int B, int H, int W, int C, int Ho, int Wo) {
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int b = (int)(p / ((int64_t)Ho * Wo));
        const float gx = grid[p * 2], gy = grid[p * 2 + 1];
        // unnormalise, align_corners=False: ((g + 1) * size - 1) / 2
        const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
        const float *xb = x + (size_t)b * H * W * C + c;
        float v = 0.f;
        const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
        if (y0ok && x0ok) v += xb[((size_t)y0 * W + x0) * C] * (wx0 * wy0);
        if (y0ok && x1ok) v += xb[((size_t)y0 * W + x1) * C] * (wx1 * wy0);
        if (y1ok && x0ok) v += xb[((size_t)y1 * W + x0) * C] * (wx0 * wy1);
        if (y1ok && x1ok) v += xb[((size_t)y1 * W + x1) * C] * (wx1 * wy1);
        y[i] = v;
    }
}

__global__ void grid_sample_bwd_kernel(const float *__restrict__ grid, const float *__restrict__ dy, float *__restrict__ dx,
                                       int B, int H, int W, int C, int Ho, int Wo) {
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        const int64_t p = i / C;
        const int b = (int)(p / ((int64_t)Ho * Wo));
        const float gx = grid[p * 2], gy = grid[p * 2 + 1];
        const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
        float *db = dx + (size_t)b * H * W * C + c;
        const float g = dy[i];
        const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
        if (y0ok && x0ok) atomicAdd(&db[((size_t)y0 * W + x0) * C], g * (wx0 * wy0));
        if (y0ok && x1ok) atomicAdd(&db[((size_t)y0 * W + x1) * C], g * (wx1 * wy0));
        if (y1ok && x0ok) atomicAdd(&db[((size_t)y1 * W + x0) * C], g * (wx0 * wy1));
        if (y1ok && x1ok) atomicAdd(&db[((size_t)y1 * W + x1) * C], g * (wx1 * wy1));
    }
}

__global__ void resize_bilinear_ac_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int Hi, int Wi, int C,
                                          int Ho, int Wo) {
    // ATen upsample_bilinear2d, align_corners=True: scale = (in-1)/(out-1)
    const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), b = (int)(p / Ho);
        const float h1r = sh * ho, w1r = sw * wo;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < Hi - 1 ? 1 : 0, w1p = w1 < Wi - 1 ? 1 : 0;
        const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
        const float *s = x + (((size_t)b * Hi + h1) * Wi + w1) * C + c;
        y[i] = h0l * (w0l * s[0] + w1l * s[(size_t)w1p * C]) +
               h1l * (w0l * s[(size_t)h1p * Wi * C] + w1l * s[((size_t)h1p * Wi + w1p) * C]);
    }
}

__global__ void resize_nearest_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int Hi, int Wi, int C,
                                      int Ho, int Wo) {
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), b = (int)(p / Ho);
        const int hi = min((int)floorf(ho * sh), Hi - 1), wi = min((int)floorf(wo * sw), Wi - 1);
        y[i] = x[(((size_t)b * Hi + hi) * Wi + wi) * C + c];
    }
}

__global__ void attn_flow_kernel(const float *__restrict__ t, float *__restrict__ flow, int B, int h) {
    const int64_t n = (int64_t)B * h * h;
    const float step = 2.0f / (float)h;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % h), y = (int)((i / h) % h), b = (int)(i / ((int64_t)h * h));
        // idt[...,0] = xx[y][x] = ax[y] (ij meshgrid: the ROW coordinate), idt[...,1] = ax[x]
        const float i0 = -1.0f + y * step, i1 = -1.0f + x * step;
        flow[((size_t)b * 2 + 0) * h * h + (size_t)y * h + x] = t[i * 2 + 0] - i0;
        flow[((size_t)b * 2 + 1) * h * h + (size_t)y * h + x] = t[i * 2 + 1] - i1;
    }
}

// ---------------------------------------------------------------- NCHW drop-ins for the reference's native ops
struct Taps {
    int xL, xR, yT, yB;
    float xL_P, xR_P, yT_P, yB_P;
};
__device__ __forceinline__ Taps k1_taps(const float *flow, int b, int yf, int xf, int yo, int xo, int Hf, int Wf, int Hs,
                                        int Ws) {
    // block_extractor_kernel.cu:62-76 (flow in pixel units, ch1 = y, ch0 = x; indices clamped to the border)
    const float flow_y = flow[((size_t)(b * 2 + 1) * Hf + yf) * Wf + xf] + yo;
    const float flow_x = flow[((size_t)(b * 2 + 0) * Hf + yf) * Wf + xf] + xo;
    const float dy = flow_y + (float)yf, dx = flow_x + (float)xf;
    Taps t;
    const float fx = floorf(dx), fy = floorf(dy);
    t.xL = max(min((int)fx, Ws - 1), 0);
    t.xR = max(min((int)fx + 1, Ws - 1), 0);
    t.yT = max(min((int)fy, Hs - 1), 0);
    t.yB = max(min((int)fy + 1, Hs - 1), 0);
    t.xR_P = dx - fx;
    t.xL_P = 1.f - t.xR_P;
    t.yB_P = dy - fy;
    t.yT_P = 1.f - t.yB_P;
    return t;
}

__global__ void block_extractor_fwd_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                           float *__restrict__ out, int B, int C, int Hs, int Ws, int Hf, int Wf, int k) {
    const int H = k * Hf, W = k * Wf;
    const int64_t n = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
        const Taps t = k1_taps(flow, b, y / k, x / k, y % k - k / 2, x % k - k / 2, Hf, Wf, Hs, Ws);
        const float *s = src + ((size_t)b * C + c) * Hs * Ws;
        float v = 0.f;
        v += t.xL_P * t.yT_P * s[(size_t)t.yT * Ws + t.xL];
        v += t.xR_P * t.yT_P * s[(size_t)t.yT * Ws + t.xR];
        v += t.xL_P * t.yB_P * s[(size_t)t.yB * Ws + t.xL];
        v += t.xR_P * t.yB_P * s[(size_t)t.yB * Ws + t.xR];
        out[i] = v;
    }
}

__global__ void block_extractor_bwd_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                                           const float *__restrict__ gout, float *__restrict__ gsrc,
                                           float *__restrict__ gflow, int B, int C, int Hs, int Ws, int Hf, int Wf,
                                           int k) {
    const int H = k * Hf, W = k * Wf;
    const int64_t n = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
        const int yf = y / k, xf = x / k;
        const Taps t = k1_taps(flow, b, yf, xf, y % k - k / 2, x % k - k / 2, Hf, Wf, Hs, Ws);
        const size_t sb = ((size_t)b * C + c) * Hs * Ws;
        const float vLT = src[sb + (size_t)t.yT * Ws + t.xL], vRT = src[sb + (size_t)t.yT * Ws + t.xR];
        const float vLB = src[sb + (size_t)t.yB * Ws + t.xL], vRB = src[sb + (size_t)t.yB * Ws + t.xR];
        const float g = gout[i];
        atomicAdd(&gsrc[sb + (size_t)t.yT * Ws + t.xL], g * t.xL_P * t.yT_P);
        atomicAdd(&gsrc[sb + (size_t)t.yT * Ws + t.xR], g * t.xR_P * t.yT_P);
        atomicAdd(&gsrc[sb + (size_t)t.yB * Ws + t.xL], g * t.xL_P * t.yB_P);
        atomicAdd(&gsrc[sb + (size_t)t.yB * Ws + t.xR], g * t.xR_P * t.yB_P);
        if (gflow) {
            const float gy = g * (-t.xL_P * vLT - t.xR_P * vRT + t.xL_P * vLB + t.xR_P * vRB);
            const float gx = g * (-t.yT_P * vLT - t.yB_P * vLB + t.yT_P * vRT + t.yB_P * vRB);
            atomicAdd(&gflow[((size_t)(b * 2 + 1) * Hf + yf) * Wf + xf], gy);
            atomicAdd(&gflow[((size_t)(b * 2 + 0) * Hf + yf) * Wf + xf], gx);
        }
    }
}

__global__ void local_attn_reshape_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int Hs, int Ws,
                                          int k, int backward) {
    // forward : out[b,0,y,x] = in[b,(y%k)*k + x%k, y/k, x/k]  (local_attn_reshape_kernel.cu:52-58)
    // backward: in-shaped gradient gathers from out-shaped gradient (:106; no contention, so a plain store)
    const int H = k * Hs, W = k * Ws;
    const int64_t n = (int64_t)B * H * W;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
        const int cs = (y % k) * k + x % k;
        const size_t j = (((size_t)b * k * k + cs) * Hs + y / k) * Ws + x / k;
        if (!backward) out[i] = in[j];
        else out[j] += in[i];
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hoig_grid_sample_fwd(const float *x, const float *grid, float *y, int B, int H, int W, int C, int Ho, int Wo,
                                    hoig_stream_t stream) {
    if (!x || !grid || !y) return HOIG_EINVAL;
    grid_sample_fwd_kernel<<<hoig_stream_grid((int64_t)B * Ho * Wo * C, NT), NT, 0, ST>>>(x, grid, y, B, H, W, C, Ho, Wo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_grid_sample_bwd(const float *grid, const float *dy, float *dx, int B, int H, int W, int C, int Ho,
                                    int Wo, hoig_stream_t stream) {
    if (!grid || !dy || !dx) return HOIG_EINVAL;
    grid_sample_bwd_kernel<<<hoig_stream_grid((int64_t)B * Ho * Wo * C, NT), NT, 0, ST>>>(grid, dy, dx, B, H, W, C, Ho, Wo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_resize_bilinear_ac(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                       hoig_stream_t stream) {
    if (!x || !y) return HOIG_EINVAL;
    resize_bilinear_ac_kernel<<<hoig_stream_grid((int64_t)B * Ho * Wo * C, NT), NT, 0, ST>>>(x, y, B, Hi, Wi, C, Ho, Wo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_resize_nearest(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                                   hoig_stream_t stream) {
    if (!x || !y) return HOIG_EINVAL;
    resize_nearest_kernel<<<hoig_stream_grid((int64_t)B * Ho * Wo * C, NT), NT, 0, ST>>>(x, y, B, Hi, Wi, C, Ho, Wo);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_attn_flow(const float *tscale, float *flow, int B, int h, hoig_stream_t stream) {
    if (!tscale || !flow) return HOIG_EINVAL;
    attn_flow_kernel<<<hoig_stream_grid((int64_t)B * h * h, NT), NT, 0, ST>>>(tscale, flow, B, h);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_block_extractor_forward(const float *source, const float *flow, float *output, int B, int C, int Hs,
                                            int Ws, int Hf, int Wf, int kernel_size, hoig_stream_t stream) {
    if (!source || !flow || !output || kernel_size <= 0) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * C * Hf * Wf * kernel_size * kernel_size;
    block_extractor_fwd_kernel<<<hoig_stream_grid(n, NT), NT, 0, ST>>>(source, flow, output, B, C, Hs, Ws, Hf, Wf,
                                                                      kernel_size);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_block_extractor_backward(const float *source, const float *flow, const float *grad_output,
                                             float *grad_source, float *grad_flow, int B, int C, int Hs, int Ws, int Hf,
                                             int Wf, int kernel_size, hoig_stream_t stream) {
    if (!source || !flow || !grad_output || !grad_source || kernel_size <= 0) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * C * Hf * Wf * kernel_size * kernel_size;
    block_extractor_bwd_kernel<<<hoig_stream_grid(n, NT), NT, 0, ST>>>(source, flow, grad_output, grad_source, grad_flow, B,
                                                                      C, Hs, Ws, Hf, Wf, kernel_size);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_local_attn_reshape_forward(const float *inputs, float *output, int B, int Hs, int Ws, int kernel_size,
                                               hoig_stream_t stream) {
    if (!inputs || !output || kernel_size <= 0) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * Hs * Ws * kernel_size * kernel_size;
    local_attn_reshape_kernel<<<hoig_stream_grid(n, NT), NT, 0, ST>>>(inputs, output, B, Hs, Ws, kernel_size, 0);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
extern "C" int hoig_local_attn_reshape_backward(const float *grad_output, float *grad_inputs, int B, int Hs, int Ws,
                                                int kernel_size, hoig_stream_t stream) {
    if (!grad_output || !grad_inputs || kernel_size <= 0) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * Hs * Ws * kernel_size * kernel_size;
    local_attn_reshape_kernel<<<hoig_stream_grid(n, NT), NT, 0, ST>>>(grad_output, grad_inputs, B, Hs, Ws, kernel_size, 1);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
