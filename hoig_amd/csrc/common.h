// Shared device helpers for the gfx950 kernels (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "hoig_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define HOIG_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return HOIG_ELAUNCH;           \
    } while (0)

// hipFuncSetAttribute applies to the CURRENT device: a kernel's "attribute already set" flag remembers WHICH device ordinals have it
// (a process-wide bool left every second GPU of a multi-device process without its > 64 KB of dynamic LDS).
struct hoig_once {
    std::atomic<unsigned long long> mask{0};
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 63;
    }
    bool done() const { return (mask.load(std::memory_order_acquire) >> dev()) & 1ull; }
    void set() { mask.fetch_or(1ull << dev(), std::memory_order_release); }
};

static inline int64_t hoig_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for HBM-bound streaming kernels: cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline int hoig_stream_grid(int64_t work_items, int block) {
    int64_t g = hoig_cdiv(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float hoig_act(float v, int act, float slope) {
    switch (act) {
        case HOIG_ACT_RELU: return v > 0.f ? v : 0.f;
        case HOIG_ACT_LRELU: return v > 0.f ? v : v * slope;
        case HOIG_ACT_TANH: return tanhf(v);
        case HOIG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// epilogue form: none / ReLU / LeakyReLU are one select + one fma (negative slope 1 / 0 / slope; "+ 0" turns ReLU's -0
// into +0), tanh / sigmoid take the (wave-uniform) slow branch
__device__ __forceinline__ float fast_act(float v, float nslope, bool special, int act, float slope) {
    if (special) return hoig_act(v, act, slope);
    return v > 0.f ? v : fmaf(v, nslope, 0.f);
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float hoig_act_grad_from_y(float y, int act, float slope) {
    switch (act) {
        case HOIG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case HOIG_ACT_LRELU: return y > 0.f ? 1.f : slope;
        case HOIG_ACT_TANH: return 1.f - y * y;
        case HOIG_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous chunk of the logical tile order; bijective for any grid size (guide §5, T1).
__device__ __forceinline__ int hoig_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float hoig_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// fp32 -> bf16 (round to nearest even) as raw bits; NaN handling not needed on this path's operands
__device__ __forceinline__ unsigned short hoig_f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float hoig_bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
