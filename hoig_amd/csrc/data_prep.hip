// The per-sample image work of the reference's data loader (HOIG_HOv3/data/hov3_dataset.py:215-223,63-91,267) for a
// whole batch on the device: cv2.resize (INTER_LINEAR, 8-bit) of the masks, cv2.warpAffine (INTER_LINEAR, BORDER_CONSTANT 0) of
// frames and masks to the 256 x 256 patch, and the conversions that follow in the same kernel -- BGR -> RGB, / 255, ToTensor's
// HWC -> CHW, Normalize(0.5, 0.5) for the frame; last channel / 128 for the mask.  The decoded 8-bit frames arrive from pinned host
// memory; nothing else of a sample's pixels touches the host.
//
// Both primitives are OpenCV 4.5.1's FIXED-POINT algorithms (opencv-python==4.5.1.48, requirements.txt:99), integer for integer:
// warpAffine inverts the 2x3 matrix in double, walks the source coordinates in 10-bit fixed point rounded to 1/32 pixel and blends
// the 2x2 footprint with 15-bit weights; resize uses 11-bit coefficients, an int row buffer and VResizeLinear<uchar>'s
// ((b * (S >> 4)) >> 16) form.  The tests compare with a CPU restatement of the same algorithms (parity unpinned: DESIGN.md section 5).
// Byte work, bound by the few MB it moves: one thread per output pixel, outputs written plane by plane (coalesced).
// Built with -ffp-contract=off: the double / float products below must round like the C code they restate.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB_SIZE = 32, REMAP_BITS = 15;
constexpr int RS_SCALE = 1 << 11;

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }

// mode 0: dst = uint8 [B][Hd][Wd][C] (cv2.warpAffine's result)
// mode 1: C == 3, dst = float [B][3][Hd][Wd]: channel c = ((float(v[2 - c]) / 255) - 0.5) / 0.5          (hov3_dataset.py:222,209,267)
// mode 2: dst = float [B][1][Hd][Wd] = float(v[C - 1]) / 128                                            (hov3_dataset.py:223)
template <int MODE>
__global__ __launch_bounds__(NT) void warp_affine_u8_kernel(const uint8_t *__restrict__ src, int B, int Hs, int Ws, int C,
                                                            const double *__restrict__ Mfwd, int Hd, int Wd, void *__restrict__ dst_) {
    const int64_t n = (int64_t)B * Hd * Wd;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), b = (int)(i / ((int64_t)Wd * Hd));
        double M[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = Mfwd[b * 6 + k];
        {   // cv::warpAffine without WARP_INVERSE_MAP
            double D = M[0] * M[4] - M[1] * M[3];
            D = D != 0 ? 1. / D : 0;
            const double A11 = M[4] * D, A22 = M[0] * D;
            M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
            const double b1 = -M[0] * M[2] - M[1] * M[5];
            const double b2 = -M[3] * M[2] - M[4] * M[5];
            M[2] = b1; M[5] = b2;
        }
        const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
        const int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
        const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        const int X = (int)((unsigned)X0 + (unsigned)adelta) >> (AB_BITS - INTER_BITS);
        const int Y = (int)((unsigned)Y0 + (unsigned)bdelta) >> (AB_BITS - INTER_BITS);
        const int sx = max(-32768, min(32767, X >> INTER_BITS)), sy = max(-32768, min(32767, Y >> INTER_BITS));
        const int fx = X & (INTER_TAB_SIZE - 1), fy = Y & (INTER_TAB_SIZE - 1);
        int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
        if ((fx | fy) == 0) { w[0] = 32767; w[3] = 1; }        // BilinearTab_i[0]: 32768 saturates to short, the sum fix-up lands on the last tap
        const uint8_t *img = src + (size_t)b * Hs * Ws * C;
        const bool x0 = sx >= 0 && sx < Ws, x1 = sx + 1 >= 0 && sx + 1 < Ws, y0 = sy >= 0 && sy < Hs, y1 = sy + 1 >= 0 && sy + 1 < Hs;
        int v[4] = {0, 0, 0, 0};
        for (int c = 0; c < C && c < 4; ++c) {
            if (MODE == 2 && c != C - 1) continue;
            int acc = 0;
            if (x0 && y0) acc += img[((size_t)sy * Ws + sx) * C + c] * w[0];
            if (x1 && y0) acc += img[((size_t)sy * Ws + sx + 1) * C + c] * w[1];
            if (x0 && y1) acc += img[((size_t)(sy + 1) * Ws + sx) * C + c] * w[2];
            if (x1 && y1) acc += img[((size_t)(sy + 1) * Ws + sx + 1) * C + c] * w[3];
            v[c] = min(255, max(0, (acc + (1 << (REMAP_BITS - 1))) >> REMAP_BITS));
        }
        if (MODE == 0) {
            uint8_t *d = static_cast<uint8_t *>(dst_) + (size_t)i * C;
            for (int c = 0; c < C && c < 4; ++c) d[c] = (uint8_t)v[c];
        } else if (MODE == 1) {
            float *d = static_cast<float *>(dst_) + (size_t)b * 3 * Hd * Wd + (size_t)y * Wd + x;
#pragma unroll
            for (int c = 0; c < 3; ++c) d[(size_t)c * Hd * Wd] = (((float)v[2 - c] / 255.0f) - 0.5f) / 0.5f;
        } else {
            static_cast<float *>(dst_)[i] = (float)v[C - 1] / 128.0f;
        }
    }
}

// cv2.resize(src, (Wd, Hd)) with INTER_LINEAR for 8-bit images: uint8 [B][Hs][Ws][C] -> uint8 [B][Hd][Wd][C]
__global__ __launch_bounds__(NT) void resize_linear_u8_kernel(const uint8_t *__restrict__ src, int B, int Hs, int Ws, int C,
                                                              uint8_t *__restrict__ dst, int Hd, int Wd, double scale_x, double scale_y) {
    const int64_t n = (int64_t)B * Hd * Wd;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int dx = (int)(i % Wd), dy = (int)((i / Wd) % Hd), b = (int)(i / ((int64_t)Wd * Hd));
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= Ws - 1) { fx = 0; sx = Ws - 1; }
        const int a0 = max(-32768, min(32767, __float2int_rn((1.f - fx) * RS_SCALE))), a1 = max(-32768, min(32767, __float2int_rn(fx * RS_SCALE)));
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        const int b0 = max(-32768, min(32767, __float2int_rn((1.f - fy) * RS_SCALE))), b1 = max(-32768, min(32767, __float2int_rn(fy * RS_SCALE)));
        const int r0 = sy >= 0 ? (sy < Hs ? sy : Hs - 1) : 0, r1 = sy + 1 >= 0 ? (sy + 1 < Hs ? sy + 1 : Hs - 1) : 0;
        const int sx1 = min(sx + 1, Ws - 1);
        const uint8_t *img = src + (size_t)b * Hs * Ws * C;
        for (int c = 0; c < C; ++c) {
            const int S0 = img[((size_t)r0 * Ws + sx) * C + c] * a0 + img[((size_t)r0 * Ws + sx1) * C + c] * a1;
            const int S1 = img[((size_t)r1 * Ws + sx) * C + c] * a0 + img[((size_t)r1 * Ws + sx1) * C + c] * a1;
            dst[(size_t)i * C + c] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

}  // namespace

extern "C" int hoig_warp_affine_u8(const uint8_t *src, int B, int Hs, int Ws, int C, const double *M, int Hd, int Wd, int mode, void *dst,
                                   hoig_stream_t stream) {
    if (!src || !M || !dst || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C < 1 || C > 4) return HOIG_EINVAL;
    if (mode < 0 || mode > 2 || (mode == 1 && C != 3)) return HOIG_EINVAL;
    const int64_t n = (int64_t)B * Hd * Wd;
    const unsigned grid = hoig_stream_grid(n, NT);
    if (mode == 0) warp_affine_u8_kernel<0><<<grid, NT, 0, (hipStream_t)stream>>>(src, B, Hs, Ws, C, M, Hd, Wd, dst);
    else if (mode == 1) warp_affine_u8_kernel<1><<<grid, NT, 0, (hipStream_t)stream>>>(src, B, Hs, Ws, C, M, Hd, Wd, dst);
    else warp_affine_u8_kernel<2><<<grid, NT, 0, (hipStream_t)stream>>>(src, B, Hs, Ws, C, M, Hd, Wd, dst);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_resize_linear_u8(const uint8_t *src, int B, int Hs, int Ws, int C, uint8_t *dst, int Hd, int Wd, hoig_stream_t stream) {
    if (!src || !dst || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C < 1 || C > 4) return HOIG_EINVAL;
    if (Hs == Hd && Ws == Wd) {          // cv::resize copies
        if (hipMemcpyAsync(dst, src, (size_t)B * Hs * Ws * C, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return HOIG_ELAUNCH;
        return HOIG_OK;
    }
    const double scale_x = 1. / ((double)Wd / Ws), scale_y = 1. / ((double)Hd / Hs);
    const int64_t n = (int64_t)B * Hd * Wd;
    resize_linear_u8_kernel<<<hoig_stream_grid(n, NT), NT, 0, (hipStream_t)stream>>>(src, B, Hs, Ws, C, dst, Hd, Wd, scale_x, scale_y);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
