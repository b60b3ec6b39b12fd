/* CPU ORACLE (test infrastructure only) -- plain-C restatement of the four
 * CUDA kernels of the reference's two extensions, which cannot run without a
 * GPU and have no CPU path (block_extractor.py:23-24 raises NotImplementedError):
 *
 *   K1  kernel_block_extractor_update_output   thirdparty/block_extractor/block_extractor_kernel.cu:20-85
 *   K2  kernel_block_extractor_backward        thirdparty/block_extractor/block_extractor_kernel.cu:89-170
 *   K3  kernel_local_attn_reshape_update_output thirdparty/local_attn_reshape/local_attn_reshape_kernel.cu:20-61
 *   K4  kernel_local_attn_reshape_backward     thirdparty/local_attn_reshape/local_attn_reshape_kernel.cu:65-108
 *
 * One loop iteration == one CUDA thread of the reference; tensors are
 * contiguous NCHW as the wrappers assert (block_extractor.py:9-10).  The
 * caller zero-fills outputs (block_extractor.py:21,35-36).  Built by
 * oracle/Makefile into oracle/_build/libhoig_oracle_c.so and used by
 * tests/test_oracle_attn.py to cross-check oracle/hogan_oracle.py's torch
 * restatement of the same kernels (two independent restatements of the .cu).
 *
 * The reference's own .cu files are NOT buildable here (ATen + CUDA headers,
 * nvcc): see DESIGN.md.
 */
#include <math.h>
#include <stddef.h>

#define IDX4(b, c, y, x, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (y)) * (W) + (x))

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* K1 */
void oracle_block_extractor_forward_f64(const double *source, const double *flow, double *output,
                                        int B, int C, int Hs, int Ws, int Hf, int Wf, int k) {
    int H = k * Hf, W = k * Wf;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    int yf = y / k, xf = x / k;
                    int yo = y % k - k / 2, xo = x % k - k / 2;                 /* :57-60 */
                    double flow_y = flow[IDX4(b, 1, yf, xf, 2, Hf, Wf)] + yo;   /* :62 */
                    double flow_x = flow[IDX4(b, 0, yf, xf, 2, Hf, Wf)] + xo;   /* :63 */
                    double dy = flow_y + (double)yf, dx = flow_x + (double)xf;  /* :66-67 */
                    int xL = clampi((int)floor(dx), 0, Ws - 1), xR = clampi((int)floor(dx) + 1, 0, Ws - 1);
                    int yT = clampi((int)floor(dy), 0, Hs - 1), yB = clampi((int)floor(dy) + 1, 0, Hs - 1);
                    double xL_P = 1 - (dx - floor(dx)), xR_P = dx - floor(dx);
                    double yT_P = 1 - (dy - floor(dy)), yB_P = dy - floor(dy);
                    double s = 0.0;
                    s += xL_P * yT_P * source[IDX4(b, c, yT, xL, C, Hs, Ws)];   /* :79-82 */
                    s += xR_P * yT_P * source[IDX4(b, c, yT, xR, C, Hs, Ws)];
                    s += xL_P * yB_P * source[IDX4(b, c, yB, xL, C, Hs, Ws)];
                    s += xR_P * yB_P * source[IDX4(b, c, yB, xR, C, Hs, Ws)];
                    output[IDX4(b, c, y, x, C, H, W)] = s;
                }
}

/* K2 */
void oracle_block_extractor_backward_f64(const double *source, const double *flow, const double *grad_output,
                                         double *grad_source, double *grad_flow,
                                         int B, int C, int Hs, int Ws, int Hf, int Wf, int k) {
    int H = k * Hf, W = k * Wf;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    int yf = y / k, xf = x / k;
                    int yo = y % k - k / 2, xo = x % k - k / 2;
                    double flow_y = flow[IDX4(b, 1, yf, xf, 2, Hf, Wf)] + yo;
                    double flow_x = flow[IDX4(b, 0, yf, xf, 2, Hf, Wf)] + xo;
                    double dy = flow_y + (double)yf, dx = flow_x + (double)xf;
                    int xL = clampi((int)floor(dx), 0, Ws - 1), xR = clampi((int)floor(dx) + 1, 0, Ws - 1);
                    int yT = clampi((int)floor(dy), 0, Hs - 1), yB = clampi((int)floor(dy) + 1, 0, Hs - 1);
                    double xL_P = 1 - (dx - floor(dx)), xR_P = dx - floor(dx);
                    double yT_P = 1 - (dy - floor(dy)), yB_P = dy - floor(dy);
                    double vLT = source[IDX4(b, c, yT, xL, C, Hs, Ws)], vRT = source[IDX4(b, c, yT, xR, C, Hs, Ws)];
                    double vLB = source[IDX4(b, c, yB, xL, C, Hs, Ws)], vRB = source[IDX4(b, c, yB, xR, C, Hs, Ws)];
                    double g = grad_output[IDX4(b, c, y, x, C, H, W)];
                    grad_source[IDX4(b, c, yT, xL, C, Hs, Ws)] += g * xL_P * yT_P;   /* :158-161 */
                    grad_source[IDX4(b, c, yT, xR, C, Hs, Ws)] += g * xR_P * yT_P;
                    grad_source[IDX4(b, c, yB, xL, C, Hs, Ws)] += g * xL_P * yB_P;
                    grad_source[IDX4(b, c, yB, xR, C, Hs, Ws)] += g * xR_P * yB_P;
                    double gy = g * (-xL_P * vLT - xR_P * vRT + xL_P * vLB + xR_P * vRB);   /* :163 */
                    double gx = g * (-yT_P * vLT - yB_P * vLB + yT_P * vRT + yB_P * vRB);   /* :164 */
                    grad_flow[IDX4(b, 1, yf, xf, 2, Hf, Wf)] += gy;                      /* :167-168 */
                    grad_flow[IDX4(b, 0, yf, xf, 2, Hf, Wf)] += gx;
                }
}

/* K3 */
void oracle_local_attn_reshape_forward_f64(const double *inputs, double *output, int B, int Hs, int Ws, int k) {
    int H = k * Hs, W = k * Ws, C = k * k;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int cs = (y % k) * k + x % k;                                         /* :52-56 */
                output[IDX4(b, 0, y, x, 1, H, W)] = inputs[IDX4(b, cs, y / k, x / k, C, Hs, Ws)];
            }
}

/* K4 */
void oracle_local_attn_reshape_backward_f64(const double *grad_output, double *grad_inputs, int B, int Hs, int Ws, int k) {
    int H = k * Hs, W = k * Ws, C = k * k;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int cs = (y % k) * k + x % k;
                grad_inputs[IDX4(b, cs, y / k, x / k, C, Hs, Ws)] += grad_output[IDX4(b, 0, y, x, 1, H, W)];  /* :106 */
            }
}
