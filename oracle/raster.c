/* CPU ORACLE (test infrastructure only) -- plain-C restatement of the forward face-index / weight-map rasteriser of the
 * reference's vendored neural_renderer, the CUDA kernels behind MANORenderer.render_fim_wim (utils/nmr.py:496-513):
 *
 *   R1  forward_face_index_map_cuda_kernel_1   thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:40-84
 *   R2  forward_face_index_map_cuda_kernel_2   .../rasterize_cuda_kernel.cu:86-186
 *   and the conventions of the Python wrapper: outputs pre-filled with -1 / 0 (neural_renderer/rasterize.py:50-52) and the
 *   vertical flip of both maps (rasterize.py:334-338); anti_aliasing=False as nmr.py:512 passes.
 *
 * One loop iteration == one CUDA thread; scalar_t = float, and the literals the kernels write without a suffix (0.5, 2., 1.,
 * 0.) are doubles, so those sub-expressions are evaluated in double and rounded to float on assignment -- kept here.
 * PARITY UNPINNED: the CUDA kernels cannot run in this image and the reference's tests hold no expected maps (its
 * rasteriser tests write PNGs); in particular nvcc's default FMA contraction of `a*x + b*y + c` (:138-140) is not
 * reproduced (compiled with -ffp-contract=off, as the HIP kernel is).  tests/test_raster_oracle.py checks this file against
 * geometric ground truth (known triangles, exact barycentric weights, depth order, back-face and near/far culling).
 */
#include <stddef.h>
#include <stdint.h>

static float clamp01(float v) { return (float)(v < 0. ? 0. : (v > 1. ? 1. : (double)v)); }   /* min(max(w, 0.), 1.)  :146 */

/* faces [B][F][3][3] (x, y, z per vertex; x, y in [-1,1] normalised image coordinates, y up), outputs fim [B][S][S]
 * (int32, -1 = no face), wim [B][S][S][3]; scratch faces_inv [B][F][9] */
static void rasterize(const float *faces, int B, int F, int S, float near, float far, int32_t *fim, float *wim,
                      float *faces_inv, int wrapper) {
    const int is = S;
    for (size_t i = 0; i < (size_t)B * F * 9; ++i) faces_inv[i] = 0.f;               /* torch.zeros_like, rasterize.py:164 */
    for (int i = 0; i < B * F; ++i) {                                                 /* R1 */
        const float *f = faces + (size_t)i * 9;
        float *inv = faces_inv + (size_t)i * 9;
        if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) continue;  /* back side  :55 */
        float p[3][2];
        for (int n = 0; n < 3; ++n)
            for (int d = 0; d < 2; ++d) p[n][d] = (float)(0.5 * (f[3 * n + d] * is + is - 1));          /* :62 */
        float m[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                      p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                      p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};     /* :67-70 */
        const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
        for (int k = 0; k < 9; ++k) inv[k] = m[k] / den;                                                 /* :75-81 */
    }
    for (int b = 0; b < B; ++b)
        for (int yi = 0; yi < is; ++yi)
            for (int xi = 0; xi < is; ++xi) {                                         /* R2 */
                const float yp = (float)((2. * yi + 1 - is) / is), xp = (float)((2. * xi + 1 - is) / is);   /* :112-113 */
                float zmin = far, wmin[3] = {0.f, 0.f, 0.f};
                int best = -1;
                for (int fn = 0; fn < F; ++fn) {
                    const float *f = faces + ((size_t)b * F + fn) * 9;
                    const float *inv = faces_inv + ((size_t)b * F + fn) * 9;
                    if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) continue;          /* :127 */
                    if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
                        ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
                        ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                        continue;                                                                           /* :131-134 */
                    float w[3], ws = 0.f;
                    for (int k = 0; k < 3; ++k) {
                        w[k] = inv[3 * k] * xi + inv[3 * k + 1] * yi + inv[3 * k + 2];                     /* :138-140 */
                        w[k] = clamp01(w[k]);
                        ws += w[k];
                    }
                    for (int k = 0; k < 3; ++k) w[k] /= ws;                                                /* :148-150 */
                    const float zp = (float)(1. / (w[0] / f[2] + w[1] / f[5] + w[2] / f[8]));              /* :152 */
                    if (zp <= near || far <= zp) continue;                                                 /* :153 */
                    if (zp < zmin) {                                                                       /* :158: strict */
                        zmin = zp;
                        best = fn;
                        for (int k = 0; k < 3; ++k) wmin[k] = w[k];
                    }
                }
                if (!wrapper) {                    /* the bare kernel: writes covered pixels only, no flip  :173-178 */
                    const size_t o = ((size_t)b * is + yi) * is + xi;
                    if (best >= 0) {
                        fim[o] = best;
                        for (int k = 0; k < 3; ++k) wim[o * 3 + k] = wmin[k];
                    }
                    continue;
                }
                const size_t o = ((size_t)b * is + (is - 1 - yi)) * is + xi;          /* vertical flip, rasterize.py:334-338 */
                fim[o] = best;
                for (int k = 0; k < 3; ++k) wim[o * 3 + k] = best >= 0 ? wmin[k] : 0.f;
            }
}

/* what nr.rasterize_face_index_map_and_weight_map(faces, S, False) returns: kernels + the wrapper's fill and flip */
void oracle_rasterize_fim_wim(const float *faces, int B, int F, int S, float near, float far, int32_t *fim, float *wim,
                              float *faces_inv) {
    rasterize(faces, B, F, S, near, far, fim, wim, faces_inv, 1);
}

/* the two CUDA kernels alone (rasterize_cuda.forward_face_index_map): fim / wim pre-filled by the caller, no flip.  Used by
 * oracle/ref_harness.py to run the reference's own Python wrapper (neural_renderer/rasterize.py) on top of it. */
void oracle_rasterize_kernels(const float *faces, int B, int F, int S, float near, float far, int32_t *fim, float *wim,
                              float *faces_inv) {
    rasterize(faces, B, F, S, near, far, fim, wim, faces_inv, 0);
}
