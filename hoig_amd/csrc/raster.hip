// Face-index / barycentric-weight rasteriser: the forward kernels behind MANORenderer.render_fim_wim (utils/nmr.py:496-513),
// i.e. neural_renderer's forward_face_index_map kernels 1 and 2 (thirdparty/neural_renderer/neural_renderer/cuda/
// rasterize_cuda_kernel.cu:40-186) with the wrapper's conventions (-1 / 0 fill, vertical flip: rasterize.py:50-52,334-338).
//
// The reference tests EVERY face at EVERY pixel (256 x 256 x ~15 k faces per view, twice per sample).  Here:
//   setup : one thread per face -- back-face test, the inverse edge matrix, and a conservative bounding box in 16 x 16-pixel
//           tiles (one pixel of slack on every side; faces with non-finite vertices get the whole image);
//   tiles : one 256-thread workgroup per tile and image.  It streams the packed boxes of all faces (4 B each, coalesced),
//           gathers the records of the faces that touch the tile into LDS (unordered, LDS atomic counter) and every thread
//           -- one pixel -- runs the reference's per-face test only over that list, reading records as LDS broadcasts.
// The reference keeps the FIRST face (in index order) among equal depths (`zp < depth_min`, strict); the unordered list
// reproduces that by breaking depth ties towards the smaller face index.  Per-pixel arithmetic is the reference's, literal
// for literal (its unsuffixed constants are doubles); the file is compiled with -ffp-contract=off (as the CPU
// restatement it is tested against), because an FMA in an edge function moves pixels across a triangle's border.
#include "common.h"
#pragma clang fp contract(off)

namespace {
constexpr int TILE = 16, NT = TILE * TILE, REC = 20, CAP = 512;

constexpr unsigned EMPTY_BOX = 0x00ff00ffu;            // tx0 = 255 > tx1 = 0: never overlaps

__device__ inline bool back_side(const float *f) {    // rasterize_cuda_kernel.cu:55,127
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

__global__ void raster_setup_kernel(const float *__restrict__ faces, int n_faces_total, int is, float *__restrict__ rec,
                                    unsigned *__restrict__ box) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n_faces_total) return;
    float f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = faces[(size_t)i * 9 + k];
    if (back_side(f)) {
        box[i] = EMPTY_BOX;
        return;
    }
    float p[3][2];                                     // :59-63: [-1,1] -> [0, is-1]
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int d = 0; d < 2; ++d) p[n][d] = (float)(0.5 * (f[3 * n + d] * is + is - 1));
    const float m[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                        p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                        p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
    const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
    float *r = rec + (size_t)i * REC;
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = f[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[9 + k] = m[k] / den;
    // bounding box in pixels (p is the pixel-space vertex), one pixel of slack, then in tiles
    const float lox = fminf(p[0][0], fminf(p[1][0], p[2][0])), hix = fmaxf(p[0][0], fmaxf(p[1][0], p[2][0]));
    const float loy = fminf(p[0][1], fminf(p[1][1], p[2][1])), hiy = fmaxf(p[0][1], fmaxf(p[1][1], p[2][1]));
    const int last = (is - 1) / TILE;
    unsigned b;
    const bool finite = isfinite(p[0][0]) && isfinite(p[1][0]) && isfinite(p[2][0]) && isfinite(p[0][1]) && isfinite(p[1][1]) &&
                        isfinite(p[2][1]);
    if (!finite) {
        b = 0u | ((unsigned)last << 8) | (0u << 16) | ((unsigned)last << 24);
    } else if (hix < -1.f || hiy < -1.f || lox > (float)is || loy > (float)is) {
        b = EMPTY_BOX;
    } else {
        const int x0 = max((int)floorf(fmaxf(lox, -1.f)) - 1, 0) / TILE, y0 = max((int)floorf(fmaxf(loy, -1.f)) - 1, 0) / TILE;
        const int x1 = min((int)ceilf(fminf(hix, (float)is)) + 1, is - 1) / TILE;
        const int y1 = min((int)ceilf(fminf(hiy, (float)is)) + 1, is - 1) / TILE;
        b = (unsigned)x0 | ((unsigned)x1 << 8) | ((unsigned)y0 << 16) | ((unsigned)y1 << 24);
    }
    box[i] = b;
}

__global__ __launch_bounds__(NT) void raster_tile_kernel(const float *__restrict__ rec, const unsigned *__restrict__ box, int F,
                                                        int is, float near, float far, int *__restrict__ fim,
                                                        float *__restrict__ wim) {
    __shared__ __attribute__((aligned(16))) float list[CAP * REC];
    __shared__ int cnt;
    const int tiles_x = (is + TILE - 1) / TILE;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
    const int xi = tx * TILE + (threadIdx.x & (TILE - 1)), yi = ty * TILE + (threadIdx.x >> 4);
    const float yp = (float)((2. * yi + 1 - is) / is), xp = (float)((2. * xi + 1 - is) / is);     // :112-113
    float zmin = far, wmin[3] = {0.f, 0.f, 0.f};
    int best = -1;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();

    auto process = [&](int n) {
        for (int e = 0; e < n; ++e) {
            const float *f = list + e * REC;           // every thread reads the same record: LDS broadcast
            if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
                ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
                ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                continue;                                                                           // :131-134
            float w[3], ws = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                w[k] = f[9 + 3 * k] * xi + f[9 + 3 * k + 1] * yi + f[9 + 3 * k + 2];               // :138-140
                w[k] = (float)fmin(fmax((double)w[k], 0.), 1.);                                    // :146
                ws += w[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) w[k] /= ws;
            const float zp = (float)(1. / (double)(w[0] / f[2] + w[1] / f[5] + w[2] / f[8]));      // :152
            if (zp <= near || far <= zp) continue;
            const int fn = __float_as_int(f[18]);
            if (zp < zmin || (zp == zmin && best >= 0 && fn < best)) {                              // :158 + index order
                zmin = zp;
                best = fn;
                wmin[0] = w[0]; wmin[1] = w[1]; wmin[2] = w[2];
            }
        }
    };

    const unsigned *bx = box + (size_t)b * F;
    const float *rc = rec + (size_t)b * F * REC;
    int total = 0;                                     // records in the list (uniform: from __syncthreads_count)
    for (int base = 0; base < F; base += NT) {
        const int fn = base + threadIdx.x;
        bool hit = false;
        if (fn < F) {
            const unsigned q = bx[fn];
            const int x0 = q & 255, x1 = (q >> 8) & 255, y0 = (q >> 16) & 255, y1 = q >> 24;
            hit = tx >= x0 && tx <= x1 && ty >= y0 && ty <= y1;
            if (hit) {
                const int slot = atomicAdd(&cnt, 1);
                const float4 *src = reinterpret_cast<const float4 *>(rc + (size_t)fn * REC);
                float4 *dst = reinterpret_cast<float4 *>(list + slot * REC);
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                float4 t = src[4];
                t.z = __int_as_float(fn);
                dst[4] = t;
            }
        }
        total += __syncthreads_count(hit);             // barrier: the records are in LDS
        if (total > CAP - NT) {                        // the next chunk might not fit: consume the list
            process(total);
            __syncthreads();
            if (threadIdx.x == 0) cnt = 0;
            total = 0;
            __syncthreads();
        }
    }
    process(total);
    if (xi < is && yi < is) {
        const size_t o = ((size_t)b * is + (is - 1 - yi)) * is + xi;          // vertical flip, rasterize.py:334-338
        fim[o] = best;
        wim[o * 3] = best >= 0 ? wmin[0] : 0.f;
        wim[o * 3 + 1] = best >= 0 ? wmin[1] : 0.f;
        wim[o * 3 + 2] = best >= 0 ? wmin[2] : 0.f;
    }
}

// ---- the vertex stage of render_fim_wim for a whole batch in ONE launch (round 6): projection (nmr.py:109-140 / the DexYCB copy's
// :146-163), y flip (:506), nr.look_at with the renderer's eye (a shift of z: the rotation is the identity) and nr.vertices_to_faces,
// for samples whose objects -- hence face lists -- differ: the lists' device addresses travel by value in the argument block.
// hoig_amd.raster.project_to_faces is the same arithmetic as ~20 batched torch ops per object group (it stays: the reference-shaped
// function, and the test fixture's subject); this file is compiled without FMA contraction, so sums are (a + b) + c as written.
struct ProjArgs {
    const float *cam, *verts;                  // [B][cam_dim], [B][V][3]
    const long long *faces[HOIG_PREP_MAX_BATCH];   // [F_b][3] int64
    int nf[HOIG_PREP_MAX_BATCH];
    int B, V, cam_dim, Fmax;
    float eye_z, pad;
    float *out;                                // [B][Fmax][3][3]
};
__global__ void project_faces_kernel(const ProjArgs a) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * NT + threadIdx.x;           // (face, corner)
    if (i >= a.Fmax * 3) return;
    const int f = i / 3;
    float *o = a.out + ((size_t)b * a.Fmax * 3 + i) * 3;
    if (f >= a.nf[b]) {                                    // padding face: all three vertices at one far-away point
        o[0] = a.pad; o[1] = a.pad; o[2] = a.pad;
        return;
    }
    const long long vid = a.faces[b][i];
    const float *v = a.verts + ((size_t)b * a.V + (size_t)vid) * 3;
    const float *c = a.cam + (size_t)b * a.cam_dim;
    float x, y, z;
    const float *tr;
    if (a.cam_dim == 15) {                                 // HOv3: OpenGL axis change, 3x3 camera matrix, perspective divide
        const float p0 = v[0], p1 = -v[1], p2 = -v[2];
        const float q0 = (p0 * c[0] + p1 * c[1]) + p2 * c[2];
        const float q1 = (p0 * c[3] + p1 * c[4]) + p2 * c[5];
        const float q2 = (p0 * c[6] + p1 * c[7]) + p2 * c[8];
        x = q0 / q2; y = q1 / q2; z = p2;
        tr = c + 9;
    } else {                                               // DexYCB: [fx, fy, cx, cy | 2x3 crop transform]
        x = v[0] / (v[2] + 1e-8f) * c[0] + c[2];
        y = v[1] / (v[2] + 1e-8f) * c[1] + c[3];
        z = v[2];
        tr = c + 4;
    }
    const float u = (tr[0] * x + tr[1] * y) + tr[2];       // the 2x3 crop transform on (x, y, 1)
    const float w = (tr[3] * x + tr[4] * y) + tr[5];
    o[0] = u / 255.0f * 2.f - 1.f;
    o[1] = -(w / 255.0f * 2.f - 1.f);                      // nmr.py:506
    o[2] = z - a.eye_z;                                    // look_at: vertices - eye
}
}  // namespace

extern "C" size_t hoig_rasterize_workspace_bytes(int B, int F) {
    return (size_t)B * F * (REC * sizeof(float) + sizeof(unsigned));
}

extern "C" int hoig_rasterize_fim_wim(const float *faces, int B, int F, int image_size, float near, float far, int32_t *fim,
                                      float *wim, void *workspace, hoig_stream_t stream) {
    if (!faces || !fim || !wim || !workspace || B <= 0 || F <= 0 || image_size <= 0 || image_size > 4096) return HOIG_EINVAL;
    float *rec = reinterpret_cast<float *>(workspace);
    unsigned *box = reinterpret_cast<unsigned *>(rec + (size_t)B * F * REC);
    hipStream_t st = (hipStream_t)stream;
    raster_setup_kernel<<<(B * F + NT - 1) / NT, NT, 0, st>>>(faces, B * F, image_size, rec, box);
    const int tiles = (image_size + TILE - 1) / TILE;
    raster_tile_kernel<<<dim3(tiles * tiles, B), NT, 0, st>>>(rec, box, F, image_size, near, far, fim, wim);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_project_faces(const float *cam, int cam_dim, const float *verts, int V, const int64_t *const *face_lists,
                                  const int *n_faces, int B, int Fmax, float eye_z, float pad_value, float *faces_out,
                                  hoig_stream_t stream) {
    if (B <= 0 || B > HOIG_PREP_MAX_BATCH) return HOIG_EUNSUPPORTED;
    if (!cam || !verts || !face_lists || !n_faces || !faces_out || V <= 0 || Fmax <= 0 || (cam_dim != 15 && cam_dim != 10)) return HOIG_EINVAL;
    ProjArgs a;
    a.cam = cam; a.verts = verts; a.B = B; a.V = V; a.cam_dim = cam_dim; a.Fmax = Fmax; a.eye_z = eye_z; a.pad = pad_value; a.out = faces_out;
    for (int b = 0; b < B; ++b) {
        if (!face_lists[b] || n_faces[b] < 0 || n_faces[b] > Fmax) return HOIG_EINVAL;
        a.faces[b] = reinterpret_cast<const long long *>(face_lists[b]);
        a.nf[b] = n_faces[b];
    }
    project_faces_kernel<<<dim3((Fmax * 3 + NT - 1) / NT, B), NT, 0, (hipStream_t)stream>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
