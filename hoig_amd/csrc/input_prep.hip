// Input preparation AFTER the rasteriser: the tensor stage of HandRecoveryFlow.forward (HOIG_HOv3/models/trainer.py:46-145)
// and of the MANORenderer helpers it calls (utils/nmr.py:567-595 encode_fim / encode_sem, 874-968 cal_bc_transform,
// 973-1058 get_texture_backward_warp, 1068-1100 sample_from_texture_dense) and util.morph (utils/util.py:142-158).
// HBM-bound table lookups, 3-tap barycentric sums, two bilinear samplers and 3x3 / 15x15 erosions; one thread per pixel /
// texel, planar (NCHW, the reference's layout at this boundary) so that every store is coalesced along x.
// The reference hard-wires 256 x 256 images and a 256 x 640 texture atlas (nmr.py:975,1040-1051,1070): so does this file.
// The visibility test truncates a float to an integer (nmr.py:1012): the barycentric sums are written with explicit
// __fmul_rn / __fadd_rn in the reference's order (products, then (a0 + a1) + a2) so that no contraction moves a value across
// an integer boundary -- and the whole file is compiled with contraction OFF (hipcc's default -ffp-contract=fast would fuse
// those intrinsics, which are plain * and + in the HIP headers, into FMAs: one ulp of a texture coordinate is 1e-4 of a
// bilinear weight at atlas column 600).
#include "common.h"
#pragma clang fp contract(off)

namespace {
constexpr int NT = 256;
constexpr int S = 256, TW = 640, NHAND = 1538;         // image side, atlas width, hand faces (trainer.py:73)

// T = sum_k tbl[f][k] * w[k] (nmr.py:919 / 1004 / 1095); XY = 3: rows are (x, y, z) with y NEGATED (trainer.py:67-68);
// XY = 2: rows are (u, v)
template <int XY>
__device__ inline float2 bary(const float *__restrict__ tbl, int f, const float *__restrict__ w) {
    const float *r = tbl + (size_t)f * 3 * XY;
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float sy = XY == 3 ? -1.f : 1.f;
    float2 t;
    t.x = __fadd_rn(__fadd_rn(__fmul_rn(r[0], w0), __fmul_rn(r[XY], w1)), __fmul_rn(r[2 * XY], w2));
    t.y = __fadd_rn(__fadd_rn(__fmul_rn(sy * r[1], w0), __fmul_rn(sy * r[XY + 1], w1)), __fmul_rn(sy * r[2 * XY + 1], w2));
    return t;
}

// F.grid_sample(bilinear, zeros) of a planar [3][H][W] image at one grid point
__device__ inline void sample3(const float *__restrict__ img, int H, int W, float gx, float gy, bool align, float out[3]) {
    const float ix = align ? ((gx + 1.f) / 2.f) * (W - 1) : ((gx + 1.f) * W - 1.f) / 2.f;
    const float iy = align ? ((gy + 1.f) / 2.f) * (H - 1) : ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fmaxf(fminf(fx, 1e6f), -1e6f), y0 = (int)fmaxf(fminf(fy, 1e6f), -1e6f), x1 = x0 + 1, y1 = y0 + 1;
    const float nw = (fx + 1.f - ix) * (fy + 1.f - iy), ne = (ix - fx) * (fy + 1.f - iy);
    const float sw = (fx + 1.f - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
    const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *p = img + (size_t)c * H * W;
        float v = 0.f;
        if (y0ok && x0ok) v += p[y0 * W + x0] * nw;
        if (y0ok && x1ok) v += p[y0 * W + x1] * ne;
        if (y1ok && x0ok) v += p[y1 * W + x0] * sw;
        if (y1ok && x1ok) v += p[y1 * W + x1] * se;
        out[c] = v;
    }
}

// nmr.py:993-1046: per atlas texel, is the face it shows hidden in the source view?  occ = 1 - visible
__global__ void prep_occlusion_kernel(const float *__restrict__ faces, const int *__restrict__ src_fim,
                                      const int *__restrict__ fim_uv, const float *__restrict__ wim_uv,
                                      unsigned char *__restrict__ occ) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= S * TW) return;
    const int f = fim_uv[i];
    unsigned char o = 0;
    if (f != -1) {
        const float2 t = bary<3>(faces, f, wim_uv + (size_t)i * 3);
        // ((T + 1) / 2.0 * 255.0).long().clamp(0, 255)   (nmr.py:1012; .long() truncates toward zero)
        const float px = __fmul_rn(__fadd_rn(t.x, 1.f) / 2.f, 255.f), py = __fmul_rn(__fadd_rn(t.y, 1.f) / 2.f, 255.f);
        const int cx = min(max((int)px, 0), 255), cy = min(max((int)py, 0), 255);
        bool vis = false;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int qx = min(max(cx + dx, 0), 255), qy = min(max(cy + dy, 0), 255);
                vis |= src_fim[qy * 256 + qx] == f;
            }
        o = vis ? 0 : 1;
    }
    occ[i] = o;
}

// nmr.py:1048-1056: sample the source image into the atlas, open the occlusion mask (erode 3x3 then dilate 3x3, util.morph
// padding: outside counts as occluded for the erosion and as free for the dilation), paint occluded texels 1.0, and put the
// object's own texture image into columns 384..639
__global__ void prep_texture_kernel(const float *__restrict__ im, const float *__restrict__ faces,
                                    const int *__restrict__ fim_uv, const float *__restrict__ wim_uv,
                                    const unsigned char *__restrict__ occ, const float *__restrict__ obj_tex,
                                    float *__restrict__ tex) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= S * TW) return;
    const int y = i / TW, x = i - y * TW;
    if (x >= 384) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tex[(size_t)c * S * TW + i] = obj_tex[((size_t)y * 256 + (x - 384)) * 3 + c];
        return;
    }
    bool o2 = false;                                       // dilate(erode(occ))
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int qy = y + dy, qx = x + dx;
            if (qy < 0 || qy >= S || qx < 0 || qx >= TW) continue;
            bool e = true;
            for (int ey = -1; ey <= 1; ++ey)
                for (int ex = -1; ex <= 1; ++ex) {
                    const int ry = qy + ey, rx = qx + ex;
                    if (ry < 0 || ry >= S || rx < 0 || rx >= TW) continue;
                    e &= occ[ry * TW + rx] != 0;
                }
            o2 |= e;
        }
    float v[3] = {1.f, 1.f, 1.f};
    if (!o2) {
        const int f = fim_uv[i];
        float2 t = make_float2(-2.f, -2.f);
        if (f != -1) t = bary<3>(faces, f, wim_uv + (size_t)i * 3);
        sample3(im, S, S, t.x, t.y, false, v);            // F.grid_sample default: align_corners=False (nmr.py:1048)
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) tex[(size_t)c * S * TW + i] = v[c];
}

// trainer.py:66-88 for one sample and one view: table lookups, hand-region mask, texture read-back; for the reference view
// also the source->reference flow T of cal_bc_transform
__global__ void prep_lookup_kernel(const int *__restrict__ fim, const float *__restrict__ wim, const float *__restrict__ map_fn,
                                   const float *__restrict__ sem, const float *__restrict__ uv_coord, int n_faces,
                                   const float *__restrict__ tex, const float *__restrict__ src_faces,
                                   float *__restrict__ cond, float *__restrict__ seg, float *__restrict__ hand_region,
                                   float *__restrict__ rend, float *__restrict__ T) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= S * S) return;
    const int f = fim[i];
    const int row = f < 0 ? n_faces : f;                   // index -1 = the last (background) row, nmr.py:576,590
    cond[i] = map_fn[row * 3];
    cond[S * S + i] = map_fn[row * 3 + 1];
    cond[2 * S * S + i] = map_fn[row * 3 + 2];
    seg[i] = sem[row];
    hand_region[i] = (f != -1 && f < NHAND) ? 0.f : 1.f;   // 1 - hand faces (trainer.py:73, before the erosion)
    float2 tt = make_float2(-2.f, -2.f);
    if (f != -1) tt = bary<2>(uv_coord, f, wim + (size_t)i * 3);
    float v[3];
    sample3(tex, S, TW, tt.x, tt.y, true, v);             // trainer.py:86,88: align_corners=True
    rend[i] = v[0];
    rend[S * S + i] = v[1];
    rend[2 * S * S + i] = v[2];
    if (T) {
        float2 t = make_float2(-2.f, -2.f);
        if (f != -1) t = bary<3>(src_faces, f, wim + (size_t)i * 3);
        T[i * 2] = t.x;
        T[i * 2 + 1] = t.y;
    }
}


// ---- batched forms (round 6): one launch for the whole batch instead of one per sample.  The per-sample object tables differ (every
// sample has its own object), so their device addresses travel BY VALUE in the argument block (B <= HOIG_PREP_MAX_BATCH); blockIdx.y is
// the sample.  Same arithmetic as the per-sample kernels above, which stay (they are the C ABI of one sample).
constexpr int MAXB = HOIG_PREP_MAX_BATCH;
struct TexBatch {
    const float *src_img, *src_faces;          // [B,3,S,S], [B,Fstride/9,3,3]
    const int *src_fim;                        // [B,S,S]
    const int *fim_uv[MAXB];
    const float *wim_uv[MAXB], *obj_tex[MAXB];
    unsigned char *occ;                        // [B, S*TW]
    float *tex;                                // [B,3,S,TW]
    long long face_stride;                     // floats between two samples' face tensors
};
__global__ void prep_occlusion_batched_kernel(const TexBatch a) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= S * TW) return;
    const float *faces = a.src_faces + (size_t)b * a.face_stride;
    const int *src_fim = a.src_fim + (size_t)b * S * S;
    const int f = a.fim_uv[b][i];
    unsigned char o = 0;
    if (f != -1) {
        const float2 t = bary<3>(faces, f, a.wim_uv[b] + (size_t)i * 3);
        const float px = __fmul_rn(__fadd_rn(t.x, 1.f) / 2.f, 255.f), py = __fmul_rn(__fadd_rn(t.y, 1.f) / 2.f, 255.f);
        const int cx = min(max((int)px, 0), 255), cy = min(max((int)py, 0), 255);
        bool vis = false;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int qx = min(max(cx + dx, 0), 255), qy = min(max(cy + dy, 0), 255);
                vis |= src_fim[qy * 256 + qx] == f;
            }
        o = vis ? 0 : 1;
    }
    a.occ[(size_t)b * S * TW + i] = o;
}
__global__ void prep_texture_batched_kernel(const TexBatch a) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= S * TW) return;
    const int y = i / TW, x = i - y * TW;
    float *tex = a.tex + (size_t)b * 3 * S * TW;
    if (x >= 384) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tex[(size_t)c * S * TW + i] = a.obj_tex[b][((size_t)y * 256 + (x - 384)) * 3 + c];
        return;
    }
    const unsigned char *occ = a.occ + (size_t)b * S * TW;
    bool o2 = false;                                       // dilate(erode(occ))
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int qy = y + dy, qx = x + dx;
            if (qy < 0 || qy >= S || qx < 0 || qx >= TW) continue;
            bool e = true;
            for (int ey = -1; ey <= 1; ++ey)
                for (int ex = -1; ex <= 1; ++ex) {
                    const int ry = qy + ey, rx = qx + ex;
                    if (ry < 0 || ry >= S || rx < 0 || rx >= TW) continue;
                    e &= occ[ry * TW + rx] != 0;
                }
            o2 |= e;
        }
    float v[3] = {1.f, 1.f, 1.f};
    if (!o2) {
        const int f = a.fim_uv[b][i];
        float2 t = make_float2(-2.f, -2.f);
        if (f != -1) t = bary<3>(a.src_faces + (size_t)b * a.face_stride, f, a.wim_uv[b] + (size_t)i * 3);
        sample3(a.src_img + (size_t)b * 3 * S * S, S, S, t.x, t.y, false, v);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) tex[(size_t)c * S * TW + i] = v[c];
}

struct LookupBatch {
    const int *fim;                            // [B,S,S]
    const float *wim;                          // [B,S,S,3]
    const float *map_fn[MAXB], *sem[MAXB], *uv_coord[MAXB];
    int n_faces[MAXB];
    const float *tex, *src_faces;              // [B,3,S,TW], [B,.,3,3]
    long long face_stride;
    float *cond, *seg, *hand_region, *rend, *T;    // [B,3,S,S], [B,S,S], [B,S,S], [B,3,S,S], [B,S,S,2] (nullable)
};
__global__ void prep_lookup_batched_kernel(const LookupBatch a) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= S * S) return;
    const size_t P = (size_t)S * S;
    const int f = a.fim[b * P + i];
    const int row = f < 0 ? a.n_faces[b] : f;
    const float *map_fn = a.map_fn[b];
    float *cond = a.cond + b * 3 * P;
    cond[i] = map_fn[row * 3];
    cond[P + i] = map_fn[row * 3 + 1];
    cond[2 * P + i] = map_fn[row * 3 + 2];
    a.seg[b * P + i] = a.sem[b][row];
    a.hand_region[b * P + i] = (f != -1 && f < NHAND) ? 0.f : 1.f;
    const float *w = a.wim + (b * P + i) * 3;
    float2 tt = make_float2(-2.f, -2.f);
    if (f != -1) tt = bary<2>(a.uv_coord[b], f, w);
    float v[3];
    sample3(a.tex + (size_t)b * 3 * S * TW, S, TW, tt.x, tt.y, true, v);
    float *rend = a.rend + b * 3 * P;
    rend[i] = v[0];
    rend[P + i] = v[1];
    rend[2 * P + i] = v[2];
    if (a.T) {
        float2 t = make_float2(-2.f, -2.f);
        if (f != -1) t = bary<3>(a.src_faces + (size_t)b * a.face_stride, f, w);
        a.T[(b * P + i) * 2] = t.x;
        a.T[(b * P + i) * 2 + 1] = t.y;
    }
}

// util.morph(mode='erode'): sum over the ks x ks window with the outside counted as 1, == ks * ks
__device__ inline float erode(const float *__restrict__ m, int y, int x, int r) {
    float s = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const int qy = y + dy, qx = x + dx;
            s += (qy < 0 || qy >= S || qx < 0 || qx >= S) ? 1.f : m[qy * S + qx];
        }
    const int ks = 2 * r + 1;
    return s == (float)(ks * ks) ? 1.f : 0.f;
}

struct AsmArgs {
    const float *src_img, *ref_img;                       // [B,3,S,S]
    const float *cond_s, *cond_r;                         // [B,3,S,S]
    const float *seg_s, *seg_r, *hr_s, *hr_r;             // [B,S,S]
    const float *rend_s, *rend_r;                         // [B,3,S,S]
    const float *T_raw;                                   // [B,S,S,2]
    float *src_bg, *tsf_bg;                               // [B,4,S,S] (tsf_bg nullable)
    float *src_obj, *tsf_obj;                             // [B,15,S,S]
    float *src_hand, *ref_hand;                           // [B,hand_c,S,S]: 6 (HOv3) or 12 (DexYCB: + six hand-part one-hots)
    float *T_hand;                                        // [B,S,S,2]
    float *smb, *rmb, *smh, *rmh;                         // [B,1,S,S]
    int B, hand_c;
};

// trainer.py:110-141
__global__ void prep_assemble_kernel(const AsmArgs a) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (int64_t)a.B * S * S) return;
    const int b = (int)(i / (S * S)), p = (int)(i - (int64_t)b * S * S), y = p / S, x = p - y * S;
    const size_t P = (size_t)S * S;
#pragma unroll
    for (int view = 0; view < 2; ++view) {
        const float *cond = (view ? a.cond_r : a.cond_s) + b * 3 * P;
        const float *hr = (view ? a.hr_r : a.hr_s) + b * P;
        const float *rend = (view ? a.rend_r : a.rend_s) + b * 3 * P;
        const float seg = (view ? a.seg_r : a.seg_s)[b * P + p];
        const float *img = (view ? a.ref_img : a.src_img) + b * 3 * P;
        const float mh = erode(hr, y, x, 1);                                     // crop_mask_hand (trainer.py:73,79)
        const float mb = erode(cond + 2 * P, y, x, 1);                           // crop_mask_bg   (trainer.py:110-111)
        (view ? a.rmh : a.smh)[b * P + p] = mh;
        (view ? a.rmb : a.smb)[b * P + p] = mb;
        const float u = cond[p], v = cond[P + p], flag = cond[2 * P + p];
        const float hm = u < 1.5f ? 1.f : 0.f, om = u > 1.5f ? 1.f : 0.f;       // trainer.py:113-125
        float *obj = (view ? a.tsf_obj : a.src_obj) + b * 15 * P + p;
        float *hand = (view ? a.ref_hand : a.src_hand) + (size_t)b * a.hand_c * P + p;
        const float r0 = rend[p], r1 = rend[P + p], r2 = rend[2 * P + p];
        const float mo = mh - mb;                                                // trainer.py:128,132
        obj[0] = r0 * mo; obj[P] = r1 * mo; obj[2 * P] = r2 * mo;
        obj[3 * P] = om * u; obj[4 * P] = om * v; obj[5 * P] = flag + 1.f - om;
#pragma unroll
        for (int j = 0; j < 9; ++j) obj[(6 + j) * P] = seg == (float)(j + 7) ? 1.f : 0.f;     // seg[:, 6:] of labels 1..15
        const float nh = 1.f - mh;                                               // trainer.py:129,133
        if (view == 0) { hand[0] = img[p] * nh; hand[P] = img[P + p] * nh; hand[2 * P] = img[2 * P + p] * nh; }
        else { hand[0] = r0 * nh; hand[P] = r1 * nh; hand[2 * P] = r2 * nh; }
        hand[3 * P] = hm * u; hand[4 * P] = hm * v; hand[5 * P] = flag + 1.f - hm;
        if (a.hand_c == 12) {                                                    // HOIG_DexYCB/models/trainer.py:131,135: seg[:, :6]
#pragma unroll
            for (int j = 0; j < 6; ++j) hand[(6 + j) * P] = seg == (float)(j + 1) ? 1.f : 0.f;
        }
        float *bg = view ? a.tsf_bg : a.src_bg;
        if (bg) {                                                                // trainer.py:136-141
            const float e = erode(cond + 2 * P, y, x, 7);
            bg += b * 4 * P + p;
            bg[0] = img[p] * e; bg[P] = img[P + p] * e; bg[2 * P] = img[2 * P + p] * e; bg[3 * P] = e;
        }
        if (view == 1) {                                                         // trainer.py:82
            const float tx = a.T_raw[i * 2], ty = a.T_raw[i * 2 + 1];
            a.T_hand[i * 2] = tx * (mh == 0.f ? 1.f : 0.f) + -2.f * (mh == 1.f ? 1.f : 0.f);
            a.T_hand[i * 2 + 1] = ty * (mh == 0.f ? 1.f : 0.f) + -2.f * (mh == 1.f ? 1.f : 0.f);
        }
    }
}
}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hoig_prep_texture(const float *src_img, const float *src_faces, const int32_t *src_fim, const int32_t *fim_uv,
                                 const float *wim_uv, const float *obj_tex_img, unsigned char *occ_ws, float *tex,
                                 hoig_stream_t stream) {
    if (!src_img || !src_faces || !src_fim || !fim_uv || !wim_uv || !obj_tex_img || !occ_ws || !tex) return HOIG_EINVAL;
    const int grid = (S * TW + NT - 1) / NT;
    prep_occlusion_kernel<<<grid, NT, 0, ST>>>(src_faces, src_fim, fim_uv, wim_uv, occ_ws);
    prep_texture_kernel<<<grid, NT, 0, ST>>>(src_img, src_faces, fim_uv, wim_uv, occ_ws, obj_tex_img, tex);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_prep_lookup(const int32_t *fim, const float *wim, const float *map_fn, const float *sem_full,
                                const float *faces_uv_coord, int n_faces, const float *tex, const float *src_faces,
                                float *cond, float *seg, float *hand_region, float *rend, float *T, hoig_stream_t stream) {
    if (!fim || !wim || !map_fn || !sem_full || !faces_uv_coord || n_faces <= 0 || !tex || !cond || !seg || !hand_region ||
        !rend || (T && !src_faces))
        return HOIG_EINVAL;
    prep_lookup_kernel<<<(S * S + NT - 1) / NT, NT, 0, ST>>>(fim, wim, map_fn, sem_full, faces_uv_coord, n_faces, tex, src_faces,
                                                             cond, seg, hand_region, rend, T);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_prep_texture_batched(int B, const float *src_img, const float *src_faces, int64_t face_stride, const int32_t *src_fim,
                                         const int32_t *const *fim_uv, const float *const *wim_uv, const float *const *obj_tex_img,
                                         unsigned char *occ_ws, float *tex, hoig_stream_t stream) {
    if (B <= 0 || B > HOIG_PREP_MAX_BATCH) return HOIG_EUNSUPPORTED;
    if (!src_img || !src_faces || !src_fim || !fim_uv || !wim_uv || !obj_tex_img || !occ_ws || !tex || face_stride <= 0) return HOIG_EINVAL;
    TexBatch a;
    a.src_img = src_img; a.src_faces = src_faces; a.src_fim = src_fim; a.occ = occ_ws; a.tex = tex; a.face_stride = face_stride;
    for (int b = 0; b < B; ++b) {
        if (!fim_uv[b] || !wim_uv[b] || !obj_tex_img[b]) return HOIG_EINVAL;
        a.fim_uv[b] = fim_uv[b]; a.wim_uv[b] = wim_uv[b]; a.obj_tex[b] = obj_tex_img[b];
    }
    const dim3 grid((S * TW + NT - 1) / NT, B);
    prep_occlusion_batched_kernel<<<grid, NT, 0, ST>>>(a);
    prep_texture_batched_kernel<<<grid, NT, 0, ST>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_prep_lookup_batched(int B, const int32_t *fim, const float *wim, const float *const *map_fn, const float *const *sem_full,
                                        const float *const *faces_uv_coord, const int *n_faces, const float *tex, const float *src_faces,
                                        int64_t face_stride, float *cond, float *seg, float *hand_region, float *rend, float *T,
                                        hoig_stream_t stream) {
    if (B <= 0 || B > HOIG_PREP_MAX_BATCH) return HOIG_EUNSUPPORTED;
    if (!fim || !wim || !map_fn || !sem_full || !faces_uv_coord || !n_faces || !tex || !cond || !seg || !hand_region || !rend ||
        (T && (!src_faces || face_stride <= 0)))
        return HOIG_EINVAL;
    LookupBatch a;
    a.fim = fim; a.wim = wim; a.tex = tex; a.src_faces = src_faces; a.face_stride = face_stride;
    a.cond = cond; a.seg = seg; a.hand_region = hand_region; a.rend = rend; a.T = T;
    for (int b = 0; b < B; ++b) {
        if (!map_fn[b] || !sem_full[b] || !faces_uv_coord[b] || n_faces[b] <= 0) return HOIG_EINVAL;
        a.map_fn[b] = map_fn[b]; a.sem[b] = sem_full[b]; a.uv_coord[b] = faces_uv_coord[b]; a.n_faces[b] = n_faces[b];
    }
    prep_lookup_batched_kernel<<<dim3((S * S + NT - 1) / NT, B), NT, 0, ST>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_prep_assemble(int B, const float *src_img, const float *ref_img, const float *cond_s, const float *cond_r,
                                  const float *seg_s, const float *seg_r, const float *hr_s, const float *hr_r,
                                  const float *rend_s, const float *rend_r, const float *T_raw, float *src_bg, float *tsf_bg,
                                  float *src_obj, float *tsf_obj, float *src_hand, float *ref_hand, int hand_channels,
                                  float *T_hand, float *smb, float *rmb, float *smh, float *rmh, hoig_stream_t stream) {
    if (hand_channels != 6 && hand_channels != 12) return HOIG_EINVAL;
    if (B <= 0 || !src_img || !ref_img || !cond_s || !cond_r || !seg_s || !seg_r || !hr_s || !hr_r || !rend_s || !rend_r ||
        !T_raw || !src_bg || !src_obj || !tsf_obj || !src_hand || !ref_hand || !T_hand || !smb || !rmb || !smh || !rmh)
        return HOIG_EINVAL;
    AsmArgs a{src_img, ref_img, cond_s, cond_r, seg_s, seg_r, hr_s, hr_r, rend_s, rend_r, T_raw, src_bg, tsf_bg, src_obj,
              tsf_obj, src_hand, ref_hand, T_hand, smb, rmb, smh, rmh, B, hand_channels};
    prep_assemble_kernel<<<(int)(((int64_t)B * S * S + NT - 1) / NT), NT, 0, ST>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
