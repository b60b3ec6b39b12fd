// The MANO hand layer in front of HandRecoveryFlow: pose / shape parameters -> 778 skinned vertices.
//   reference call sites: HOIG_HOv3/models/networks/hmr.py:55,84-85 (smplx 0.1.28 MANO layer: full axis-angle pose, flat hand
//   mean, + transl) and HOIG_DexYCB/models/networks/hmr.py:55-60,85-86 (manopth ManoLayer: 45 PCA coefficients + hand mean,
//   + trans, x 1000 / 1000).  Both are the published linear blend skinning (smplx.lbs.lbs): shape blend, joint regression,
//   Rodrigues, pose blend, the 16-joint kinematic chain, blend of the joint transforms per vertex.
// One 256-thread workgroup per sample does all of it in one launch (the reference issues ~40 small torch kernels per call):
//   phase 1  16 threads: axis-angle (PCA expansion included) -> rotation matrices; 48 threads: the rest joints
//            J = J_template + J_shapedirs . betas (the regressor applied to the blend shapes ONCE on the host: it is linear)
//   phase 2  135 threads: pose feature (R - I of the 15 finger joints); one thread: the chain, 15 products of 3x4 transforms
//   phase 3  every thread, vertices tid, tid + 256, ..: shape blend (10 terms), pose blend (135 terms, coalesced rows of
//            posedirs), the blended transform (16 joints x 12) and the skinned position; 1.3 MB of model data per sample from L2.
// HBM-bound by nothing at this size (2.4 MB model, B <= 64): latency of three dependent phases, ~10 us per launch.
#include "common.h"

namespace {
constexpr int NJ = 16, NPF = 135, NB = 10, NHP = 45;

struct ManoArgs {
    const float *v_template, *shapedirs, *posedirs, *j_template, *j_shapedirs, *weights;
    const int *parents;
    const float *hands_mean, *comps;
    int ncomps, V;
    const float *root, *hand, *betas, *transl;
    float *verts, *joints;
    int ld_v;
};

__global__ __launch_bounds__(256) void mano_lbs_kernel(const ManoArgs p) {
    __shared__ float rot[NJ][9], J[NJ][3], G[NJ][12], A[NJ][12], pf[NPF], beta[NB], tr[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < NB) beta[tid] = p.betas[b * NB + tid];
    if (tid < 3) tr[tid] = p.transl ? p.transl[b * 3 + tid] : 0.f;
    __syncthreads();
    if (tid < NJ) {
        float r[3];
        if (tid == 0) {
            for (int k = 0; k < 3; ++k) r[k] = p.root[b * 3 + k];
        } else {
            for (int k = 0; k < 3; ++k) {
                const int e = (tid - 1) * 3 + k;
                float v;
                if (p.comps) {                       // manopth: hands_mean + coeffs @ components
                    v = 0.f;
                    for (int c = 0; c < p.ncomps; ++c) v += p.hand[b * p.ncomps + c] * p.comps[c * NHP + e];
                } else {
                    v = p.hand[b * NHP + e];
                }
                r[k] = v + (p.hands_mean ? p.hands_mean[e] : 0.f);
            }
        }
        // smplx.lbs.batch_rodrigues: angle = |r + 1e-8| (epsilon on every component), axis = r / angle
        const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float x = r[0] / angle, y = r[1] / angle, z = r[2] / angle;
        const float s = sinf(angle), c1 = 1.f - cosf(angle);
        // R = I + sin K + (1 - cos) K^2,  K = [[0,-z,y],[z,0,-x],[-y,x,0]]
        rot[tid][0] = 1.f + c1 * (-(y * y) - z * z); rot[tid][1] = -s * z + c1 * (x * y);       rot[tid][2] = s * y + c1 * (x * z);
        rot[tid][3] = s * z + c1 * (x * y);        rot[tid][4] = 1.f + c1 * (-(x * x) - z * z); rot[tid][5] = -s * x + c1 * (y * z);
        rot[tid][6] = -s * y + c1 * (x * z);       rot[tid][7] = s * x + c1 * (y * z);        rot[tid][8] = 1.f + c1 * (-(x * x) - y * y);
    } else if (tid >= 64 && tid < 64 + NJ * 3) {
        const int e = tid - 64;
        float v = p.j_template[e];
        for (int l = 0; l < NB; ++l) v += p.j_shapedirs[e * NB + l] * beta[l];
        J[e / 3][e % 3] = v;
    }
    __syncthreads();
    if (tid < NPF) {
        const int j = 1 + tid / 9, e = tid % 9;
        pf[tid] = rot[j][e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    if (tid == 255) {                                // smplx.lbs.batch_rigid_transform
        for (int i = 0; i < NJ; ++i) {
            const int par = p.parents[i];
            float t[3];
            for (int k = 0; k < 3; ++k) t[k] = J[i][k] - (i > 0 ? J[par][k] : 0.f);
            if (i == 0) {
                for (int r_ = 0; r_ < 3; ++r_) {
                    for (int c = 0; c < 3; ++c) G[0][r_ * 4 + c] = rot[0][r_ * 3 + c];
                    G[0][r_ * 4 + 3] = t[r_];
                }
            } else {
                for (int r_ = 0; r_ < 3; ++r_) {
                    const float g0 = G[par][r_ * 4], g1 = G[par][r_ * 4 + 1], g2 = G[par][r_ * 4 + 2], g3 = G[par][r_ * 4 + 3];
                    for (int c = 0; c < 3; ++c) G[i][r_ * 4 + c] = g0 * rot[i][c] + g1 * rot[i][3 + c] + g2 * rot[i][6 + c];
                    G[i][r_ * 4 + 3] = g0 * t[0] + g1 * t[1] + g2 * t[2] + g3;
                }
            }
        }
        for (int i = 0; i < NJ; ++i)                 // relative to the rest pose: A = [G_R | G_t - G_R J]
            for (int r_ = 0; r_ < 3; ++r_) {
                const float g0 = G[i][r_ * 4], g1 = G[i][r_ * 4 + 1], g2 = G[i][r_ * 4 + 2];
                A[i][r_ * 4] = g0; A[i][r_ * 4 + 1] = g1; A[i][r_ * 4 + 2] = g2;
                A[i][r_ * 4 + 3] = G[i][r_ * 4 + 3] - (g0 * J[i][0] + g1 * J[i][1] + g2 * J[i][2]);
            }
    }
    __syncthreads();
    if (p.joints && tid < NJ * 3) p.joints[(size_t)b * NJ * 3 + tid] = G[tid / 3][(tid % 3) * 4 + 3] + tr[tid % 3];
    for (int v = tid; v < p.V; v += 256) {
        float vp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = p.v_template[v * 3 + k];
            const float *sd = p.shapedirs + ((size_t)v * 3 + k) * NB;
#pragma unroll
            for (int l = 0; l < NB; ++l) a += sd[l] * beta[l];
            vp[k] = a;
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        const float *pd = p.posedirs + (size_t)v * 3;
        const size_t ldp = (size_t)p.V * 3;
        for (int q = 0; q < NPF; ++q) {
            const float f = pf[q];
            o0 += f * pd[q * ldp]; o1 += f * pd[q * ldp + 1]; o2 += f * pd[q * ldp + 2];
        }
        vp[0] += o0; vp[1] += o1; vp[2] += o2;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        const float *w = p.weights + (size_t)v * NJ;
        for (int j = 0; j < NJ; ++j) {
            const float wj = w[j];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] += wj * A[j][e];
        }
        float *out = p.verts + ((size_t)b * p.ld_v + v) * 3;
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_)
            out[r_] = T[r_ * 4] * vp[0] + T[r_ * 4 + 1] * vp[1] + T[r_ * 4 + 2] * vp[2] + T[r_ * 4 + 3] + tr[r_];
    }
}
}  // namespace

extern "C" int hoig_mano_lbs(const float *v_template, const float *shapedirs, const float *posedirs, const float *j_template,
                             const float *j_shapedirs, const float *lbs_weights, const int32_t *parents,
                             const float *hands_mean, const float *hands_components, int ncomps, int V, const float *root,
                             const float *hand, const float *betas, const float *transl, float *verts, int ld_v, float *joints,
                             int B, hoig_stream_t stream) {
    if (!v_template || !shapedirs || !posedirs || !j_template || !j_shapedirs || !lbs_weights || !parents || !root || !hand ||
        !betas || !verts)
        return HOIG_EINVAL;
    if (B <= 0 || V <= 0 || ld_v < V || (hands_components && (ncomps <= 0 || ncomps > NHP))) return HOIG_EINVAL;
    ManoArgs a{v_template, shapedirs, posedirs, j_template, j_shapedirs, lbs_weights, parents, hands_mean, hands_components,
               ncomps, V, root, hand, betas, transl, verts, joints, ld_v};
    mano_lbs_kernel<<<B, 256, 0, (hipStream_t)stream>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
