// Weight gradient of the attention's VALID 5x5 convolutions (models/networks/extract_attn.py:18) on a FLATTENED pixel axis.
//
// dW[co][(r,s)][ci] = sum over output positions of dy[pos][co] * x[pos + (r,s)][ci].  wgrad_halo_bf16_kernel (conv_igemm_bf16.hip)
// does this on 2 x 32-pixel tiles, which needs an output width that is a multiple of 32: the source-side convolution's 36-wide
// output fell to the generic kernel (160 TFLOP/s: it re-gathers x for each of the 25 taps), and a 64-pixel tile pays a 6 x 36
// halo for 2 x 32 outputs.  Here the batch is ONE sequence of canvas positions q = (b, y, x) over the INPUT grid (pitch Wc = Wi):
// dy is placed on the canvas (zero where y >= Ho or x >= Wo), tap (r, s) reads x at q + r * Wc + s, a pixel tile is 128
// consecutive positions and its x halo 128 + 4 * (Wc + 1) positions (1.3x instead of 3.4x).  Everything else is the halo kernel's:
// a workgroup owns dW[64 co][25 taps][32 ci], ten waves = (32-channel group of co) x (tap row) with five accumulators each, both
// operands staged as they arrive ([position][channel], split to bf16) and read with ds_read_b64_tr_b16 (x rows 64 B apart: the
// four rows a 32-lane half reads are 256 contiguous bytes at every tap offset), pixel tiles split over blockIdx.y, fp32 atomics.
#include "conv_bf16_common.h"

namespace hoig_detail {
namespace {

typedef short s4_t __attribute__((ext_vector_type(4)));
constexpr int KS = 5, PTW = 128, BM = 64, BC = 32, NT = 640;
constexpr int PSTR = 192, QSTR = 64;                       // dy / x row strides in LDS (wgrad_halo_bf16_kernel)
constexpr int PSL = (PTW * 16 + NT - 1) / NT;              // dy float4s per thread (16 per position)
constexpr int QSL = 5;                                     // x float4s per thread: halo positions * 8 <= QSL * NT

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *a, int stride4) {
    const s4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)a);
    const s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a + stride4));
    bf16x8 f;
    f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
    f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
    return f;
}

template <int NSX>
__global__ __launch_bounds__(NT) void wgrad_flat_kernel(const WFlatArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: dy, x
    constexpr int PLANE_P = PTW * PSTR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HPOS = p.HPOS;                               // halo positions of a tile: PTW + (KS-1) * (Wc + 1)
    const int PLANE_Q = (HPOS * QSTR + 255) / 256 * 256;
    unsigned char *Ph = smem, *Pl = smem + PLANE_P;
    unsigned char *Qh = smem + NS * PLANE_P, *Ql = Qh + PLANE_Q;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 1, tr = wave >> 1;               // 32-channel group of co, tap row
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int c0 = (tile / p.nblk_ci) * BM, ci0 = (tile % p.nblk_ci) * BC;
    const int mt_begin = blockIdx.y * p.mt_per_split;
    const int mt_end = min(p.n_mtiles, mt_begin + p.mt_per_split);

    float4 rp[PSL], rq[QSL];
    const bool do_bias = p.DB != nullptr && ci0 == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tiles = [&](int mt) {
        const int q0 = mt * PTW;
#pragma unroll
        for (int i = 0; i < PSL; ++i) {
            const int idx = tid + NT * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < PTW * 16) {
                const int q = q0 + (idx >> 4), c4 = idx & 15;
                if (q < p.Q) {
                    const int b = q / p.HWc, rem = q - b * p.HWc;
                    const int y = rem / p.Wc, x = rem - y * p.Wc;
                    if (y < p.Ho && x < p.Wo)
                        v = *reinterpret_cast<const float4 *>(p.DY + ((size_t)(b * p.Ho + y) * p.Wo + x) * p.Co + c0 + c4 * 4);
                }
            }
            rp[i] = v;       // (the bias sum takes it in store_tiles: wgrad_igemm_bf16.hip, load_tiles)
        }
#pragma unroll
        for (int i = 0; i < QSL; ++i) {
            const int idx = tid + NT * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int q = q0 + (idx >> 3);
            if ((idx >> 3) < HPOS && q < p.Q) v = *reinterpret_cast<const float4 *>(p.X + (size_t)q * p.Ci + ci0 + (idx & 7) * 4);
            rq[i] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < PSL; ++i) {
            const int idx = tid + NT * i;
            if (do_bias) {                 // this thread always holds the same four channels (NT % 16 == 0); idle slices hold zeros
                bsum.x += rp[i].x; bsum.y += rp[i].y; bsum.z += rp[i].z; bsum.w += rp[i].w;
            }
            if (idx < PTW * 16) {
                uint2 hi, lo;
                split4(rp[i], hi, lo);
                const int off = (idx >> 4) * PSTR + (idx & 15) * 8;
                *reinterpret_cast<uint2 *>(Ph + off) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < QSL; ++i) {
            const int idx = tid + NT * i;
            if ((idx >> 3) < HPOS) {
                uint2 hi, lo;
                split4(rq[i], hi, lo);
                *reinterpret_cast<uint2 *>(Qh + idx * 8) = hi;
                if (NB == 2) *reinterpret_cast<uint2 *>(Ql + idx * 8) = lo;
            }
        }
    };

    // transpose-read addressing (wgrad_bf16_kernel): 16-lane group g, lane 4q+c -> row 8*(g>>1)+q, channels 16*(g&1)+4c
    const int grp = lane >> 4, li = lane & 15;
    const int trow = (grp >> 1) * 8 + (li >> 2), tch = ((grp & 1) * 16 + (li & 3) * 4) * 2;
    const int trP = trow * PSTR + tch + cb * 64, trQ = (trow + tr * p.Wc) * QSTR + tch;

    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (mt_begin < mt_end) {
        load_tiles(mt_begin);
        store_tiles();
    }
    __syncthreads();
    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const bool nxt = mt + 1 < mt_end;
        if (nxt) load_tiles(mt + 1);
#pragma unroll
        for (int kk = 0; kk < PTW / 16; ++kk) {            // 16 consecutive positions per k-step
            const bf16x8 ah = tr_frag(Ph + trP + kk * 16 * PSTR, 4 * PSTR);
            bf16x8 al;
            if (NS == 2) al = tr_frag(Pl + trP + kk * 16 * PSTR, 4 * PSTR);
            bf16x8 bh[KS], bl[KS];
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                const int qoff = trQ + (kk * 16 + t) * QSTR;
                bh[t] = tr_frag(Qh + qoff, 4 * QSTR);
                if (NB == 2) bl[t] = tr_frag(Ql + qoff, 4 * QSTR);
            }
            // term-major: the KS accumulators take turns, so no MFMA waits on the one issued just before it
            if (NS == 2) {
#pragma unroll
                for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
            }
            if (NB == 2) {
#pragma unroll
                for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
        }
        __syncthreads();                      // every wave is done reading the stage
        if (nxt) store_tiles();
        __syncthreads();
    }

    if (do_bias) {                         // the threads that hold partial sums of the same four channels combine in LDS
        float *red = reinterpret_cast<float *>(smem);          // (the tiles are dead: the loop ended with a barrier)
        if (tid < BM) red[tid] = 0.f;
        __syncthreads();
        const int ch = (tid & 15) * 4;
        atomicAdd(&red[ch + 0], bsum.x);
        atomicAdd(&red[ch + 1], bsum.y);
        atomicAdd(&red[ch + 2], bsum.z);
        atomicAdd(&red[ch + 3], bsum.w);
        __syncthreads();
        if (tid < BM) atomicAdd(&p.DB[c0 + tid], red[tid]);
    }
    const int l31 = lane & 31, lh = lane >> 5;
    const int K = KS * KS * p.Ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = c0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float *row = p.DW + (size_t)co * K + (tr * KS) * p.Ci + ci0 + l31;
#pragma unroll
        for (int t = 0; t < KS; ++t) atomicAdd(row + t * p.Ci, acc[t][r]);
    }
}

}  // namespace

// dW (and the bias gradient) of a valid 5x5 stride-1 convolution: x [Bn][Hi][Wi][Ci], dy [Bn][Hi-4][Wi-4][Co]
int launch_wgrad_flat5(const float *x, const float *dy, float *dw, float *dbias, int Bn, int Hi, int Wi, int Ci, int Co, int ns,
                       hipStream_t st) {
    if (Co % 64 || Ci % 32 || Hi < 5 || Wi < 5) return HOIG_EUNSUPPORTED;
    WFlatArgs a;
    a.X = x; a.DY = dy; a.DW = dw; a.DB = dbias;
    a.Ci = Ci; a.Co = Co; a.Wc = Wi; a.HWc = Hi * Wi; a.Q = Bn * Hi * Wi; a.Ho = Hi - 4; a.Wo = Wi - 4;
    a.HPOS = PTW + (KS - 1) * (Wi + 1);
    if (a.HPOS * 8 > QSL * NT) return HOIG_EUNSUPPORTED;                        // canvas too wide for one halo image (Wi <= 66)
    a.nblk_ci = Ci / 32;
    a.nblk = (Co / 64) * a.nblk_ci;
    a.n_mtiles = (int)hoig_cdiv(a.Q, PTW);
    int splits = (int)hoig_cdiv(256, a.nblk);           // every pixel split costs |dW| fp32 atomics: one round of workgroups
    if (splits > a.n_mtiles) splits = a.n_mtiles;
    if (splits < 1) splits = 1;
    a.mt_per_split = (int)hoig_cdiv(a.n_mtiles, splits);
    splits = (int)hoig_cdiv(a.n_mtiles, a.mt_per_split);
    const int shm = ns_a(ns) * PTW * PSTR + ns_b(ns) * ((a.HPOS * QSTR + 255) / 256 * 256);
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_flat_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_flat_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_flat_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    dim3 grid(a.nblk, splits);
    HOIG_NS_SWITCH(ns, wgrad_flat_kernel<NSX><<<grid, NT, shm, st>>>(a));
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace hoig_detail
