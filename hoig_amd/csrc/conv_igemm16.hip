// Generic implicit-GEMM convolution (strided, transposed, 1x1, valid 5x5, 4x4: every shape the halo kernels do not take) on
// v_mfma_f32_16x16x32_{f16,bf16}: the structure of igemm_bf16_kernel (conv_igemm_bf16.hip) -- BM x BN output tiles, BK = 32, one
// LDS stage so that several workgroups share a CU, tap-outer k walk with per-tap pointer set-up, split-K over blockIdx.y -- on
// the 16x16 MFMA shape (MI355X_MICROARCH.md 'DVFS give-back' item 7: it holds a higher clock on random data than 32x32x16).
//
// LDS images as in conv_halo16.hip: each operand is two half images [row][32 B] (k-chunks {0,1} | {2,3}), the second 64 B past
// a multiple of 128 B; ds_read_b128 fragments and the staging stores are conflict-free.
//
// Two result orientations:
//   plain launches : weights are the A operand, so a lane holds four consecutive CHANNELS of one pixel -> 16-B stores;
//   split-K        : pixels are the A operand, so a register holds 16 channels x 4 pixels; one v_permlane16_swap per register
//                    pair of two adjacent channel tiles makes every atomic instruction two 128-B runs in two rows, the
//                    full-rate shape of the memory-side atomic units.
#include "conv_bf16_common.h"

namespace hoig_detail {
namespace {

template <bool F16>
__device__ __forceinline__ f32x4 mfma_m16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

typedef unsigned u2_t __attribute__((ext_vector_type(2)));

template <int BM, int BN, int WM, int WN, int NSX, bool F16, bool SPLITK>
__global__ __launch_bounds__(WM * WN * 64) void igemm_m16_kernel(const Args p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: pixels (activations / dy), weights
    constexpr int NT = WM * WN * 64;
    constexpr int MT = BM / (16 * WM), NTW = BN / (16 * WN);          // 16-pixel / 16-channel tiles per wave
    constexpr int RA = BM * 8 / NT;         // float4 gathers per thread
    constexpr int RB = BN * 4 / NT;         // 16-B weight chunks per thread per plane
    constexpr int AROWS = NT / 8, BROWS = NT / 4;
    constexpr int A23 = BM * 32 + 64, PLANE_A = (A23 + BM * 32 + 127) / 128 * 128;
    constexpr int B23 = BN * 32 + 64, PLANE_B = (B23 + BN * 32 + 127) / 128 * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * PLANE_A + NB * PLANE_B];
    unsigned char *Ah = smem, *Al = Ah + PLANE_A;
    unsigned char *Bh = smem + NS * PLANE_A, *Bl = Bh + PLANE_B;

    const Geom &g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int m0 = (tile / p.nblk_n) * BM, n0 = (tile % p.nblk_n) * BN;

    const int c4 = tid & 7, lrow = tid >> 3;
    int pb[RA], bh[RA], bw[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lrow + AROWS * i;
        if (m < p.M) {
            int b, hp, wp;
            decode_m(g, m, b, hp, wp);
            pb[i] = b * g.Hg;
            bh[i] = row_base(g, hp);
            bw[i] = row_base(g, wp);
        } else {
            pb[i] = -1;
            bh[i] = bw[i] = 0;
        }
    }
    int t_hp = 0, t_wp = 0;
    if (g.tile_skip) {
        int b;
        decode_m(g, m0, b, t_hp, t_wp);
    }
    const int brow = tid >> 2, bpos = tid & 3;

    f32x4 acc[NTW][MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[RA];
    uint4 rbh[RB], rbl[RB];
    const int cpb = g.Cg / 32, RS = g.R * g.S;
    const float *aptr[RA];
    const unsigned short *wrow_h[RB], *wrow_l[RB];
    int aoff[RA], boff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = brow + BROWS * i;
        const int n = n0 + row;
        const size_t o = ((size_t)(n >> 5) * (p.K >> 5)) * 1024 + (n & 31) * 32 + bpos * 8;      // physical position bpos of row n
        wrow_h[i] = n < p.N ? p.Wh + o : nullptr;
        wrow_l[i] = (NB == 2 && n < p.N) ? p.Wl + o : nullptr;
        const int c = bpos ^ ((row >> 2) & 3);                    // the logical chunk that position holds (plane_index)
        boff[i] = (c >> 1) * B23 + row * 32 + (c & 1) * 16;
    }
#pragma unroll
    for (int i = 0; i < RA; ++i) aoff[i] = (c4 >> 2) * A23 + (lrow + AROWS * i) * 32 + (c4 & 3) * 8;
    // split-K: this workgroup multiplies k-blocks [begin, begin + steps_per_split) of the (tap, channel-block) walk
    const int begin = p.ksplit > 1 ? (int)blockIdx.y * p.steps_per_split : 0;
    int steps_left = p.ksplit > 1 ? min(p.steps_per_split, RS * cpb - begin) : 0x7fffffff;
    int rs = begin / cpb - 1, cb = cpb - 1, wk = 0;
    int cb_next_tap = begin - (begin / cpb) * cpb;      // channel block to start the first tap at
    auto advance = [&]() -> bool {      // move (rs, cb) to the next live k-block; false when K is exhausted
        if (steps_left-- <= 0) return false;
        if (++cb < cpb) return true;
        cb = cb_next_tap;
        cb_next_tap = 0;
        do {
            ++rs;
        } while (rs < RS && g.tile_skip && !tap_alive(g, t_hp, t_wp, rs));
        if (rs >= RS) return false;
        const int r = rs / g.S, s_ = rs - r * g.S;
        wk = rs * g.Cg;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            aptr[i] = nullptr;
            if (pb[i] >= 0) {
                const int hg = gcoord(g, bh[i], r, g.Hg), wg = gcoord(g, bw[i], s_, g.Wg);
                if (hg >= 0 && wg >= 0) aptr[i] = p.A + ((size_t)(pb[i] + hg) * g.Wg + wg) * g.Cg + c4 * 4;
            }
        }
        return true;
    };
    auto load_tiles = [&]() {
        const int c = cb * 32;
#pragma unroll
        for (int i = 0; i < RA; ++i)
            ra[i] = aptr[i] ? *reinterpret_cast<const float4 *>(aptr[i] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            rbh[i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + (size_t)(wk + c) * 32) : make_uint4(0, 0, 0, 0);
            if (NB == 2)
                rbl[i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + (size_t)(wk + c) * 32) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            uint2 hi, lo;
            split4t<F16>(ra[i], hi, lo);
            *reinterpret_cast<uint2 *>(Ah + aoff[i]) = hi;
            if (NS == 2) *reinterpret_cast<uint2 *>(Al + aoff[i]) = lo;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            *reinterpret_cast<uint4 *>(Bh + boff[i]) = rbh[i];
            if (NB == 2) *reinterpret_cast<uint4 *>(Bl + boff[i]) = rbl[i];
        }
    };
    int pread[MT], wread[NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m) pread[m] = (lg >> 1) * A23 + (wm * (MT * 16) + m * 16 + l15) * 32 + (lg & 1) * 16;
#pragma unroll
    for (int j = 0; j < NTW; ++j) wread[j] = (lg >> 1) * B23 + (wn * (NTW * 16) + j * 16 + l15) * 32 + (lg & 1) * 16;

    bool more = advance();
    if (more) {
        load_tiles();
        store_tiles();
    }
    __syncthreads();
    while (more) {
        const bool nxt = advance();
        if (nxt) load_tiles();
        bf16x8 ph[MT], pl[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ph[m] = *reinterpret_cast<const bf16x8 *>(Ah + pread[m]);
            if (NS == 2) pl[m] = *reinterpret_cast<const bf16x8 *>(Al + pread[m]);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(Bh + wread[j]);
            bf16x8 wl;
            if (NB == 2) wl = *reinterpret_cast<const bf16x8 *>(Bl + wread[j]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (SPLITK) {
                    if (NS == 2) acc[j][m] = mfma_m16<F16>(pl[m], wh, acc[j][m]);
                    if (NB == 2) acc[j][m] = mfma_m16<F16>(ph[m], wl, acc[j][m]);
                    acc[j][m] = mfma_m16<F16>(ph[m], wh, acc[j][m]);
                } else {
                    if (NS == 2) acc[j][m] = mfma_m16<F16>(wh, pl[m], acc[j][m]);
                    if (NB == 2) acc[j][m] = mfma_m16<F16>(wl, ph[m], acc[j][m]);
                    acc[j][m] = mfma_m16<F16>(wh, ph[m], acc[j][m]);
                }
            }
        }
        __syncthreads();
        if (nxt) store_tiles();
        __syncthreads();
        more = nxt;
    }

    if (!SPLITK) {
        // lane -> pixel (lane & 15) of pixel tile m, channels 4 * (lane >> 4) .. + 3 of channel tile j
        const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
        const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
        float4 bias_r[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = n0 + wn * (NTW * 16) + j * 16 + lg * 4;
            bias_r[j] = (p.bias && n < p.N) ? *reinterpret_cast<const float4 *>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mm = m0 + wm * (MT * 16) + m * 16 + l15;
            if (mm >= p.M) continue;
            size_t pix = mm;
            if (g.phase_major) {
                int b, hp, wp;
                decode_m(g, mm, b, hp, wp);
                pix = ((size_t)b * g.Hp + hp) * g.Wp + wp;
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = n0 + wn * (NTW * 16) + j * 16 + lg * 4;
                if (n < p.N) {
                    float4 o;
                    o.x = fast_act(acc[j][m][0] * p.oscale + bias_r[j].x, nslope, special, p.act, p.slope);
                    o.y = fast_act(acc[j][m][1] * p.oscale + bias_r[j].y, nslope, special, p.act, p.slope);
                    o.z = fast_act(acc[j][m][2] * p.oscale + bias_r[j].z, nslope, special, p.act, p.slope);
                    o.w = fast_act(acc[j][m][3] * p.oscale + bias_r[j].w, nslope, special, p.act, p.slope);
                    *reinterpret_cast<float4 *>(p.C + pix * p.N + n) = o;
                }
            }
        }
    } else {
        // register r of acc[j][m]: pixel 4 * (lane >> 4) + r of pixel tile m, channel (lane & 15) of channel tile j.  After the swap
        // of the registers of tiles j, j + 1: lanes 0-31 / 32-63 of the first hold channels 0..31 of the pair at pixels r / 8 + r,
        // of the second at pixels 4 + r / 12 + r (wgrad_halo16.hip)
        static_assert(!SPLITK || NTW % 2 == 0, "split-K pairs adjacent channel tiles");
        const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NTW; j += 2) {
                    const int n = n0 + wn * (NTW * 16) + j * 16 + l31;
                    const u2_t sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[j][m][r]), __float_as_uint(acc[j + 1][m][r]),
                                                                     false, false);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int mm = m0 + wm * (MT * 16) + m * 16 + r + 8 * lh + 4 * h;
                        if (mm < p.M && n < p.N) {
                            float v = __uint_as_float(sw[h]) * p.oscale;
                            if (blockIdx.y == 0 && p.bias) v += p.bias[n];
                            atomicAdd(&p.C[(size_t)mm * p.N + n], v);
                        }
                    }
                }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(Args a, int ns, hipStream_t st) {
    constexpr int NT = WM * WN * 64;
    const int nbm = (int)hoig_cdiv(a.M, BM), nbn = (int)hoig_cdiv(a.N, BN);
    a.nblk_n = nbn;
    a.nblk = nbm * nbn;
    if (a.g.gatherT && a.g.stride == 2 && (a.g.Hp % 2 == 0) && (a.g.Wp % 2 == 0)) {
        a.g.phase_major = 1;
        const long per_phase = (long)a.g.Bn * (a.g.Hp / 2) * (a.g.Wp / 2);
        a.g.tile_skip = (per_phase % BM == 0) ? 1 : 0;
    }
    // few output tiles but a long K (the attention's 5x5 convolutions: 128 outputs, K = 25 * C): split K over blockIdx.y
    a.ksplit = 1;
    a.steps_per_split = 0;
    const int steps = a.K / 32;
    if (a.nblk < 192 && steps >= 32 && a.act == HOIG_ACT_NONE && !a.g.tile_skip && !a.g.phase_major) {
        int want = (int)hoig_cdiv(1024, a.nblk);
        if (want > steps / 8) want = steps / 8;
        if (want > 1) {
            a.steps_per_split = (int)hoig_cdiv(steps, want);
            a.ksplit = (int)hoig_cdiv(steps, a.steps_per_split);
            if (hipMemsetAsync(a.C, 0, (size_t)a.M * a.N * sizeof(float), st) != hipSuccess) return HOIG_ELAUNCH;
        }
    }
    dim3 grid(a.nblk, a.ksplit);
    if (a.ksplit > 1) {
        if (a.f16) HOIG_NS_SWITCH(ns, igemm_m16_kernel<BM, BN, WM, WN, NSX, true, true><<<grid, NT, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, igemm_m16_kernel<BM, BN, WM, WN, NSX, false, true><<<grid, NT, 0, st>>>(a));
    } else {
        if (a.f16) HOIG_NS_SWITCH(ns, igemm_m16_kernel<BM, BN, WM, WN, NSX, true, false><<<grid, NT, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, igemm_m16_kernel<BM, BN, WM, WN, NSX, false, false><<<grid, NT, 0, st>>>(a));
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace

int launch_igemm_m16(Args a, int ns, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: return launch<128, 128, 2, 2>(a, ns, st);
        case 1: return launch<128, 128, 2, 4>(a, ns, st);
        case 2: return launch<64, 128, 2, 2>(a, ns, st);
        case 3: return launch<128, 64, 2, 2>(a, ns, st);
        case 4: return launch<64, 64, 2, 2>(a, ns, st);
        default: return HOIG_EUNSUPPORTED;
    }
}

}  // namespace hoig_detail
