// Stride-2 3x3 layers (Conv2d s2 p1 and ConvTranspose2d s2 p1 op1, forward and data gradient) on the 16x16 MFMA, statically walked.
#include "conv_m16_common.h"
#include "tuning.h"
#include <type_traits>

namespace hoig_detail {
namespace {

// The same kernel with its loads actually IN FLIGHT behind the MFMAs (round 5).  In conv_halo_s2_m16_kernel above the step table is
// walked at run time: "does step + 2 load a halo image?", "is there a next step?" are branches, and every halo load sits under its
// bounds test.  hipcc's wait-count pass gives up on that shape: the ISA has `s_waitcnt vmcnt(0)` right behind the weight loads of a step
// and again in front of its MFMAs, so a step is (weight latency) + (halo latency) + (MFMAs) in sequence -- 4.3 us per step for 0.7 us of
// MFMA work on the 256x256x64 layer (172 us against an HBM floor of 90).  Here the walk is STATIC: the five steps of a 32-channel block
// (gather) are written out, the last block is a second instantiation (no "is there a next" test inside a body), the two register sets
// of the halo prefetch are named, not indexed, and halo loads are unconditional (an out-of-image position reads its image's first
// pixel and is zeroed when the set is stored) -- the recipe of wgrad_dma.hip.  Arithmetic, LDS images, step order and epilogue are
// those of the kernel above: results are bit-identical.
//
// WM = 4 (round 5): the same walk over an 8 x 32 tile with eight waves -- a weight tile then serves 256 pixels instead of 128 and the halo
// overhead falls from 1.29 to 1.16, i.e. a third fewer bytes through the CU's load path per MFMA (that path, not the MFMA pipe, bounds
// these layers: 53 KB per 96-MFMA step and workgroup at WM = 2).  One workgroup per CU (72 KB of LDS, 2 waves per SIMD), which is
// exactly where the static walk pays.
template <int NSX, int BN, bool SCATTER, bool F16, int WM>
__global__ __launch_bounds__(128 * WM) void conv_halo_s2_m16p_kernel(const HaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;
    constexpr int TH = 2 * WM, TW = 32, WN = 2, NT = 64 * WM * WN;
    constexpr int RB = BN * 4 / NT;
    static_assert(RB >= 1, "more threads than 16-B weight chunks per tap");
    constexpr int HH = TH + 1, HW = TW + 1, HPIX = HH * HW;
    constexpr int PHALF = HPIX * 32, P23 = round128(PHALF) + 64, PLANE_P = round128(P23 + PHALF);
    constexpr int W23 = BN * 32 + 64, PLANE_W = round128(W23 + BN * 32);
    constexpr int MT = 4, NTW = BN / (16 * WN);
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];    // NS * PLANE_P + 2 * BBUF (s2p_lds)
    unsigned char *Ph = smem, *Pl = smem + PLANE_P;
    constexpr int BBUF = 2 * NB * PLANE_W;                 // two tap tiles of (Wh, Wl); DOUBLE-buffered
    unsigned char *Wbase = smem + NS * PLANE_P;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    int P = 0, Q = 0;
    if (SCATTER) {
        P = (tile >> 1) & 1;
        Q = tile & 1;
        tile >>= 2;
    }
    int mt_ = tile / p.nblk_n;
    const int n0 = (tile % p.nblk_n) * BN;
    const int tx_ = mt_ % p.tiles_x;
    mt_ /= p.tiles_x;
    const int ty_ = mt_ % p.tiles_y, b = mt_ / p.tiles_y;
    const int y0 = ty_ * TH, x0 = tx_ * TW;                // coarse-grid tile origin

    int wread[NTW], pread[MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wread[j] = (lg >> 1) * W23 + (wn * (NTW * 16) + j * 16 + l15) * 32 + (lg & 1) * 16;
#pragma unroll
    for (int m = 0; m < MT; ++m)
        pread[m] = (lg >> 1) * P23 + ((wm * 2 + (m >> 1)) * HW + (m & 1) * 16 + l15) * 32 + (lg & 1) * 16;

    f32x4 acc[NTW][MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ncb = p.Cg >> 5;
    // halo offset (rows, columns in {0,1}) of tap (r,s):  gather: (r != 0, s != 0)   scatter: (r == 0, s == 0)
    auto tap_off = [&](int tap) __attribute__((always_inline)) -> int {
        const int r = tap / 3, s_ = tap - r * 3;
        const int dr = SCATTER ? (r == 0) : (r != 0), dc = SCATTER ? (s_ == 0) : (s_ != 0);
        return (dr * HW + dc) * 32;
    };

    // ---- weights: the two tap tiles of the NEXT step go global -> LDS by LDS-DMA into the other weight buffer (the recipe of
    // conv_halo3_m16_kernel's WDMA: inline asm, M0 = the piece's LDS address, no staging registers, no ds_write; a 32-row block of a
    // blocked plane is 2 KB contiguous; piece q = (tap t, plane, block, half image h) = 1 KB of LDS: lane l copies row l >> 1's chunk
    // 2h + (l & 1), which the plane keeps at position chunk ^ ((row >> 2) & 3)).  Issued at the TOP of a step, before that step's halo
    // loads: the `s_waitcnt vmcnt(<halo loads of this step>)` that ends the step then finds the pieces landed and leaves the halo
    // loads in flight.
    constexpr int NWAVE = WM * WN, NPIECE = 2 * NB * (BN / 32) * 2, NPW = (NPIECE + NWAVE - 1) / NWAVE;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_w0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)Wbase);
    const int drow = lane >> 1;
    const unsigned dlane0 = drow * 32 + (((0 + (lane & 1)) ^ ((drow >> 2) & 3)) << 3);       // element offset inside a block, h = 0
    const unsigned dlane1 = drow * 32 + (((2 + (lane & 1)) ^ ((drow >> 2) & 3)) << 3);       // h = 1
    int wb = 1;                                            // the weight buffer the CURRENT step reads (wave-uniform)
    auto load_b = [&](int tap0, int tap1, int cb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int q = wave_u + NWAVE * i;
            if (q < NPIECE) {
                const int h = q & 1, blk = (q >> 1) % (BN / 32), tp = (q >> 1) / (BN / 32);  // tp = t * NB + plane
                const int t = tp / NB, pl = tp - t * NB;
                const size_t koff = (size_t)((t ? tap1 : tap0) * p.Cg + cb * 32) * 32;
                const unsigned short *src = (pl ? p.Wl : p.Wh) + ((size_t)((n0 >> 5) + blk) * (p.K >> 5)) * 1024 + koff +
                                            (h ? dlane1 : dlane0);
                const unsigned to = __builtin_amdgcn_readfirstlane(lds_w0 + (wb ^ 1) * BBUF + tp * PLANE_W + h * W23 + blk * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(to) : "memory");
            }
        }
    };

    // ---- halo images: loads are unconditional (an out-of-image position reads its image's first pixel and is zeroed at the store)
    constexpr int HSLICES = (HPIX * 8 + NT - 1) / NT;
    const float *const Ac = p.A + (size_t)b * p.H * p.W * p.Cg + (tid & 7) * 4;       // p.H x p.W: the gathered tensor
    struct HaloSet {
        float4 r[HSLICES];
        unsigned in;
    };
    HaloSet hs0, hs1;
    auto halo_load = [&](HaloSet &h, int cb, int pp, int qq) __attribute__((always_inline)) {
        // (opaque: with the phase a compile-time constant of each call site hipcc keeps every site's slice offsets in registers across
        // the whole block -- forty of them -- and spills; computed at the load they are ten VALU instructions per slice)
        asm volatile("" : "+s"(pp), "+s"(qq));
        h.in = 0;
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            // (positions recomputed per image: ten registers matter more here than ten VALU instructions; a slice past the halo fails
            // the bounds test)
            const int i = tid + NT * sl, pix = i >> 3;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int gy = SCATTER ? y0 + hy : 2 * (y0 - 1 + hy) + pp;
            const int gx = SCATTER ? x0 + hx : 2 * (x0 - 1 + hx) + qq;
            const bool in = i < HPIX * 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            h.in |= in ? (1u << sl) : 0u;
            h.r[sl] = *reinterpret_cast<const float4 *>(Ac + (in ? (size_t)(gy * p.W + gx) * p.Cg : (size_t)0) + cb * 32);
        }
    };
    auto halo_store = [&](HaloSet &h) __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            // (pins the split HERE: hipcc otherwise moves it up behind the load and waits for the data there)
            asm volatile("" : "+v"(h.r[sl].x), "+v"(h.r[sl].y), "+v"(h.r[sl].z), "+v"(h.r[sl].w));
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                uint2 hi, lo;
                split4t<F16>((h.in >> sl) & 1u ? h.r[sl] : make_float4(0.f, 0.f, 0.f, 0.f), hi, lo);
                const int off = (c4 >> 2) * P23 + pix * 32 + (c4 & 3) * 8;
                *reinterpret_cast<uint2 *>(Ph + off) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
            }
        }
    };
    auto compute = [&](int tap0, int tap1, int ntap) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t >= ntap) break;
            const int tapoff = tap_off(t ? tap1 : tap0);
            const unsigned char *Wh = Wbase + wb * BBUF + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
            bf16x8 ph[MT], pl[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ph[m] = *reinterpret_cast<const bf16x8 *>(Ph + pread[m] + tapoff);
                if (NS == 2) pl[m] = *reinterpret_cast<const bf16x8 *>(Pl + pread[m] + tapoff);
            }
            bf16x8 wh[2], wl[2];
            wh[0] = *reinterpret_cast<const bf16x8 *>(Wh + wread[0]);
            if (NB == 2) wl[0] = *reinterpret_cast<const bf16x8 *>(Wl + wread[0]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                if (j + 1 < NTW) {
                    wh[(j + 1) & 1] = *reinterpret_cast<const bf16x8 *>(Wh + wread[j + 1]);
                    if (NB == 2) wl[(j + 1) & 1] = *reinterpret_cast<const bf16x8 *>(Wl + wread[j + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (NS == 2) acc[j][m] = mfma_m16<F16>(wh[j & 1], pl[m], acc[j][m]);
                    if (NB == 2) acc[j][m] = mfma_m16<F16>(wl[j & 1], ph[m], acc[j][m]);
                    acc[j][m] = mfma_m16<F16>(wh[j & 1], ph[m], acc[j][m]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // the end of a step.  PEND = the halo loads this step issued behind its DMA pieces: they stay in flight, the pieces have landed;
    // everyone is done reading the step's weight buffer (the next DMA overwrites it) and the halo image; the next image goes in
    typedef std::integral_constant<int, HSLICES> Halo;
    typedef std::integral_constant<int, 0> None;
    auto turn = [&](HaloSet *h, auto pend_c) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(pend_c)::value) : "memory");
        __syncthreads();
        if (h) {
            halo_store(*h);
            __syncthreads();
        }
        wb ^= 1;
    };
    // the first step's operands: DMA into buffer 0, image from hs_first
    auto prologue = [&](HaloSet &h) __attribute__((always_inline)) {
        halo_store(h);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wb = 0;
    };

    if constexpr (!SCATTER) {
        // a 32-channel block = five steps: phase image (1,1) taps {0,2} then {6,8}; (1,0) {1,7}; (0,1) {3,5}; (0,0) {4}.  A phase
        // image is fetched two steps before the step that reads it and stored at the end of the step in between:
        //   step 0 fetches (1,0) -> hs0      step 1 fetches (0,1) -> hs1, stores hs0      step 2 fetches (0,0) -> hs0, stores hs1
        //   step 3 fetches the next block's (1,1) -> hs1, stores hs0                      step 4 stores hs1
        load_b(0, 2, 0);
        halo_load(hs1, 0, 1, 1);
        prologue(hs1);
        __syncthreads();
        auto block = [&](const int cb, auto last_c) __attribute__((always_inline)) {
            constexpr bool last = decltype(last_c)::value;
            load_b(6, 8, cb);
            halo_load(hs0, cb, 1, 0);
            compute(0, 2, 2);
            turn(nullptr, Halo{});
            load_b(1, 7, cb);
            halo_load(hs1, cb, 0, 1);
            compute(6, 8, 2);
            turn(&hs0, Halo{});
            load_b(3, 5, cb);
            halo_load(hs0, cb, 0, 0);
            compute(1, 7, 2);
            turn(&hs1, Halo{});
            load_b(4, 4, cb);
            if constexpr (!last) halo_load(hs1, cb + 1, 1, 1);
            compute(3, 5, 2);
            turn(&hs0, std::conditional_t<last, None, Halo>{});
            if constexpr (!last) load_b(0, 2, cb + 1);
            compute(4, 4, 1);
            if constexpr (!last) turn(&hs1, None{});
        };
#pragma unroll 1
        for (int cb = 0; cb + 1 < ncb; ++cb) block(cb, std::false_type{});
        block(ncb - 1, std::true_type{});
    } else {
        // scatter: ONE halo image per 32-channel block; output phase (P, Q) reads taps r in {P ? 0 : 1, 2 if P}, s likewise
        const int r0 = P ? 0 : 1, s0 = Q ? 0 : 1;
        if (P && Q) {
            // four taps, two steps per block: {(r0,s0),(r0,2)} then {(2,s0),(2,2)}; the next block's image is fetched in the first step
            // and stored at the end of the second
            const int ta0 = r0 * 3 + s0, ta1 = r0 * 3 + 2, tb0 = 6 + s0, tb1 = 8;
            load_b(ta0, ta1, 0);
            halo_load(hs0, 0, 0, 0);
            prologue(hs0);
            __syncthreads();
            auto block = [&](const int cb, auto last_c) __attribute__((always_inline)) {
                constexpr bool last = decltype(last_c)::value;
                load_b(tb0, tb1, cb);
                if constexpr (!last) halo_load(hs0, cb + 1, 0, 0);
                compute(ta0, ta1, 2);
                turn(nullptr, std::conditional_t<last, None, Halo>{});
                if constexpr (!last) load_b(ta0, ta1, cb + 1);
                compute(tb0, tb1, 2);
                if constexpr (!last) turn(&hs0, None{});
            };
#pragma unroll 1
            for (int cb = 0; cb + 1 < ncb; ++cb) block(cb, std::false_type{});
            block(ncb - 1, std::true_type{});
        } else {
            // one step per block (two taps, or the centre tap alone): the image of block cb + 2 is fetched in step cb (even blocks
            // in hs0, odd in hs1) and stored at the end of step cb + 1
            const int ntap = (P || Q) ? 2 : 1;
            const int t0 = r0 * 3 + s0, t1 = P ? 6 + s0 : (Q ? r0 * 3 + 2 : t0);
            load_b(t0, t1, 0);
            halo_load(hs0, 0, 0, 0);
            prologue(hs0);
            halo_load(hs1, ncb > 1 ? 1 : 0, 0, 0);         // (one block only: a harmless re-read)
            __syncthreads();
            // has1: block cb + 1 exists (its weights are fetched, its image stored); has2: block cb + 2 exists (its image is fetched)
            auto step = [&](const int cb, HaloSet &mine, HaloSet &other, auto has1_c, auto has2_c) __attribute__((always_inline)) {
                constexpr bool has1 = decltype(has1_c)::value, has2 = decltype(has2_c)::value;
                if constexpr (has1) load_b(t0, t1, cb + 1);
                if constexpr (has2) halo_load(mine, cb + 2, 0, 0);
                compute(t0, t1, ntap);
                if constexpr (has1) turn(&other, std::conditional_t<has2, Halo, None>{});
            };
            int cb = 0;
#pragma unroll 1
            for (; cb + 3 < ncb; cb += 2) {
                step(cb, hs0, hs1, std::true_type{}, std::true_type{});
                step(cb + 1, hs1, hs0, std::true_type{}, std::true_type{});
            }
            const int rest = ncb - cb;                     // 1, 2 or 3 steps left, the first of them an even block
            if (rest == 3) {
                step(cb, hs0, hs1, std::true_type{}, std::true_type{});
                step(cb + 1, hs1, hs0, std::true_type{}, std::false_type{});
                step(cb + 2, hs0, hs1, std::false_type{}, std::false_type{});
            } else if (rest == 2) {
                step(cb, hs0, hs1, std::true_type{}, std::false_type{});
                step(cb + 1, hs1, hs0, std::false_type{}, std::false_type{});
            } else {
                step(cb, hs0, hs1, std::false_type{}, std::false_type{});
            }
        }
    }

    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    float4 bias_r[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = n0 + wn * (NTW * 16) + j * 16 + lg * 4;
        bias_r[j] = (p.bias && n < p.N) ? *reinterpret_cast<const float4 *>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float st1[NTW][4], st2[NTW][4];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) st1[j][q] = st2[j][q] = 0.f;
    // output grid: gather mode = the coarse grid; scatter mode = twice the coarse grid, phase (P,Q)
    const int Ho = SCATTER ? 2 * p.H : p.tiles_y * TH, Wo = SCATTER ? 2 * p.W : p.tiles_x * TW;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int cy = y0 + wm * 2 + (m >> 1), cx = x0 + (m & 1) * 16 + l15;
        const int oy = SCATTER ? 2 * cy + P : cy, ox = SCATTER ? 2 * cx + Q : cx;
        const size_t pix = ((size_t)b * Ho + oy) * Wo + ox;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = n0 + wn * (NTW * 16) + j * 16 + lg * 4;
            if (n < p.N) {
                float v[4];
                const float bq[4] = {bias_r[j].x, bias_r[j].y, bias_r[j].z, bias_r[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fast_act(acc[j][m][q] * p.oscale + bq[q], nslope, special, p.act, p.slope);
                if (p.addend) {
                    const float4 ad = *reinterpret_cast<const float4 *>(p.addend + pix * p.N + n);
                    v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    st1[j][q] += v[q];
                    st2[j][q] += v[q] * v[q];
                }
                *reinterpret_cast<float4 *>(p.C + pix * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (p.stats) m16_stats_epilogue<NTW, WM, BN, NT>(st1, st2, smem, p.stats + (size_t)b * 2 * p.N, p.N, n0, wm, wn, lane, tid);
}

template <int BN, int WM>
constexpr int s2p_lds(int nsx) {
    constexpr int HPIX = (2 * WM + 1) * 33, PHALF = HPIX * 32, P23 = round128(PHALF) + 64, PLANE_P = round128(P23 + PHALF);
    constexpr int W23 = BN * 32 + 64, PLANE_W = round128(W23 + BN * 32);
    return (nsx == 1 ? 1 : 2) * PLANE_P + 2 * (2 * (nsx == 2 ? 2 : 1) * PLANE_W);
}

template <int NSX, int BN, bool SCATTER, bool F16, int WM>
int launch_s2p_one(const HaloArgs &a, hipStream_t st) {
    constexpr int shm = s2p_lds<BN, WM>(NSX);
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo_s2_m16p_kernel<NSX, BN, SCATTER, F16, WM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, shm) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    conv_halo_s2_m16p_kernel<NSX, BN, SCATTER, F16, WM><<<a.nblk, 128 * WM, shm, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

template <bool SCATTER, int WM>
int launch_s2p(const HaloArgs &a, int ns, hipStream_t st) {
    if ((a.N % 128) != 0) {
        if constexpr (WM == 2) {
            if (a.f16) HOIG_NS_SWITCH(ns, return launch_s2p_one<NSX, 64, SCATTER, true, 2>(a, st));
            HOIG_NS_SWITCH(ns, return launch_s2p_one<NSX, 64, SCATTER, false, 2>(a, st));
        }
        return HOIG_EUNSUPPORTED;
    }
    if (a.f16) HOIG_NS_SWITCH(ns, return launch_s2p_one<NSX, 128, SCATTER, true, WM>(a, st));
    HOIG_NS_SWITCH(ns, return launch_s2p_one<NSX, 128, SCATTER, false, WM>(a, st));
    return HOIG_EINVAL;
}

}  // namespace

// `a` arrives with the 4-row geometry launch_halo_s2 (conv_igemm_bf16.hip) computes; rows8: re-tile to 8 x 32 (the caller has checked
// that the coarse height is a multiple of 8 and N of 128)
int launch_halo_s2_m16p(const HaloArgs &a_, int ns, bool scatter, bool rows8, hipStream_t st) {
    if (!rows8) return scatter ? launch_s2p<true, 2>(a_, ns, st) : launch_s2p<false, 2>(a_, ns, st);
    HaloArgs a = a_;
    if ((a.tiles_y & 1) || (a.N % 128)) return HOIG_EUNSUPPORTED;
    a.tiles_y /= 2;
    a.nblk /= 2;
    return scatter ? launch_s2p<true, 4>(a, ns, st) : launch_s2p<false, 4>(a, ns, st);
}

}  // namespace hoig_detail
