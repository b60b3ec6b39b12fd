// 3x3 stride-1 "same" convolution (forward and data gradient) on v_mfma_f32_16x16x32_{f16,bf16}: the same workgroup tile and
// step structure as conv_halo3_bf16_kernel MODE 2 (conv_igemm_bf16.hip: 8 rows x 32 pixels x BN channels, 8 waves, the input
// halo of a 32-channel block staged once and split to 16-bit hi / lo planes, one TAP ROW of weight tiles per step,
// double-buffered), on the 16x16 MFMA shape, which holds a higher clock than 32x32x16 on random data (MI355X_MICROARCH.md,
// 'DVFS give-back' item 7; cdna_hip_programming.md rule 28).
//
// Operand roles are swapped against the 32x32 kernel: the WEIGHTS are the A operand (rows of the result = output channels)
// and the PIXELS the B operand (columns), so a lane ends up with FOUR CONSECUTIVE CHANNELS of one pixel (D: col = lane & 15,
// row = 4 * (lane >> 4) + reg) and the epilogue stores 16 B per lane: 16 dwordx4 stores per wave instead of 64 dword stores.
//
// LDS images: both operands are stored as two "half images" of [row][32 B] -- the two 16-B k-chunks {0,1} (resp. {2,3}) of a
// row side by side, no padding.  A ds_read_b128 of this MFMA shape has lanes 0-15 on chunk 0 of rows r..r+15, lanes 16-31 on
// chunk 1, lanes 32-47 / 48-63 on chunks 2 / 3 (the other half image); its four bank cycles each see 8 even and 8 odd 16-B
// slots (2 * row + chunk): conflict-free at every tap shift.  The second half image starts 64 B past a multiple of 128 B, so
// that the staging stores (ds_write_b64 of the split halo: 2 pixels x 8 lanes; ds_write_b128 of the weight chunks: 2 rows x 4
// lanes) fill a whole 128-B bank window per lane group.
#include "conv_bf16_common.h"
#include "conv_m16_common.h"
#include "tuning.h"

namespace hoig_detail {
namespace {

template <int WM, int BN>
struct M16Layout {
    static constexpr int TH = 2 * WM, HW = 34, HPIX = (TH + 2) * HW;
    static constexpr int PHALF = HPIX * 32;                       // one half image of the halo: [pixel][32 B]
    static constexpr int P23 = round128(PHALF) + 64;              // offset of the second half image (k-chunks 2, 3)
    static constexpr int PLANE_P = round128(P23 + PHALF);
    static constexpr int W23 = BN * 32 + 64;
    static constexpr int PLANE_W = round128(W23 + BN * 32);
    static constexpr size_t bytes(int nsx) { return (size_t)ns_a(nsx) * PLANE_P + 2 * 3 * ns_b(nsx) * PLANE_W; }
};

// ASPLIT (data-gradient launches, round 5): the gathered tensor arrives PRE-SPLIT -- per pixel [hi: Cg bf16][lo: Cg bf16], written by
// the producer of that gradient (hoig_split_planes_bf16; the backward of a norm) -- and the halo goes global -> registers -> LDS in
// 16-B pieces with no VALU work: the split of a 32-channel block (ten VALU instructions and two ds_write_b64 per 16 B of input,
// between two barriers, beside nobody's MFMAs) is gone, as is the same split in every other workgroup that reads these pixels.
// WDMA (round 5; tuning key `wdma16`): the weight tiles of a step go global -> LDS by LDS-DMA (global_load_lds_dwordx4 from inline asm:
// through the builtin hipcc guards the step's first ds_read with `s_waitcnt vmcnt(0)` and the copy lands BEFORE the MFMAs instead of
// behind them -- which is what round 2's "9 % slower" measurement of the builtin form ran into).  A 32-row block of a blocked plane is
// 2 KB contiguous in HBM; a piece = (block, half image h) = 1 KB of LDS: lane l copies row l >> 1's chunk 2h + (l & 1), which the plane
// keeps at position chunk ^ ((row >> 2) & 3).  The step's 3 x NB x BN/32 x 2 pieces are spread over the waves and, per wave, over the
// step's MFMA groups; they are issued for step + 1 into the other weight buffer and waited for (vmcnt(0)) before the step's closing
// barrier.  No staging registers (24 VGPRs), no ds_write, no load two steps ahead.
// NORMIN (round 5; inference only): HaloArgs::in_scale / in_shift / in_relu_c0 -- the instance norm (+ affine) and ReLU that sit between
// the producing convolution and this one are applied HERE, to the raw tensor, when the halo image is converted (one FMA and one max per
// element, beside the split that is there anyway): the norm's apply pass -- a read and a write of the tensor -- does not run.
template <int NSX, int WM, int WN, int BN, bool F16, bool ASPLIT = false, bool WDMA = false, bool NORMIN = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_halo3_m16_kernel(const HaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;      // operand planes: pixels (activations / dy), weights
    using LY = M16Layout<WM, BN>;
    constexpr int KS = 3, TH = LY::TH, HW = LY::HW, HPIX = LY::HPIX;
    constexpr int NT = 64 * WM * WN;
    constexpr int P23 = LY::P23, PLANE_P = LY::PLANE_P, W23 = LY::W23, PLANE_W = LY::PLANE_W;
    constexpr int RB = BN * 4 >= NT ? BN * 4 / NT : 1;     // 16-B weight chunks per thread per plane per tap
    constexpr bool B_PART = BN * 4 < NT;                   // more threads than chunks
    constexpr int MT = 4;                                  // pixel tiles of 16 per wave: 2 rows x 2 halves
    constexpr int NTW = BN / (16 * WN);                    // channel tiles of 16 per wave
    constexpr int ABUF = NS * PLANE_P, BBUF = KS * NB * PLANE_W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // ABUF + 2 * BBUF
    unsigned char *Pbase = smem;
    unsigned char *Wbase = smem + ABUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int n_mt = p.nblk / p.nblk_n;
    int mt_ = p.nmajor ? tile % n_mt : tile / p.nblk_n;
    const int n0 = (p.nmajor ? tile / n_mt : tile % p.nblk_n) * BN;
    const int tx_ = mt_ % p.tiles_x;
    mt_ /= p.tiles_x;
    const int ty_ = mt_ % p.tiles_y, b = mt_ / p.tiles_y;
    const int y0 = ty_ * TH, x0 = tx_ * 32;
    // grouped launch (p.b_split > 0): images b >= b_split are a SECOND problem of the same shape -- own input, weights, bias, addend,
    // output -- whose tiles ride in this grid (src_model's and tsf_model's layer as one launch that fills the chip, instead of two
    // half-chip launches on two streams); everything below addresses "its" tensors through these
    const bool g2 = p.b_split > 0 && b >= p.b_split;
    const int bi = g2 ? b - p.b_split : b;
    const float *const gA = g2 ? p.A_g2 : p.A;
    const unsigned short *const gWh = g2 ? p.Wh_g2 : p.Wh, *const gWl = g2 ? p.Wl_g2 : p.Wl;
    const float *const gbias = g2 ? p.bias_g2 : p.bias, *const gadd = g2 ? p.addend_g2 : p.addend;
    float *const gC = g2 ? p.C_g2 : p.C;

    // weight staging: thread -> (row, position in the 64-B row of the blocked plane); the plane's position holds logical chunk
    // pos ^ ((row >> 2) & 3) (plane_index)
    const int brow = tid >> 2, bpos = tid & 3;
    const unsigned short *wrow_h[RB], *wrow_l[RB];
    int woff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = brow + (NT / 4) * i;
        const int n = n0 + row;
        const bool ok = n < p.N && (!B_PART || row < BN);
        const size_t o = ((size_t)(n >> 5) * (p.K >> 5)) * 1024 + (n & 31) * 32 + bpos * 8;
        wrow_h[i] = ok ? gWh + o : nullptr;
        wrow_l[i] = (NB == 2 && ok) ? gWl + o : nullptr;
        const int c = bpos ^ ((row >> 2) & 3);
        woff[i] = (c >> 1) * W23 + row * 32 + (c & 1) * 16;
    }
    // fragment read offsets
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

    uint4 rbh[KS][RB], rbl[KS][RB];
    const int ncb = p.Cg >> 5, T = ncb * KS;               // step = (channel block, tap row)
    auto load_b = [&](int step) {
        const int cb = step / KS, r = step - cb * KS;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const int tap = r * KS + t;
            const int wtap = p.flip ? (KS * KS - 1 - tap) : tap;
            const size_t koff = (size_t)(wtap * p.Cg + cb * 32) * 32;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                rbh[t][i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + koff) : make_uint4(0, 0, 0, 0);
                if (NB == 2)
                    rbl[t][i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + koff) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            unsigned char *Wh = Wbase + buf * BBUF + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (B_PART && brow >= BN) continue;
                *reinterpret_cast<uint4 *>(Wh + woff[i]) = rbh[t][i];
                if (NB == 2) *reinterpret_cast<uint4 *>(Wl + woff[i]) = rbl[t][i];
            }
        }
    };
    // ---- WDMA: piece q of a step = (tap t, plane pl, 32-row block blk, half image h); wave w issues pieces w, w + 8, ..
    constexpr int NPIECE_STEP = KS * NB * (BN / 32) * 2, NPW = (NPIECE_STEP + WM * WN - 1) / (WM * WN);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_w0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)Wbase);
    const int drow = lane >> 1;
    const unsigned dlane0 = drow * 32 + (((0 + (lane & 1)) ^ ((drow >> 2) & 3)) << 3);       // element offset inside a block, h = 0
    const unsigned dlane1 = drow * 32 + (((2 + (lane & 1)) ^ ((drow >> 2) & 3)) << 3);       // h = 1
    auto dma_piece = [&](int step, int buf, int i) {
        const int q = wave_u + (WM * WN) * i;
        if (q >= NPIECE_STEP) return;
        const int h = q & 1, blk = (q >> 1) % (BN / 32), tp = (q >> 1) / (BN / 32);          // tp = t * NB + plane
        const int t = tp / NB, pl = tp - t * NB;
        const int cb = step / KS, r = step - cb * KS;
        const int tap = r * KS + t;
        const int wtap = p.flip ? (KS * KS - 1 - tap) : tap;
        const size_t koff = (size_t)(wtap * p.Cg + cb * 32) * 32;
        const unsigned short *plane = pl ? gWl : gWh;
        const unsigned short *src = plane + ((size_t)((n0 >> 5) + blk) * (p.K >> 5)) * 1024 + koff + (h ? dlane1 : dlane0);
        const unsigned to = __builtin_amdgcn_readfirstlane(lds_w0 + buf * BBUF + tp * PLANE_W + h * W23 + blk * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(to) : "memory");
    };
    constexpr int HSLICES = (HPIX * 8 + NT - 1) / NT;
    float4 hreg[HSLICES];
    float4 nsc = make_float4(1.f, 1.f, 1.f, 1.f), nsh = make_float4(0.f, 0.f, 0.f, 0.f);      // NORMIN: scale / shift of the image in hreg[]
    bool nrelu = false;
    unsigned hin = 0;                                      // NORMIN: bit sl = slice sl lies inside the image
    auto halo_load = [&](int cb) {
        if constexpr (ASPLIT) {
            // slice i -> (pixel i >> 3, plane (i >> 2) & 1, 16-B chunk i & 3 of the block's 32 channels)
            const unsigned short *Aimg = reinterpret_cast<const unsigned short *>(gA) + (size_t)bi * p.H * p.W * 2 * p.Cg + cb * 32;
#pragma unroll
            for (int sl = 0; sl < HSLICES; ++sl) {
                const int i = tid + NT * sl;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < HPIX * 8 && (NS == 2 || !(i & 4))) {
                    const int pix = i >> 3, plane = (i >> 2) & 1, c8 = i & 3;
                    const int hy = pix / HW, hx = pix - hy * HW;
                    const int gy = y0 - p.pad + hy, gx = x0 - p.pad + hx;
                    if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                        v = *reinterpret_cast<const float4 *>(Aimg + ((size_t)gy * p.W + gx) * 2 * p.Cg + plane * p.Cg + c8 * 8);
                }
                hreg[sl] = v;
            }
            return;
        }
        const bool second = p.A2 != nullptr && cb * 32 >= p.cg1;
        const int ld = p.A2 ? (second ? p.Cg - p.cg1 : p.cg1) : p.Cg;
        const float *Aimg = (second ? p.A2 : gA) + (size_t)bi * p.H * p.W * ld + (second ? cb * 32 - p.cg1 : cb * 32);
        if constexpr (NORMIN) {            // this thread's four channels of the block (slice i -> channel quad i & 7 = tid & 7: NT % 8 == 0)
            const size_t co = (size_t)bi * p.Cg + cb * 32 + (tid & 7) * 4;
            nsc = *reinterpret_cast<const float4 *>(p.in_scale + co);
            nsh = *reinterpret_cast<const float4 *>(p.in_shift + co);
            nrelu = cb * 32 + (tid & 7) * 4 >= p.in_relu_c0;
            hin = 0;
        }
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int gy = y0 - p.pad + hy, gx = x0 - p.pad + hx;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    v = *reinterpret_cast<const float4 *>(Aimg + ((size_t)gy * p.W + gx) * ld + c4 * 4);
                    if constexpr (NORMIN) hin |= 1u << sl;
                }
            }
            hreg[sl] = v;
        }
    };
    auto halo_store = [&]() {
        unsigned char *Ph = Pbase, *Pl = Ph + PLANE_P;
        if constexpr (ASPLIT) {
#pragma unroll
            for (int sl = 0; sl < HSLICES; ++sl) {
                const int i = tid + NT * sl;
                if (i < HPIX * 8 && (NS == 2 || !(i & 4))) {
                    const int pix = i >> 3, plane = (i >> 2) & 1, c8 = i & 3;
                    *reinterpret_cast<float4 *>(Pbase + plane * PLANE_P + (c8 >> 1) * P23 + pix * 32 + (c8 & 1) * 16) = hreg[sl];
                }
            }
            return;
        }
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                uint2 hi, lo;
                float4 v = hreg[sl];
                if constexpr (NORMIN) {
                    if ((hin >> sl) & 1u) {             // (the zero padding stays zero: the norm applies to the tensor, not to its frame)
                        v.x = fmaf(v.x, nsc.x, nsh.x); v.y = fmaf(v.y, nsc.y, nsh.y);
                        v.z = fmaf(v.z, nsc.z, nsh.z); v.w = fmaf(v.w, nsc.w, nsh.w);
                        if (nrelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                }
                split4t<F16>(v, hi, lo);
                const int off = (c4 >> 2) * P23 + pix * 32 + (c4 & 3) * 8;
                *reinterpret_cast<uint2 *>(Ph + off) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
            }
        }
    };

    // One step = the three taps of row r.  A group = (tap, channel tile): its two weight fragments (hi, lo) against the
    // tap's eight pixel fragments (4 tiles x hi / lo, read once per tap): 12 MFMAs.  The fragments of the NEXT group -- and a
    // quarter of the next tap's pixel fragments -- are read before the MFMAs of the current group issue.
    struct PF {
        bf16x8 h[MT], l[MT];
    };
    struct WF {
        bf16x8 h, l;
    };
    auto compute = [&](int r, int bbuf, int dma_step) {      // dma_step >= 0 (WDMA): the step whose weight pieces ride between the groups
        const unsigned char *Ph = Pbase, *Pl = Ph + PLANE_P;
        const unsigned char *Wst = Wbase + bbuf * BBUF;
        auto read_p = [&](PF &f, int t, int m) {
            const int off = pread[m] + (r * HW + t) * 32;
            f.h[m] = *reinterpret_cast<const bf16x8 *>(Ph + off);
            if (NS == 2) f.l[m] = *reinterpret_cast<const bf16x8 *>(Pl + off);
        };
        auto read_w = [&](WF &f, int t, int j) {
            const unsigned char *Wh = Wst + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
            f.h = *reinterpret_cast<const bf16x8 *>(Wh + wread[j]);
            if (NB == 2) f.l = *reinterpret_cast<const bf16x8 *>(Wl + wread[j]);
        };
        PF pf[2];
        WF wf[2];
#pragma unroll
        for (int m = 0; m < MT; ++m) read_p(pf[0], 0, m);
        read_w(wf[0], 0, 0);
        constexpr int NG = KS * NTW;
        constexpr int PPG = (MT + NTW - 1) / NTW;          // pixel tiles of the next tap read per group
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int t = g / NTW, j = g % NTW;
            if (g + 1 < NG) read_w(wf[(g + 1) & 1], (g + 1) / NTW, (g + 1) % NTW);
            if (t + 1 < KS) {
#pragma unroll
                for (int q = 0; q < PPG; ++q)
                    if (j * PPG + q < MT) read_p(pf[(t + 1) & 1], t + 1, j * PPG + q);
            }
            if constexpr (WDMA) {
                if (dma_step >= 0 && g < NPW) dma_piece(dma_step, bbuf ^ 1, g);
            }
            __builtin_amdgcn_sched_barrier(0);      // reads of the next group stay AHEAD of this group's MFMAs
            const PF &pc = pf[t & 1];
            const WF &wc = wf[g & 1];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (NS == 2) acc[j][m] = mfma_m16<F16>(wc.h, pc.l[m], acc[j][m]);
                if (NB == 2) acc[j][m] = mfma_m16<F16>(wc.l, pc.h[m], acc[j][m]);
                acc[j][m] = mfma_m16<F16>(wc.h, pc.h[m], acc[j][m]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    halo_load(0);
    halo_store();
    if constexpr (WDMA) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma_piece(0, 0, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_b(0);
        store_b(0);
        if (T > 1) load_b(1);
    }
    __syncthreads();
    int bbuf = 0;
#pragma unroll 1
    for (int step = 0; step < T; ++step) {
        const int cb = step / KS, r = step - cb * KS;
        const bool more = step + 1 < T;
        const bool boundary = more && r == KS - 1;
        if constexpr (!WDMA) {
            if (more) store_b(bbuf ^ 1);              // weights of step+1 (registers loaded during the previous step)
            if (step + 2 < T) load_b(step + 2);
        }
        if (boundary) halo_load(cb + 1);
        compute(r, bbuf, (WDMA && more) ? step + 1 : -1);
        if (boundary) {
            __syncthreads();                          // every wave is done with the halo
            halo_store();
        }
        if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step + 1 have landed
        __syncthreads();
        bbuf ^= 1;
    }

    // epilogue: lane -> pixel (lane & 15) of pixel tile m, channels 4 * (lane >> 4) .. + 3 of channel tile j
    const float nslope = p.act == HOIG_ACT_NONE ? 1.f : (p.act == HOIG_ACT_RELU ? 0.f : p.slope);
    const bool special = p.act == HOIG_ACT_TANH || p.act == HOIG_ACT_SIGMOID;
    float4 bias_r[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = n0 + wn * (NTW * 16) + j * 16 + lg * 4;
        bias_r[j] = (gbias && n < p.N) ? *reinterpret_cast<const float4 *>(gbias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float st1[NTW][4], st2[NTW][4];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) st1[j][q] = st2[j][q] = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int oy = y0 + wm * 2 + (m >> 1), ox = x0 + (m & 1) * 16 + l15;
        const size_t pix = ((size_t)bi * p.H + oy) * p.W + ox;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = n0 + wn * (NTW * 16) + j * 16 + lg * 4;
            if (n < p.N) {
                float v[4];
                const float bq[4] = {bias_r[j].x, bias_r[j].y, bias_r[j].z, bias_r[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fast_act(acc[j][m][q] * p.oscale + bq[q], nslope, special, p.act, p.slope);
                if (gadd) {
                    const float4 ad = *reinterpret_cast<const float4 *>(gadd + pix * p.N + n);
                    v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    st1[j][q] += v[q];
                    st2[j][q] += v[q] * v[q];
                }
                const float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (!p.C2) *reinterpret_cast<float4 *>(gC + pix * p.N + n) = o;
                else if (n < p.n1) *reinterpret_cast<float4 *>(gC + pix * p.n1 + n) = o;      // (a 64-column group goes one way)
                else *reinterpret_cast<float4 *>(p.C2 + pix * (p.N - p.n1) + (n - p.n1)) = o;
            }
        }
    }
    if (p.stats) m16_stats_epilogue<NTW, WM, BN, NT>(st1, st2, smem, p.stats + (size_t)b * 2 * p.N, p.N, n0, wm, wn, lane, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 3x3 layers (Conv2d s2 p1 and ConvTranspose2d s2 p1 op1, forward and data gradient) on the 16x16 MFMA: the parity-phase
// decomposition, step table and tile of conv_halo_s2_bf16_kernel (conv_igemm_bf16.hip: 4 x 32 coarse pixels x BN channels, four
// waves, steps of <= 2 taps over a (4+1) x 33 halo image, two workgroups per CU), with the LDS images and the swapped operand
// roles of conv_halo3_m16_kernel above (a lane ends up with four consecutive channels of one pixel: 16-B stores, also strided
// over the fine grid in scatter mode).
template <int NSX, int BN, bool SCATTER, bool F16>
__global__ __launch_bounds__(256) void conv_halo_s2_m16_kernel(const HaloArgs p) {
    constexpr int NS = NSX == 1 ? 1 : 2, NB = NSX == 2 ? 2 : 1;
    constexpr int TH = 4, TW = 32, NT = 256, WN = 2;
    constexpr int RB = BN * 4 / NT;
    constexpr int HH = TH + 1, HW = TW + 1, HPIX = HH * HW;
    constexpr int PHALF = HPIX * 32, P23 = round128(PHALF) + 64, PLANE_P = round128(P23 + PHALF);
    constexpr int W23 = BN * 32 + 64, PLANE_W = round128(W23 + BN * 32);
    constexpr int MT = 4, NTW = BN / (16 * WN);
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * PLANE_P + 2 * NB * PLANE_W];
    unsigned char *Ph = smem, *Pl = smem + PLANE_P;
    unsigned char *Wbase = smem + NS * PLANE_P;            // two tap tiles of (Wh, Wl)

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

    const int brow = tid >> 2, bpos = tid & 3;
    const unsigned short *wrow_h[RB], *wrow_l[RB];
    int woff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = brow + (NT / 4) * i;
        const int n = n0 + row;
        const size_t o = ((size_t)(n >> 5) * (p.K >> 5)) * 1024 + (n & 31) * 32 + bpos * 8;
        wrow_h[i] = n < p.N ? p.Wh + o : nullptr;
        wrow_l[i] = (NB == 2 && n < p.N) ? p.Wl + o : nullptr;
        const int c = bpos ^ ((row >> 2) & 3);
        woff[i] = (c >> 1) * W23 + row * 32 + (c & 1) * 16;
    }
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

    // ---- step table (wave-uniform scalar code; see conv_halo_s2_bf16_kernel) ----
    struct Step {
        int cb, ntap, tap[2], pp, qq;
        bool load;
    };
    const int spc = SCATTER ? ((P ? 2 : 1) * (Q ? 2 : 1) + 1) / 2 : 5;     // steps per 32-channel block
    const int ncb = p.Cg >> 5, T = ncb * spc;
    auto step_info = [&](int step) -> Step {
        Step s_;
        s_.cb = step / spc;
        const int idx = step - s_.cb * spc;
        s_.pp = s_.qq = 0;
        if (!SCATTER) {
            s_.load = idx != 1;
            switch (idx) {
                case 0: s_.pp = 1; s_.qq = 1; s_.ntap = 2; s_.tap[0] = 0; s_.tap[1] = 2; break;
                case 1: s_.pp = 1; s_.qq = 1; s_.ntap = 2; s_.tap[0] = 6; s_.tap[1] = 8; break;
                case 2: s_.pp = 1; s_.qq = 0; s_.ntap = 2; s_.tap[0] = 1; s_.tap[1] = 7; break;
                case 3: s_.pp = 0; s_.qq = 1; s_.ntap = 2; s_.tap[0] = 3; s_.tap[1] = 5; break;
                default: s_.ntap = 1; s_.tap[0] = 4; s_.tap[1] = 4; break;
            }
        } else {
            s_.load = idx == 0;
            const int r0 = P ? 0 : 1, r1 = 2, s0 = Q ? 0 : 1, s1 = 2;
            if (P && Q) {
                s_.ntap = 2;
                s_.tap[0] = (idx ? r1 : r0) * 3 + s0;
                s_.tap[1] = (idx ? r1 : r0) * 3 + s1;
            } else if (P) {
                s_.ntap = 2; s_.tap[0] = r0 * 3 + s0; s_.tap[1] = r1 * 3 + s0;
            } else if (Q) {
                s_.ntap = 2; s_.tap[0] = r0 * 3 + s0; s_.tap[1] = r0 * 3 + s1;
            } else {
                s_.ntap = 1; s_.tap[0] = s_.tap[1] = 4;
            }
        }
        return s_;
    };
    // halo offset (rows, columns in {0,1}) of tap (r,s):  gather: (r != 0, s != 0)   scatter: (r == 0, s == 0)
    auto tap_off = [&](int tap) -> int {
        const int r = tap / 3, s_ = tap - r * 3;
        const int dr = SCATTER ? (r == 0) : (r != 0), dc = SCATTER ? (s_ == 0) : (s_ != 0);
        return (dr * HW + dc) * 32;
    };

    uint4 rbh[2][RB], rbl[2][RB];
    auto load_b = [&](const Step &s_) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const size_t koff = (size_t)(s_.tap[t] * p.Cg + s_.cb * 32) * 32;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                rbh[t][i] = wrow_h[i] ? *reinterpret_cast<const uint4 *>(wrow_h[i] + koff) : make_uint4(0, 0, 0, 0);
                if (NB == 2)
                    rbl[t][i] = wrow_l[i] ? *reinterpret_cast<const uint4 *>(wrow_l[i] + koff) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned char *Wh = Wbase + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                *reinterpret_cast<uint4 *>(Wh + woff[i]) = rbh[t][i];
                if (NB == 2) *reinterpret_cast<uint4 *>(Wl + woff[i]) = rbl[t][i];
            }
        }
    };
    const float *Aimg = p.A + (size_t)b * p.H * p.W * p.Cg;       // p.H x p.W: the gathered tensor (fine grid in gather mode)
    constexpr int HSLICES = (HPIX * 8 + NT - 1) / NT;
    float4 hregs[2][HSLICES];                                     // halo images fetched TWO steps ahead, alternating sets
    auto halo_load = [&](const Step &s_, float4 (&hreg)[HSLICES]) {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int gy = SCATTER ? y0 + hy : 2 * (y0 - 1 + hy) + s_.pp;
                const int gx = SCATTER ? x0 + hx : 2 * (x0 - 1 + hx) + s_.qq;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v = *reinterpret_cast<const float4 *>(Aimg + ((size_t)gy * p.W + gx) * p.Cg + s_.cb * 32 + c4 * 4);
            }
            hreg[sl] = v;
        }
    };
    auto halo_store = [&](const float4 (&hreg)[HSLICES]) {
#pragma unroll
        for (int sl = 0; sl < HSLICES; ++sl) {
            const int i = tid + NT * sl;
            if (i < HPIX * 8) {
                const int pix = i >> 3, c4 = i & 7;
                uint2 hi, lo;
                split4t<F16>(hreg[sl], hi, lo);
                const int off = (c4 >> 2) * P23 + pix * 32 + (c4 & 3) * 8;
                *reinterpret_cast<uint2 *>(Ph + off) = hi;
                if (NS == 2) *reinterpret_cast<uint2 *>(Pl + off) = lo;
            }
        }
    };
    auto compute = [&](const Step &s_) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t >= s_.ntap) break;
            const int tapoff = tap_off(s_.tap[t]);
            const unsigned char *Wh = Wbase + t * NB * PLANE_W, *Wl = Wh + PLANE_W;
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

    Step cur = step_info(0);
    halo_load(cur, hregs[0]);
    load_b(cur);
    halo_store(hregs[0]);
    store_b();
    Step nxt = cur;
    if (T > 1) {
        nxt = step_info(1);
        if (nxt.load) halo_load(nxt, hregs[1]);
    }
    __syncthreads();
    auto one_step = [&](int step, float4 (&mine)[HSLICES], float4 (&other)[HSLICES]) {
        const bool more = step + 1 < T;
        if (more) load_b(nxt);
        if (step + 2 < T) {
            const Step n2 = step_info(step + 2);
            if (n2.load) halo_load(n2, mine);             // stored at the end of step+1
        }
        compute(cur);
        if (more) {
            __syncthreads();                  // every wave has finished reading the weight tiles (and the halo)
            if (nxt.load) halo_store(other);  // fetched during step-1
            store_b();
            __syncthreads();
            cur = nxt;
            if (step + 2 < T) nxt = step_info(step + 2);
        }
    };
#pragma unroll 1
    for (int step = 0; step < T; step += 2) {
        one_step(step, hregs[0], hregs[1]);
        if (step + 1 < T) one_step(step + 1, hregs[1], hregs[0]);
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
    if (p.stats) m16_stats_epilogue<NTW, 2, BN, NT>(st1, st2, smem, p.stats + (size_t)b * 2 * p.N, p.N, n0, wm, wn, lane, tid);
}

template <int NS, int BN>
int launch_one(const HaloArgs &a, hipStream_t st) {
    constexpr int WM = 4, WN = 2;
    constexpr size_t shm = M16Layout<WM, BN>::bytes(NS);
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    if (a.a_split) {                                       // pre-split gathered tensor: bf16 data-gradient launches without x split
        if (a.in_scale) return HOIG_EUNSUPPORTED;
        if constexpr (NS == 2) return HOIG_EUNSUPPORTED;
        else {
            if (a.f16 || a.A2) return HOIG_EUNSUPPORTED;
            static hoig_once once_s;
            if (!once_s.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                    return HOIG_ELAUNCH;
                once_s.set();
            }
            if (hoig_tuning(HOIG_TUNE_WDMA16) != 0 && a.N % BN == 0) {
                static hoig_once once_sd;
                if (!once_sd.done()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, false, true, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                        return HOIG_ELAUNCH;
                    once_sd.set();
                }
                conv_halo3_m16_kernel<NS, WM, WN, BN, false, true, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
            } else
                conv_halo3_m16_kernel<NS, WM, WN, BN, false, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
            HOIG_LAUNCH_CHECK();
            return HOIG_OK;
        }
    }
    if (a.in_scale) {                                      // inference chain: norm + ReLU of the gathered tensor applied in the loader
        if (!a.f16 || a.N % BN || !a.in_shift || a.b_split > 0) return HOIG_EUNSUPPORTED;
        static hoig_once once_n;
        if (!once_n.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, true, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                return HOIG_ELAUNCH;
            once_n.set();
        }
        conv_halo3_m16_kernel<NS, WM, WN, BN, true, false, true, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    if (hoig_tuning(HOIG_TUNE_WDMA16) != 0 && a.N % BN == 0) {       // weight tiles by LDS-DMA (whole channel tiles only)
        static hoig_once once_d;
        if (!once_d.done()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, true, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_halo3_m16_kernel<NS, WM, WN, BN, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                return HOIG_ELAUNCH;
            once_d.set();
        }
        if (a.f16) conv_halo3_m16_kernel<NS, WM, WN, BN, true, false, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
        else conv_halo3_m16_kernel<NS, WM, WN, BN, false, false, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
        HOIG_LAUNCH_CHECK();
        return HOIG_OK;
    }
    if (a.f16) conv_halo3_m16_kernel<NS, WM, WN, BN, true><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
    else conv_halo3_m16_kernel<NS, WM, WN, BN, false><<<a.nblk, 64 * WM * WN, shm, st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

template <bool SCATTER>
int launch_s2(const HaloArgs &a, int ns, hipStream_t st) {
    const bool n64 = (a.N % 128) != 0;
    // conv_s2_16.hip, the statically walked form (loads in flight behind the MFMAs, weight tiles by LDS-DMA), where it measured
    // faster (key s2_pipe = 1; profiles/r05_s2_dma_ab.txt):
    //  * 8 x 32 tiles, eight waves, one workgroup per CU, wherever that still gives every CU a workgroup and N is a multiple of 128:
    //    a weight tile serves 256 pixels -- a third fewer bytes through the CU's load path per MFMA, the path that bounds these
    //    layers: 12-24 % faster on every such launch of the step;
    //  * grids of at most one workgroup per CU: 4 x 32 tiles, 15-30 % faster;
    //  * in between and for 64-channel tiles: the kernel below (two workgroups per CU fill each other's stalls there; the static walk
    //    on 4 x 32 tiles is 0-40 % SLOWER with two workgroups on a CU).
    // 2 / 3: the 4-row / 8-row form wherever it can run (tests, A/B); 0: never.
    const int pipe = hoig_tuning(HOIG_TUNE_S2_PIPE);
    const bool rows8_ok = a.nblk >= 512 && !(a.tiles_y & 1) && a.N % 128 == 0;
    if (rows8_ok && (pipe == 1 || pipe == 3)) return launch_halo_s2_m16p(a, ns, SCATTER, true, st);
    if (pipe == 2 || (pipe != 0 && a.nblk <= 256)) return launch_halo_s2_m16p(a, ns, SCATTER, false, st);
    if (n64) {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_s2_m16_kernel<NSX, 64, SCATTER, true><<<a.nblk, 256, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_s2_m16_kernel<NSX, 64, SCATTER, false><<<a.nblk, 256, 0, st>>>(a));
    } else {
        if (a.f16) HOIG_NS_SWITCH(ns, conv_halo_s2_m16_kernel<NSX, 128, SCATTER, true><<<a.nblk, 256, 0, st>>>(a));
        else HOIG_NS_SWITCH(ns, conv_halo_s2_m16_kernel<NSX, 128, SCATTER, false><<<a.nblk, 256, 0, st>>>(a));
    }
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace

// the stride-2 3x3 layers: `a` arrives with the geometry launch_halo_s2 (conv_igemm_bf16.hip) computes
int launch_halo_s2_m16(const HaloArgs &a, int ns, bool scatter, hipStream_t st) {
    if (a.N % 64 || a.Cg % 32) return HOIG_EUNSUPPORTED;
    return scatter ? launch_s2<true>(a, ns, st) : launch_s2<false>(a, ns, st);
}

// `a` arrives with the geometry of an 8-row tiling filled in (tiles_x / tiles_y / nblk_n / nblk / nmajor) for channel tiles
// of `bn` = 128 or 64
int launch_halo3_m16(HaloArgs a, int ns, int bn, hipStream_t st) {
    if (a.H % 8 || a.W % 32 || a.Cg % 32 || a.N % 4) return HOIG_EUNSUPPORTED;
    if (a.b_split > 0 && (a.A2 || a.C2 || a.stats || a.b_split >= a.Bn)) return HOIG_EUNSUPPORTED;
    if (bn == 128) HOIG_NS_SWITCH(ns, return launch_one<NSX, 128>(a, st));
    if (bn == 64) HOIG_NS_SWITCH(ns, return launch_one<NSX, 64>(a, st));
    return HOIG_EUNSUPPORTED;
}

}  // namespace hoig_detail
