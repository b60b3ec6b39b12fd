// Weight gradient of the stride-1 "same" 3x3 convolutions from PRE-SPLIT dy, staged by LDS-DMA (round 5).
//
// wgrad_halo_bf16_kernel (conv_igemm_bf16.hip) pulls fp32 dy through registers: every workgroup loads its 64-KB dy tile, splits
// it to bf16 hi/lo on the VALU and stores it to LDS -- a tile that fifteen other workgroups (the other ci blocks) load and split
// as well; its waves spend a third of a tile's cycles issuing those loads, 11 % splitting and storing, and the MFMA pipes are
// busy 40 % (profiles/r03_pmc_dominant_wgrad.json).  Here dy ARRIVES split: its producer (the backward of the norm that follows
// the convolution; hoig_split_planes_bf16 for anything else) writes, per pixel, [hi: Co bf16][lo: Co bf16] -- the same 4 B per
// element as fp32, and exactly the two values the in-kernel split makes, so the sums are bit for bit those of the register
// kernel -- and a workgroup copies its tile global -> LDS with global_load_lds_dwordx4: no staging registers, no VALU, no
// ds_write.  With no registers tied to a tile in flight the dy tile is DOUBLE-buffered (2 x 64 KB): the DMA of tile t + 1 is
// issued before the MFMAs of tile t and lands behind them; one barrier per tile instead of two.
//
// Tiling as the TH = 4, CM = 2 form of wgrad_halo_bf16_kernel: a workgroup owns dW[128 co][9 taps][32 ci] over a range of pixel
// tiles (4 rows x 32 pixels); twelve waves = (32-channel group of co) x (tap row), 48 accumulator registers each.
//
// LDS image of a dy plane: [128 pixels][256 B] UNPADDED (a DMA wave-instruction writes 1 KB = four pixel rows contiguously, so rows
// cannot be padded apart), the 16-B chunk c of pixel p stored at chunk position c ^ (4 * (p & 3)): the four consecutive pixels a
// 32-lane half of ds_read_b64_tr_b16 reads then sit in four different 64-B bank columns (conflict-free, as the 320-B rows of the
// register kernel were).  The permutation is applied on the SOURCE address of the DMA (lane l of a piece fetches chunk
// (l & 15) ^ (4 * (l >> 4)) of pixel 4 * piece + (l >> 4)) and again on the read address -- cdna_hip_programming.md rule 21.
// The x halo (6 x 34 pixels x 32 ci, 26 KB of fp32) stays on the register path -- it is a quarter of the bytes, rounded to ONE bf16
// plane -- but is double-buffered too, converted and stored at the END of a tile, beside nobody's MFMAs.
#include "conv_bf16_common.h"
#include "tuning.h"
#include <type_traits>

namespace hoig_detail {
namespace {

constexpr int D_TH = 4, D_TW = 32, D_NPX = D_TH * D_TW, D_BM = 128, D_BC = 32, D_NT = 768, D_KS = 3;
constexpr int D_HH = D_TH + 2, D_HWID = D_TW + 2, D_HPIX = D_HH * D_HWID;      // 6 x 34 halo pixels
constexpr int D_QSTR = 64;                                                    // halo rows: 32 ci x 2 B, unpadded (4 rows = 64 banks)
constexpr int D_PLANE = D_NPX * 256;                                          // one dy plane of a tile: 32 KB
constexpr int D_QBUF = ((D_HPIX * D_QSTR + 255) / 256) * 256;                 // 13 056 B
constexpr int wgrad_dma_lds(int ns) { return 2 * ns * D_PLANE + 2 * D_QBUF; } // 157 184 B with two dy planes

template <int NS>      // dy planes: 2 (hi + lo: HOIG_PREC_F16X2) or 1 (hi only: HOIG_PREC_BF16)
__global__ __launch_bounds__(D_NT) void wgrad_dma_kernel(const WHaloArgs p) {
    constexpr int TH = D_TH, TW = D_TW, BM = D_BM, BC = D_BC, NT = D_NT, KS = D_KS, HWID = D_HWID, HPIX = D_HPIX, QSTR = D_QSTR;
    constexpr int PBUF = NS * D_PLANE;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char *const Pbase = smem, *const Qbase = smem + 2 * PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3, tr = wave >> 2;                // 32-channel group of co, tap row
    const int tile = hoig_xcd_remap(blockIdx.x, p.nblk);
    const int c0 = (tile / p.nblk_ci) * BM, ci0 = (tile % p.nblk_ci) * BC;
    const int mt_begin = blockIdx.y * p.mt_per_split;
    const int mt_end = min(p.n_mtiles, mt_begin + p.mt_per_split);
    const size_t pstr = (size_t)2 * p.Co;                   // pixel stride of the split tensor, in bf16 units
    // grouped launch: this workgroup's tiles all lie in one of the two problems (the launcher aligns the tile ranges); its tensors are
    // addressed from image b_split
    const bool g2 = p.b_split > 0 && mt_begin >= p.b_split * p.tiles_x * p.tiles_y;
    const int b_off = g2 ? p.b_split : 0;
    const unsigned short *const DYS = reinterpret_cast<const unsigned short *>(g2 ? p.DY_g2 : p.DY) -
                                      (size_t)b_off * p.H * p.W * pstr;                    // split: [pixel][2][Co] bf16
    float *const DWg = g2 ? p.DW_g2 : p.DW;

    // ---- dy: LDS-DMA.  Piece j of a plane = pixels 4j .. 4j+3 (1 KB); lane l: pixel 4j + (l >> 4), LDS chunk l & 15 <- source chunk
    // (l & 15) ^ (4 * (l >> 4)).  The lane's share of the address is the same for every piece.
    // The DMA is issued from inline asm: through __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) cannot tell the buffer being filled
    // from the one being read and puts `s_waitcnt vmcnt(0)` in front of the tile's first ds_read -- the copy would land BEFORE the MFMAs
    // instead of behind them.  An asm LDS-DMA has no register destination; its completion is counted by hand (the vmcnt(0) that
    // closes a tile, below).  M0 = the wave-uniform LDS byte address of the piece (cdna_hip_programming.md 5.7).
    const size_t lane_src = (size_t)(lane >> 4) * pstr + (size_t)(((lane & 15) ^ (4 * (lane >> 4))) * 8);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    // piece i of this wave (plane-major: j = wave + 12 i over the NS * 32 pieces of a tile) of the tile at `base`, into buffer bsel
    constexpr int NPIECE = (NS * 32 + 11) / 12;
    auto tile_base = [&](int b, int ty, int tx) -> const unsigned short * {
        return DYS + (((size_t)b * p.H + ty * TH) * p.W + tx * TW) * pstr + c0 + lane_src;
    };
    auto dma_piece = [&](const unsigned short *base, int bsel, int i) {
        const int j = wave + 12 * i;                        // (wave-uniform: the branch is scalar)
        if (j < NS * 32) {
            const int plane = j >> 5, jj = j & 31;          // tile row jj >> 3, pixels 4 * (jj & 7) ..
            const unsigned short *src = base + (size_t)plane * p.Co + ((size_t)(jj >> 3) * p.W + 4 * (jj & 7)) * pstr;
            const unsigned to = __builtin_amdgcn_readfirstlane(lds0 + bsel * PBUF + plane * D_PLANE + jj * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(to) : "memory");
        }
    };

    // ---- x halo: registers -> one bf16 plane
    constexpr int QSL = (HPIX * 8 + NT - 1) / NT;           // float4s per thread (3)
    float4 rq[QSL];
    // The loads are UNCONDITIONAL (an out-of-image position reads element 0 of its image and is zeroed when the tile is stored): a load
    // under the bounds branch merges with the zero of the other path, and hipcc resolved that merge with register copies -- and an
    // `s_waitcnt vmcnt(0)` -- right behind the load.  Everything that does not change from tile to tile is computed ONCE: this thread's
    // halo positions (slice i -> row / column of the 6 x 34 halo), the channel offset; a tile contributes three scalars (image, first
    // row, first column: stepped, not divided out of the tile index -- an integer division is ~30 VALU instructions, and the VALU
    // shares its issue port with the MFMAs of the two other waves of the SIMD: the first version's address arithmetic cost 14 us).
    int hyc[QSL], hxc[QSL];
#pragma unroll
    for (int i = 0; i < QSL; ++i) {
        const int idx = tid + NT * i, hp = idx >> 3;
        hyc[i] = idx < HPIX * 8 ? hp / HWID : (1 << 24);    // (a slice past the halo fails every bounds test)
        hxc[i] = hp - (hp / HWID) * HWID;
    }
    const bool second = p.X2 != nullptr && ci0 >= p.ci1;
    const int ldx = p.X2 ? (second ? p.Ci - p.ci1 : p.ci1) : p.Ci;
    const size_t ximg = (size_t)p.Hin * p.Win * ldx;
    const float *const xc = (second ? p.X2 : (g2 ? p.X_g2 : p.X)) - (size_t)b_off * ximg + (second ? ci0 - p.ci1 : ci0) + (tid & 7) * 4;
    struct Tile { int b, ty, tx; };
    auto tile_of = [&](int mt) -> Tile {
        const int tx = mt % p.tiles_x, t2 = mt / p.tiles_x;
        return Tile{t2 / p.tiles_y, t2 % p.tiles_y, tx};
    };
    auto tile_next = [&](Tile t) -> Tile {                  // (scalar: every operand is wave-uniform)
        if (++t.tx == p.tiles_x) {
            t.tx = 0;
            if (++t.ty == p.tiles_y) {
                t.ty = 0;
                ++t.b;
            }
        }
        return t;
    };
    unsigned xin = 0;                                       // bit i: slice i of the tile in rq[] lies inside the image
    auto load_x = [&](const Tile t) {
        const float *img = xc + (size_t)t.b * ximg;
        const int y0 = t.ty * TH - p.pad, x0 = t.tx * TW - p.pad;
        xin = 0;
#pragma unroll
        for (int i = 0; i < QSL; ++i) {
            const int gy = y0 + hyc[i], gx = x0 + hxc[i];
            const bool in = (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
            xin |= in ? (1u << i) : 0u;
            rq[i] = *reinterpret_cast<const float4 *>(img + (size_t)(in ? (gy * p.Win + gx) * ldx : 0));
        }
    };
    auto store_x = [&](int bsel) {
        unsigned char *Q = Qbase + bsel * D_QBUF;
#pragma unroll
        for (int i = 0; i < QSL; ++i) {
            const int idx = tid + NT * i;
            // (pins the conversion HERE: hipcc otherwise moves it up behind the load and waits for the data there)
            asm volatile("" : "+v"(rq[i].x), "+v"(rq[i].y), "+v"(rq[i].z), "+v"(rq[i].w));
            if (idx < HPIX * 8)
                *reinterpret_cast<uint2 *>(Q + idx * 8) =
                    (xin >> i) & 1u ? make_uint2(cvt2(rq[i].x, rq[i].y), cvt2(rq[i].z, rq[i].w)) : make_uint2(0u, 0u);
        }
    };

    // ---- transpose-read addressing (wgrad_halo_bf16_kernel): 16-lane group g, lane 4q + c -> row 8 * (g >> 1) + q, channels
    // 16 * (g & 1) + 4c of the wave's 32
    const int grp = lane >> 4, li = lane & 15;
    const int trow = (grp >> 1) * 8 + (li >> 2);
    const int tch = ((grp & 1) * 16 + (li & 3) * 4) * 2;                        // byte offset inside the 64 B of a 32-channel group
    // dy: pixel row (256 B) + swizzled chunk ((p & 3) == li >> 2 in both halves of a fragment) + the 8-B half of the chunk
    const int trP = trow * 256 + ((4 * (cb ^ (li >> 2)) + (tch >> 4)) << 4) + (tch & 8);
    const int trQ = (trow + tr * HWID) * QSTR + tch;
    typedef short s4_t __attribute__((ext_vector_type(4)));
    auto frag = [&](const unsigned char *a, int stride4) -> bf16x8 {
        const s4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)a);
        const s4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a + stride4));
        bf16x8 f;
        f[0] = lo4[0]; f[1] = lo4[1]; f[2] = lo4[2]; f[3] = lo4[3];
        f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
        return f;
    };

    f32x16 acc[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    Tile tnext = tile_of(mt_begin < mt_end ? mt_begin : 0);
    if (mt_begin < mt_end) {
        const unsigned short *b0 = tile_base(tnext.b, tnext.ty, tnext.tx);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(b0, 0, i);
        load_x(tnext);
        store_x(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // One tile; `nxt` (is there a next tile to stage?) is a compile-time constant of each call: written as a run-time test inside one
    // loop body, hipcc's wait-count pass sees a path "x loads issued, store_x skipped" and guards every later write to those registers
    // with s_waitcnt vmcnt(0) -- behind a DMA just issued, i.e. a full memory round trip in three k-steps of every tile.
    auto tile_body = [&](const int mt, auto nxt_c) {
        constexpr bool nxt = decltype(nxt_c)::value;
        const int cur = (mt - mt_begin) & 1;
        // the next tile's DMA pieces are issued ONE PER K-STEP below, not in a burst here: twelve waves pushing 64 KB of requests into the
        // CU's one load path at once stall each other at issue (round 2's stamps of the register kernel: a third of a tile's cycles),
        // and a wave that waits at issue feeds no MFMAs; with no registers tied to a DMA in flight nothing forces the burst
        const unsigned short *nb = nullptr;
        if constexpr (nxt) {
            tnext = tile_next(tnext);
            nb = tile_base(tnext.b, tnext.ty, tnext.tx);
            load_x(tnext);                                  // (the quarter of the bytes that goes through registers: up front)
        }
        auto stage = [&](int kk) {                          // k-step kk carries DMA piece kk of the next tile
            if constexpr (nxt) {
                if (kk < NPIECE) dma_piece(nb, cur ^ 1, kk);
            }
        };
        const unsigned char *Ph = Pbase + cur * PBUF + trP, *Pl = Ph + D_PLANE;
        const unsigned char *Qh = Qbase + cur * D_QBUF + trQ;
        struct KFrag { bf16x8 ah, al, bh[KS]; };
        auto read_k = [&](KFrag &f, int kk) {               // k-step kk: 16 consecutive pixels of tile row kk >> 1
            f.ah = frag(Ph + kk * 4096, 1024);
            if (NS == 2) f.al = frag(Pl + kk * 4096, 1024);
            const int qrow0 = (kk >> 1) * HWID + (kk & 1) * 16;
#pragma unroll
            for (int t = 0; t < KS; ++t) f.bh[t] = frag(Qh + (qrow0 + t) * QSTR, 4 * QSTR);
        };
        auto mma_k = [&](const KFrag &f) {
            if (NS == 2) {
#pragma unroll
                for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < KS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[t], acc[t], 0, 0, 0);
        };
        // the fragments of k-step kk + 1 are read before the MFMAs of k-step kk issue (fences: see wgrad_halo_bf16_kernel)
        KFrag f0, f1;
        read_k(f0, 0);
#pragma unroll
        for (int kk = 0; kk < TH * 2; kk += 2) {
            read_k(f1, kk + 1);
            stage(kk);
            __builtin_amdgcn_sched_barrier(0);
            mma_k(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < TH * 2) read_k(f0, kk + 2);
            stage(kk + 1);
            // the next tile's x halo is converted and stored BESIDE the last k-steps' MFMAs of the other waves, not between the last MFMA and
            // the barrier (its buffer was last read in the previous tile, a barrier ago; its loads were issued at the top of this tile and
            // the last DMA piece a k-step ago, so the wait in front of the conversion finds everything landed)
            if constexpr (nxt) {
                if (kk == TH * 2 - 2) store_x(cur ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_k(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (nxt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's DMA pieces of tile mt + 1 have landed ...
            __syncthreads();                                // ... and so have everyone's; everyone is done reading tile mt
        }
    };
    for (int mt = mt_begin; mt + 1 < mt_end; ++mt) tile_body(mt, std::true_type{});
    if (mt_begin < mt_end) tile_body(mt_end - 1, std::false_type{});

    const int l31 = lane & 31, lh = lane >> 5;
    const int K = KS * KS * p.Ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = c0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float *row = DWg + (size_t)co * K + (tr * KS) * p.Ci + ci0 + l31;
#pragma unroll
        for (int t = 0; t < KS; ++t) atomicAdd(row + t * p.Ci, acc[t][r]);
    }
}

// fp32 [npix][C] -> split [npix][2][C] bf16 (hi = bf16(v), lo = bf16(v - hi): split4 of conv_bf16_common.h), 16-B accesses
__global__ __launch_bounds__(256) void split_planes_kernel(const float *__restrict__ x, unsigned short *__restrict__ out, int64_t n4,
                                                           int C4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t px = i / C4;
        const int c4 = (int)(i - px * C4);
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        uint2 hi, lo;
        split4(v, hi, lo);
        uint2 *o = reinterpret_cast<uint2 *>(out) + px * (2 * C4) + c4;
        o[0] = hi;
        o[C4] = lo;
    }
}

// split [npix][2][C] bf16 -> fp32 [npix][C] = hi + lo (what every 16-bit consumer of the split tensor multiplies with)
__global__ __launch_bounds__(256) void unsplit_planes_kernel(const unsigned short *__restrict__ in, float *__restrict__ out, int64_t n4,
                                                             int C4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t px = i / C4;
        const int c4 = (int)(i - px * C4);
        const uint2 *s = reinterpret_cast<const uint2 *>(in) + px * (2 * C4) + c4;
        const uint2 hi = s[0], lo = s[C4];
        float4 v;
        v.x = __uint_as_float(hi.x << 16) + __uint_as_float(lo.x << 16);
        v.y = __uint_as_float(hi.x & 0xFFFF0000u) + __uint_as_float(lo.x & 0xFFFF0000u);
        v.z = __uint_as_float(hi.y << 16) + __uint_as_float(lo.y << 16);
        v.w = __uint_as_float(hi.y & 0xFFFF0000u) + __uint_as_float(lo.y & 0xFFFF0000u);
        reinterpret_cast<float4 *>(out)[i] = v;
    }
}

}  // namespace

int launch_wgrad_dma(const WHaloArgs &a, int ns, dim3 grid, hipStream_t st) {
    if (a.tout || a.DB || ns == 2 || a.W % 32 || a.H % 4 || a.Ci % 32 || a.Co % 128 || a.pad != 1 || a.Hin != a.H || a.Win != a.W)
        return HOIG_EUNSUPPORTED;
    if (a.b_split > 0 && (a.X2 || a.b_split >= a.Bn || (a.b_split * a.tiles_x * a.tiles_y) % a.mt_per_split)) return HOIG_EUNSUPPORTED;
    static hoig_once once;
    if (!once.done()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_dma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                wgrad_dma_lds(2)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_dma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                wgrad_dma_lds(1)) != hipSuccess)
            return HOIG_ELAUNCH;
        once.set();
    }
    if (ns == 3) wgrad_dma_kernel<2><<<grid, D_NT, wgrad_dma_lds(2), st>>>(a);
    else wgrad_dma_kernel<1><<<grid, D_NT, wgrad_dma_lds(1), st>>>(a);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

}  // namespace hoig_detail

extern "C" int hoig_split_planes_bf16(const float *x, uint16_t *out, int64_t npix, int C, hoig_stream_t stream) {
    if (!x || !out || npix < 0 || C <= 0 || (C & 3)) return HOIG_EINVAL;
    if (npix == 0) return HOIG_OK;
    const int64_t n4 = npix * (C / 4);
    hoig_detail::split_planes_kernel<<<hoig_stream_grid(n4, 256), 256, 0, (hipStream_t)stream>>>(x, out, n4, C / 4);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}

extern "C" int hoig_unsplit_planes_bf16(const uint16_t *in, float *out, int64_t npix, int C, hoig_stream_t stream) {
    if (!in || !out || npix < 0 || C <= 0 || (C & 3)) return HOIG_EINVAL;
    if (npix == 0) return HOIG_OK;
    const int64_t n4 = npix * (C / 4);
    hoig_detail::unsplit_planes_kernel<<<hoig_stream_grid(n4, 256), 256, 0, (hipStream_t)stream>>>(in, out, n4, C / 4);
    HOIG_LAUNCH_CHECK();
    return HOIG_OK;
}
