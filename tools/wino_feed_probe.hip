// Feed-rate probe for DESIGN.md section 3i (round 6): how long do the OPERAND BYTES of one dominant launch (3x3 512->512 at 32x32, 16 images) take
// to arrive in the CUs, with nothing else in the kernel -- (a) in the shipped direct kernel's pattern, (b) in the fused Winograd
// F(2x2,3x3) kernel's pattern?  No MFMA, no LDS reads, no stores: a lower bound of what either kernel's load path costs.
//   (a) direct:   256 workgroups (256 px x 128 ch tiles), per 32-channel block 147 KB of weight tiles (9 taps x 128 ch x 32 ci x hi/lo,
//                 shared by the 64 workgroups of a channel tile) by LDS-DMA + 44 KB of fp32 halo (its own) by LDS-DMA, 16 blocks
//   (b) Winograd: 512 workgroups (T = 64 tiles = 16x16 px, Ct = 64 ch), per 32-channel block 131 KB of transformed weights (16 positions
//                 x 64 ch x 32 ci x hi/lo, shared by the 64 workgroups of a channel tile; each of the 8 waves loads the 16 KB of ITS two
//                 positions straight into registers: they do not fit LDS beside V) + 41 KB of fp32 halo (18 x 18 px) by LDS-DMA
// Build + run (on the GPU box):  hipcc -O3 --offload-arch=gfx950 tools/wino_feed_probe.hip -o /tmp/wino_feed_probe && /tmp/wino_feed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void *gptr;
typedef __attribute__((address_space(3))) void *lptr;

// MODE 0: direct pattern, MODE 1: Winograd pattern.  W: the (transformed) weight planes, X: the fp32 activations.
template <int MODE>
__global__ __launch_bounds__(512) void feed_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, unsigned *__restrict__ sink,
                                                   int n_pix_tiles, int kblocks) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // blocks b and b + 8 share an XCD: give each XCD a contiguous chunk of the tile order, as the shipped kernels do (hoig_xcd_remap),
    // so that the workgroups of one channel tile read its weights through ONE L2
    const int nblk = gridDim.x, q = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int nt = tile / n_pix_tiles, pt = tile % n_pix_tiles;                  // channel tile, pixel tile (pixel tiles of a channel tile adjacent)
    constexpr int W_PER_KB = MODE == 0 ? 147456 : 131072;                       // weight bytes per k-block of one channel tile
    constexpr int H_PER_KB = MODE == 0 ? 44 * 1024 : 41 * 1024 + 512;           // halo bytes per k-block of one pixel tile (rounded to 512 B)
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(W) + (size_t)nt * kblocks * W_PER_KB;
    const unsigned char *xbase = reinterpret_cast<const unsigned char *>(X) + (size_t)pt * kblocks * H_PER_KB;
    uint4 acc = make_uint4(0, 0, 0, 0);
    // Both patterns are software-pipelined one stage deep, as the real kernels are: the loads of stage s + 1 are issued BEFORE the wave
    // waits for stage s (vmcnt counts a wave's loads in order), one barrier per stage.
    if (MODE == 0) {
        // stage = one tap row of one 32-channel block: 48 KB of weight tiles into alternating buffers; the block's halo image (44 KB)
        // rides with its first stage
        const int hp = (H_PER_KB / 1024 - wave + 7) / 8;                          // halo pieces of this wave: 6 or 5
        auto issue = [&](int st) {
            const int kb = st / 3, step = st - kb * 3;
            const unsigned char *w = wbase + (size_t)kb * W_PER_KB + step * 49152;
            unsigned char *buf = smem + (st & 1) * 49152;
            if (step == 0) {
                const unsigned char *x = xbase + (size_t)kb * H_PER_KB;
                for (int p = wave; p < H_PER_KB / 1024; p += 8)
                    __builtin_amdgcn_global_load_lds((gptr)(x + p * 1024 + lane * 16), (lptr)(smem + 2 * 49152 + p * 1024), 16, 0, 0);
            }
            for (int p = wave; p < 48; p += 8)
                __builtin_amdgcn_global_load_lds((gptr)(w + p * 1024 + lane * 16), (lptr)(buf + p * 1024), 16, 0, 0);
        };
        const int nst = kblocks * 3;
        issue(0);
        for (int st = 0; st < nst; ++st) {
            if (st + 1 < nst) {
                issue(st + 1);
                const int nxt = 6 + (((st + 1) % 3 == 0) ? hp : 0);             // loads of stage st + 1 may stay in flight
                if (nxt == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (nxt == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // stage = one 32-channel block: 128 KB of U + 41 KB of halo = 169 one-KB pieces.  In the kernel this stands for, U would go to
        // VGPRs (it does not fit LDS beside V); for the RATE that makes no difference (MI355X_MICROARCH.md 'Indexed rows': LDS-DMA and
        // register staging read at the same rate), and hipcc drains the DMA queue at every use of a plain load's result, which would
        // un-pipeline the probe -- so everything goes by LDS-DMA here, into an 80-KB window that is simply overwritten (nobody reads it).
        constexpr int NP = 128 + H_PER_KB / 1024;                                 // 169 pieces per stage
        const int mine = (NP - wave + 7) / 8;                                    // 22 (wave 0) or 21
        auto issue = [&](int kb) {
            const unsigned char *w = wbase + (size_t)kb * W_PER_KB, *x = xbase + (size_t)kb * H_PER_KB;
            for (int p = wave; p < NP; p += 8) {
                const unsigned char *src = p < 128 ? w + p * 1024 : x + (p - 128) * 1024;
                __builtin_amdgcn_global_load_lds((gptr)(src + lane * 16), (lptr)(smem + (p % 80) * 1024), 16, 0, 0);
            }
        };
        issue(0);
        for (int kb = 0; kb < kblocks; ++kb) {
            if (kb + 1 < kblocks) {
                issue(kb + 1);
                if (mine == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    if (kblocks < 0) sink[tid] = acc.x + smem[tid];
}

template <int MODE>
static void run(const char *name, int wgs, int n_pix_tiles, size_t lds) {
    const int kblocks = 16;
    const size_t wbytes = (size_t)(wgs / n_pix_tiles) * kblocks * (MODE == 0 ? 147456 : 131072);
    const size_t xbytes = (size_t)n_pix_tiles * kblocks * (MODE == 0 ? 44 * 1024 : 41 * 1024 + 512);
    uint4 *W, *X; unsigned *sink;
    CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&X, xbytes)); CK(hipMalloc(&sink, 4096));
    std::vector<unsigned> h(wbytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(W, h.data(), wbytes, hipMemcpyHostToDevice));
    h.assign(xbytes / 4, 7u);
    CK(hipMemcpy(X, h.data(), xbytes, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&feed_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) feed_kernel<MODE><<<wgs, 512, lds, 0>>>(W, X, sink, n_pix_tiles, kblocks);
    CK(hipDeviceSynchronize());
    const int iters = 50;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) feed_kernel<MODE><<<wgs, 512, lds, 0>>>(W, X, sink, n_pix_tiles, kblocks);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double fed = (double)wgs * kblocks * ((MODE == 0 ? 147456 : 131072) + (MODE == 0 ? 44 * 1024 : 41 * 1024 + 512));
    printf("%-52s %4d workgroups  %7.1f us per launch  %6.2f GB into the CUs  %5.1f TB/s = %5.1f GB/s per CU  (unique bytes in HBM/L2: %.1f MB)\n",
           name, wgs, us, fed / 1e9, fed / us / 1e6, fed / us / 1e3 / 256, (wbytes + xbytes) / 1e6);
    CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(sink));
}

int main() {
    for (int r = 0; r < 3; ++r) {
        run<0>("direct kernel's operand stream (weights + halo by LDS-DMA)", 256, 64, 2 * 49152 + 44 * 1024);
        run<1>("fused Winograd's operand stream (U + halo, 169 KB per block)", 512, 64, 80 * 1024);
    }
    return 0;
}
