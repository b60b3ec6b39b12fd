// Hardware probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 (fp6) operands: operand lane / bit layout and the meaning of
// the E8M0 scale operand, checked with exact small integers (the forward's two cross terms on
// block-scaled fp6).  Build + run on the MI355X:  hipcc --offload-arch=gfx950 tools/mfma_f6_probe.hip -o /tmp/f6probe && /tmp/f6probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// e2m3: sign | 2-bit exponent (bias 1) | 3-bit mantissa.  exponent 0: m/8; e >= 1: (1 + m/8) * 2^(e-1).  Integers 0..7 are exact.
__host__ __device__ inline unsigned enc_e2m3(int v) {
    const unsigned s = v < 0 ? 32u : 0u;
    const int a = v < 0 ? -v : v;
    const unsigned code[8] = {0, 8, 16, 20, 24, 26, 28, 30};
    return s | code[a];
}

// hypothesis H1: lane l holds 32 consecutive k (k = 32 * (l >> 5) + j) of row / column (l & 31); element j sits in bits
// [6j, 6j + 6) of the lane's 8 dwords (the upper 2 dwords unused)
__global__ void probe(const int *a_vals, const int *b_vals, float *c_out, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    unsigned ra[8] = {0}, rb[8] = {0};
    for (int j = 0; j < 32; ++j) {
        const unsigned ea = enc_e2m3(a_vals[(l & 31) * 64 + 32 * (l >> 5) + j]);      // A[row][k]
        const unsigned eb = enc_e2m3(b_vals[(32 * (l >> 5) + j) * 32 + (l & 31)]);    // B[k][col]
        const int bit = 6 * j;
        ra[bit >> 5] |= ea << (bit & 31);
        if ((bit & 31) > 26) ra[(bit >> 5) + 1] |= ea >> (32 - (bit & 31));
        rb[bit >> 5] |= eb << (bit & 31);
        if ((bit & 31) > 26) rb[(bit >> 5) + 1] |= eb >> (32 - (bit & 31));
    }
    v8i va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (int)ra[i]; vb[i] = (int)rb[i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // cbsz = 2, blgp = 2: both operands e2m3; opsel 0: scale byte 0 of the scale VGPR
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 2, 2, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        c_out[row * 32 + col] = acc[r];
    }
}

// per-lane scales: lane l carries the scale of (row l & 31, k block l >> 5) -- checked with a different exponent per lane
__global__ void probe_lane_scales(const int *a_vals, const int *b_vals, float *c_out) {
    const int l = threadIdx.x;
    unsigned ra[8] = {0}, rb[8] = {0};
    for (int j = 0; j < 32; ++j) {
        const unsigned ea = enc_e2m3(a_vals[(l & 31) * 64 + 32 * (l >> 5) + j]);
        const unsigned eb = enc_e2m3(b_vals[(32 * (l >> 5) + j) * 32 + (l & 31)]);
        const int bit = 6 * j;
        ra[bit >> 5] |= ea << (bit & 31);
        if ((bit & 31) > 26) ra[(bit >> 5) + 1] |= ea >> (32 - (bit & 31));
        rb[bit >> 5] |= eb << (bit & 31);
        if ((bit & 31) > 26) rb[(bit >> 5) + 1] |= eb >> (32 - (bit & 31));
    }
    v8i va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (int)ra[i]; vb[i] = (int)rb[i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int sa = 127 + ((l & 31) % 3) + (l >> 5);        // A: 2^(row % 3 + kblock)
    const int sb = 127 - ((l & 31) % 2);                   // B: 2^-(col % 2)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 2, 2, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) c_out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

// issue rate: independent accumulators, one wave per SIMD (256 CUs x 4), back to back
template <int F6>
__global__ void rate(float *out, int iters) {
    v8i va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = threadIdx.x * 7 + i; vb[i] = threadIdx.x * 13 + i; }
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(float)(threadIdx.x + i); hb[i] = (__bf16)(float)(threadIdx.x - i); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (F6) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[k], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[k], 0, 0, 0);
        }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    int ha[32 * 64], hb[64 * 32];
    srand(7);
    for (int i = 0; i < 32 * 64; ++i) ha[i] = rand() % 7 - 3;
    for (int i = 0; i < 64 * 32; ++i) hb[i] = rand() % 7 - 3;
    int *da, *db; float *dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 32 * 32 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    float hc[32 * 32];
    const int cfg[3][2] = {{0x7F7F7F7F, 0x7F7F7F7F}, {0x7F7F7F80, 0x7F7F7F7F}, {0x7F7F7F7F, 0x7F7F7F7E}};
    const float expect_scale[3] = {1.f, 2.f, 0.5f};
    for (int t = 0; t < 3; ++t) {
        probe<<<1, 64>>>(da, db, dc, cfg[t][0], cfg[t][1]);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float ref = 0.f;
            for (int k = 0; k < 64; ++k) ref += (float)(ha[i * 64 + k] * hb[k * 32 + j]);
            ref *= expect_scale[t];
            if (hc[i * 32 + j] != ref) { if (bad < 6) printf("  cfg %d C[%d][%d] = %g expect %g\n", t, i, j, hc[i * 32 + j], ref); ++bad; }
        }
        printf("cfg %d (scale_a %#x scale_b %#x, expect x%g): %s (%d mismatches)\n", t, cfg[t][0], cfg[t][1], expect_scale[t],
               bad ? "MISMATCH" : "OK", bad);
    }
    {
        probe_lane_scales<<<1, 64>>>(da, db, dc);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float ref = 0.f;
            for (int kb = 0; kb < 2; ++kb) {
                float part = 0.f;
                for (int k = 32 * kb; k < 32 * kb + 32; ++k) part += (float)(ha[i * 64 + k] * hb[k * 32 + j]);
                ref += part * (float)(1 << ((i % 3) + kb)) / (float)(1 << (j % 2));
            }
            if (hc[i * 32 + j] != ref) { if (bad < 6) printf("  lane scales C[%d][%d] = %g expect %g\n", i, j, hc[i * 32 + j], ref); ++bad; }
        }
        printf("per-lane scales (A: row, k block; B: column): %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    }
    {
        float *dout; hipMalloc(&dout, 1024 * 64 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        for (int f6 = 0; f6 < 2; ++f6) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (f6) rate<1><<<1024, 64>>>(dout, iters); else rate<0><<<1024, 64>>>(dout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double k = f6 ? 64.0 : 16.0;
            const double flops = 1024.0 * iters * 4 * 2.0 * 32 * 32 * k;
            printf("%s: %.3f ms for %d x 4 MFMAs per wave, 1024 waves -> %.0f TFLOP/s (zeros-like operands, one wave per SIMD)\n",
                   f6 ? "mfma_scale 32x32x64 e2m3" : "mfma 32x32x16 bf16     ", ms, iters, flops / ms / 1e9);
        }
    }
    return 0;
}
