// Hardware check of ds_read_b64_tr_b16 lane semantics used by the bf16 wgrad kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
#define ROWSTRIDE 160   // in bf16 elements (320 B)
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short T[32 * ROWSTRIDE];   // T[k][co], value = k*100+co
  for (int i = threadIdx.x; i < 32 * ROWSTRIDE; i += 64) { int kk = i / ROWSTRIDE, co = i % ROWSTRIDE; T[i] = kk * 100 + co; }
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const int co_base = 16 * (g & 1), k_base = 8 * (g >> 1);
  // lane 4q+p supplies the address of row q (k), columns 4p..4p+3
  const short* a0 = &T[(k_base + q) * ROWSTRIDE + co_base + 4 * p];
  const short* a1 = &T[(k_base + 4 + q) * ROWSTRIDE + co_base + 4 * p];
  s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)a0);
  s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)a1);
  for (int e = 0; e < 4; ++e) { out[l * 8 + e] = v0[e]; out[l * 8 + 4 + e] = v1[e]; }
}
int main() {
  short* d; hipMalloc(&d, 64 * 8 * 2); k<<<1, 64>>>(d); short h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
    int row = l & 31, hh = l >> 5;               // MFMA A operand: A[row][k = 8*hh + j]
    int expect = (8 * hh + j) * 100 + row;       // = T[k][co=row]
    if (h[l * 8 + j] != expect) { if (bad < 10) printf("lane %d j %d got %d expect %d\n", l, j, h[l*8+j], expect); ++bad; }
  }
  printf("tr16 check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
