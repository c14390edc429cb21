// probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds element index e at 16-bit slot e.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lp;
__global__ void k(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((char*)sm + addr[threadIdx.x]));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = t[e];
}
int main() {
  int h_addr[64]; short h_out[256];
  // rows of 128 elements (256 B); lane 4q+p of each 16-lane group g: row = 4g + q, cols 4p..4p+3
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3; h_addr[l] = ((4 * g + q) * 128 + 4 * p) * 2; }
  int* d_addr; short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d_addr, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d addr(row %2d col %2d):", l, h_addr[l] / 256, (h_addr[l] % 256) / 2);
    for (int e = 0; e < 4; ++e) printf("  (r%d,c%d)", h_out[l * 4 + e] / 128, h_out[l * 4 + e] % 128);
    printf("\n");
  }
  return 0;
}
