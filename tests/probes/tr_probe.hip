// Diagnostic (not part of the product): prints which LDS element every lane receives from
// ds_read_b64_tr_b16 so the wgrad fragment maps can be checked against cdna_hip_programming.md T10.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(uint16_t* out) {
  __shared__ uint16_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;  // value = element index
  __syncthreads();
  int lane = threadIdx.x;
  int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  // rows of 32 elements (64 B): lane addresses row (4*grp + q), elements 4p..4p+3
  const uint16_t* addr = lds + (4 * grp + q) * 32 + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (row %2d,col %2d)", h[l * 4 + j] / 32, h[l * 4 + j] % 32);
    printf("\n");
  }
  return 0;
}
