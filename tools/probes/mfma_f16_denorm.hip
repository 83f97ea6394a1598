// Probe: does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs (gfx950)?  And v_cvt_pk_f16_f32: rounding mode + subnormal results.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f16_denorm.hip -o tools/probes/_build/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float* cvt, const float* in) {
  const int lane = threadIdx.x;
  h8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)0.f, b[i] = (_Float16)0.f;
  // A[m = lane & 31][k = 8 (lane >> 5) + i];  B[k][n = lane & 31]
  if (lane == 0) {
    uint16_t sub = 0x0010;  // 2^-20, subnormal in f16
    _Float16 s; memcpy(&s, &sub, 2);
    a[0] = s;
    b[0] = (_Float16)1024.f;
  }
  if (lane == 1) { a[0] = (_Float16)6.103515625e-05f; /* smallest normal 2^-14 */ }
  if (lane == 0) b[1] = (_Float16)0.f;
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  // D[m][n]: lane = n + 32 * ((m >> 2) & 1), reg = (m & 3) + 4 * (m >> 3): D[0][0] = lane 0 reg 0; D[1][0] = lane 0 reg 1
  if (lane == 0) { out[0] = acc[0]; out[1] = acc[1]; }
  if (lane == 0) {
    f2 x = {in[0], in[1]};   // tie -> even for RNE (1.0), 3e-6 is f16-subnormal
    h2 h = __builtin_convertvector(x, h2);
    cvt[0] = (float)h[0]; cvt[1] = (float)h[1];
    f2 y = {in[2], in[3]};
    h2 g = __builtin_convertvector(y, h2);
    cvt[2] = (float)g[0]; cvt[3] = (float)g[1];
  }
}
int main() {
  float *d, *c, *in, h[2], hc[4];
  float hin[4] = {1.0f + 0x1p-11f, 3e-6f, 1.0f + 0x1.8p-11f, -(1.0f + 0x1.8p-11f)};
  hipMalloc(&in, 16); hipMemcpy(in, hin, 16, hipMemcpyHostToDevice);
  hipMalloc(&d, 8); hipMalloc(&c, 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, in);
  hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("subnormal A (2^-20) x 1024 -> %g (expected 2^-10 = %g if honoured, 0 if flushed)\n", h[0], 0x1p-10);
  printf("smallest normal A (2^-14) x 1024 -> %g (expected %g)\n", h[1], 0x1p-4);
  printf("cvt_pk_f16_f32: 1+2^-11 -> %.10g (RNE: 1)   3e-6 -> %.10g (subnormal kept: ~2.98e-6)   1+1.5*2^-11 -> %.10g / %.10g (RNE: 1.0009765625)\n", hc[0], hc[1], hc[2], hc[3]);
  return 0;
}
