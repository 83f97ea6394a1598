// Probe: what a WRITE-ONLY kernel reaches on this device, by store width and shape -- the bound of the PolyExp launches
// (fb_prep_polyexp_mfma_kernel writes 80 KB per 64 x 64 image as 8-byte and 4-byte stores, 256 / 128 contiguous bytes per
// half wave).  256 persistent workgroups of 512 threads, 0.36 GB per launch (4 224 images x 80 KB + change), 20 launches.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/store_rate.hip -o tools/probes/_build/store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: 16 B per lane, a wave instruction covers 1 KB contiguous
// MODE 1: 8 B per lane, a half wave covers 256 B contiguous, the two halves 4 KB apart (the (c0, c1) / (c2, c3) planes' shape)
// MODE 2: 4 B per lane, a half wave 128 B, halves 2 KB apart (the c4 plane's shape)
// MODE 3: the PolyExp epilogue's mix per image: waves 0-3 MODE 1 shape on two planes, waves 4-7 MODE 2 on the third
template <int MODE>
__global__ __launch_bounds__(512) void fill_kernel(float* __restrict__ out, long long n_img, int ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (long long im = blockIdx.x; im < n_img; im += gridDim.x) {
    float* d = out + im * 20480;      // 80 KB per image
    const float v = (float)im;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 10; ++i) *reinterpret_cast<f4*>(d + (i * 512 + tid) * 4) = (f4){v, v, v, v};
    } else if (MODE == 1) {
      // 8 waves x 20 instructions x 512 B
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        const int row = (wave * 20 + i) * 2 + half;      // 320 rows of 256 B
        *reinterpret_cast<f2*>(d + row * 64 + col * 2) = (f2){v, v};
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 40; ++i) {
        const int row = (wave * 40 + i) * 2 + half;      // 640 rows of 128 B
        d[row * 32 + col] = v;
      }
    } else if (MODE == 3) {
      const int sm = wave & 3, strip = sm & 1, mbo = sm >> 1;
      if (wave < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int y = 32 * strip + 8 * (r >> 2) + 4 * half + (r & 3), x = 32 * mbo + col;
          *reinterpret_cast<f2*>(d + (y * 64 + x) * 2) = (f2){v, v};
          *reinterpret_cast<f2*>(d + 8192 + (y * 64 + x) * 2) = (f2){v, v};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int y = 32 * strip + 8 * (r >> 2) + 4 * half + (r & 3), x = 32 * mbo + col;
          d[16384 + y * 64 + x] = v;
        }
      }
    }
    // ws > 0: a stand-in for the compute phases between two epilogues (no memory traffic): ws x ~1 000 cycles, then a barrier
    for (int i = 0; i < ws; ++i) __builtin_amdgcn_s_sleep(16);
    if (ws > 0) __syncthreads();
  }
}

template <int MODE>
static void run(const char* name, float* buf, long long n_img, int ws = 0) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_kernel<MODE>, dim3(256), dim3(512), 0, 0, buf, n_img, ws);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fill_kernel<MODE>, dim3(256), dim3(512), 0, 0, buf, n_img, ws);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 20;
  printf("%-58s ws %2d %7.1f us  %5.2f TB/s   %6.2f us per image and CU\n", name, ws, ms * 1e3, n_img * 81920.0 / ms / 1e9, ms * 1e3 / (n_img / 256.0));
}

int main() {
  const long long n_img = 4224 + 256;
  float* buf;
  hipMalloc(&buf, n_img * 81920);
  run<0>("16 B per lane, 1 KB per wave instruction", buf, n_img);
  run<1>("8 B per lane, 2 x 256 B per wave instruction", buf, n_img);
  run<2>("4 B per lane, 2 x 128 B per wave instruction", buf, n_img);
  run<3>("PolyExp epilogue's mix (8 B pairs + 4 B plane)", buf, n_img);
  // MODE 4 = no stores at all: what the stand-in alone takes
  for (int ws : {4, 8, 12}) {
    run<4>("no stores", buf, n_img, ws);
    run<0>("16 B per lane", buf, n_img, ws);
    run<3>("PolyExp epilogue's mix", buf, n_img, ws);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemsetAsync(buf, 0, n_img * 81920, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipMemsetAsync(buf, 0, n_img * 81920, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.1f us  %5.2f TB/s\n", "hipMemsetAsync", ms / 20 * 1e3, n_img * 81920.0 / (ms / 20) / 1e9);
  return 0;
}
