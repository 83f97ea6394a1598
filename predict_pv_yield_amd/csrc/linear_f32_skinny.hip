// fc1 of the f32 model (F.linear with K = 1 003 520 input features, <= 32 rows, <= 128 outputs; reference:
// predict_pv_yield/models/conv3d/model.py:92-103,125-130 in float32): forward and input gradient as streams over the 513 MB weight
// with EXACT f32 products on v_mfma_f32_32x32x2_f32 (round 5).
//
// Both were calls of the generic split-product GEMM (gemm_bf16x3: 300 + 129 us of split-K reduce, 348 us): 1.5 TB/s on a matrix
// that is read once.  Now: forward 128-138 + 5 us (LDS-staged; straight from memory: 290 + 22), dx 145-158 us.  The work is 8.2 GFLOP per pass -- 52 us of the f32 matrix instruction spread over 1 024 waves -- so a
// kernel that keeps the loads coming is bound by the weight's bytes, and needs no operand splitting at all.
//   forward  y[m][n] = sum_k x[m][k] w[n][k]: wave = 32 outputs n, lane (n, h) loads 16 bytes = k0 + 4h .. +3 of its row, and the
//            four matrix instructions of a load take element t of every lane (the k order inside a group of 8 is permuted the
//            same way on both operands); a workgroup owns a contiguous range of k and leaves a [32][n] slab, summed in slab order.
//   dx[m][k] = sum_n g[m][n] w[n][k]: wave = 128 columns k, lane (j, h) loads w[2i + h][k0 + 4j .. +3] (512 contiguous bytes per
//            half wave), accumulator t holds column k0 + 4j + t: 16-byte stores; g (32 x 128) lives in 64 registers per lane.
#include "pv_common.h"
#include <stdlib.h>

namespace pv {

// D[m][n] of a 32x32 accumulator: lane = n + 32 hi, register r -> m
__device__ __forceinline__ int ls_acc_row(int r, int hi) { return 8 * (r >> 2) + 4 * hi + (r & 3); }

constexpr int LS_U = 8;      // 16-byte loads in flight per operand and lane

__global__ __launch_bounds__(256) void linear_fwd_f32_skinny_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     float* __restrict__ slabs, int m, int n, long long k,
                                                                     long long k_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  // the workgroups take 64-column groups in turn (group g -> workgroup g % grid).  Measured: 288-293 us = 2.2 TB/s whichever way
  // k is dealt out (contiguous ranges per workgroup, interleaved groups) and with plain or non-temporal loads -- a lane fetches
  // 32-byte pieces of its own row, a load instruction touches 32 lines, and the 16 KB vector cache does not hold a line for the
  // four instructions that share it (the dx kernel below, 512 contiguous bytes per half wave, streams the same matrix at
  // 4.4 TB/s).  Staging the weight tile through LDS from row-contiguous loads is the next form.
  const long long kstep = (long long)gridDim.x * 8 * LS_U;
  (void)k_per_wg;
  const int nrow = 32 * wave + r32;
  const bool n_ok = nrow < n, m_ok = r32 < m;
  const float* wr = w + (size_t)(n_ok ? nrow : 0) * k + 4 * h;
  const float* xr = x + (size_t)(m_ok ? r32 : 0) * k + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (32 * wave < n) {
    for (long long k0 = (long long)blockIdx.x * 8 * LS_U; k0 < k; k0 += kstep) {      // (k is a multiple of 8 LS_U: the launcher's condition)
      f32x4 wv[LS_U], xv[LS_U];
#pragma unroll
      for (int u = 0; u < LS_U; ++u) {
        wv[u] = n_ok ? *reinterpret_cast<const f32x4*>(wr + k0 + 8 * u) : (f32x4){0.f, 0.f, 0.f, 0.f};
        xv[u] = m_ok ? *reinterpret_cast<const f32x4*>(xr + k0 + 8 * u) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < LS_U; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u][t], wv[u][t], acc, 0, 0, 0);
      }
    }
    float* s = slabs + (size_t)blockIdx.x * 32 * n;
    if (n_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ls_acc_row(r, h) * n + nrow] = acc[r];
    }
  }
}

// The forward through LDS: a 128-column tile of w (128 rows) and x (32 rows) arrives as ROW-CONTIGUOUS 16-byte loads (a wave
// instruction = 512 bytes of two rows; the tile after next is already in registers while this one is multiplied) and is read
// back as operand fragments from rows pitched 132 floats apart (conflict-free 16-byte reads).  One workgroup of 8 waves per CU:
// wave (nb, kh) multiplies output block nb over half kh of the tile and leaves its own slab.
constexpr int LF_KT = 128, LF_PITCH = LF_KT + 4;
__global__ __launch_bounds__(512) void linear_fwd_f32_skinny_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         float* __restrict__ slabs, int m, int n, long long k,
                                                                         long long n_tiles) {
  __shared__ __attribute__((aligned(16))) float wl[128 * LF_PITCH];
  __shared__ __attribute__((aligned(16))) float xl[32 * LF_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = wave & 3, kh = wave >> 2, r32 = lane & 31, h = lane >> 5;
  // loader role: chunk c = tid & 31 (16 bytes) of rows tid >> 5 + 16 q (w: q < 8; x: q < 2)
  const int lc = tid & 31, lr = tid >> 5;
  f32x4 wreg[8], xreg[2];
  auto fetch = [&](long long tile) {
    const long long k0 = tile * LF_KT + 4 * lc;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = lr + 16 * q;
      wreg[q] = row < n ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + (size_t)row * k + k0)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = lr + 16 * q;
      xreg[q] = row < m ? *reinterpret_cast<const f32x4*>(x + (size_t)row * k + k0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(wl + (lr + 16 * q) * LF_PITCH + 4 * lc) = wreg[q];
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(xl + (lr + 16 * q) * LF_PITCH + 4 * lc) = xreg[q];
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  long long tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();      // the previous tile's fragments have been read
    park();
    __syncthreads();
    if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);      // in flight under this tile's products
    const float* wf = wl + (32 * nb + r32) * LF_PITCH + 64 * kh + 4 * h;
    const float* xf = xl + r32 * LF_PITCH + 64 * kh + 4 * h;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wf + 8 * u);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xf + 8 * u);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t], wv[t], acc, 0, 0, 0);
    }
  }
  float* sl = slabs + ((size_t)blockIdx.x * 2 + kh) * 32 * n;
  const int nrow = 32 * nb + r32;
  if (nrow < n) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sl[ls_acc_row(r, h) * n + nrow] = acc[r];
  }
}

// y[m][n] = sum of the slabs + bias, ReLU.  An element's slabs are summed by 8 threads, a contiguous eighth of them each in slab
// order, and the eight partial sums are added in order: the same bits every time (one thread per element walked 512 slabs: 21 us).
__global__ __launch_bounds__(256) void linear_fwd_f32_skinny_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                                                            float* __restrict__ y, int m, int n, int n_slabs, int relu) {
  __shared__ float part[8][32];
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  const bool ok = e < m * n;
  const int mi = ok ? e / n : 0, ni = ok ? e - mi * n : 0;
  const float* s = slabs + mi * n + ni;
  const size_t stride = (size_t)32 * n;
  const int per = (n_slabs + 7) / 8, lo = grp * per, hi = min(lo + per, n_slabs);
  float acc = 0.f;
  int i = lo;
  for (; i + 8 <= hi; i += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[(size_t)(i + j) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __fadd_rn(acc, v[j]);
  }
  for (; i < hi; ++i) acc = __fadd_rn(acc, s[(size_t)i * stride]);
  part[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && ok) {
    float t = part[0][el];
#pragma unroll
    for (int g = 1; g < 8; ++g) t = __fadd_rn(t, part[g][el]);
    if (bias) t = __fadd_rn(t, bias[ni]);
    y[e] = relu ? fmaxf(t, 0.f) : t;
  }
}

__global__ __launch_bounds__(256) void linear_dx_f32_skinny_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                    float* __restrict__ dx, int m, int n, long long k,
                                                                    long long n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  // A operand: lane (m = j, h) supplies g[m][2 i + h] to the instructions of row pair i
  float gr[64];
  const int npairs = (n + 1) >> 1;
#pragma unroll
  for (int i = 0; i < 64; ++i) gr[i] = (j < m && i < npairs && 2 * i + h < n) ? g[j * n + 2 * i + h] : 0.f;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
    const long long k0 = tile * 128 + 4 * j;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const float* wp = w + (size_t)h * k + k0;
#pragma unroll
    for (int i0 = 0; i0 < 64; i0 += LS_U) {
      f32x4 wv[LS_U];
#pragma unroll
      for (int u = 0; u < LS_U; ++u) {
        const int row = 2 * (i0 + u) + h;
        wv[u] = row < n ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + (size_t)(2 * (i0 + u)) * k)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < LS_U; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[i0 + u], wv[u][t], acc[t], 0, 0, 0);
      }
    }
    // accumulator t, lane (j, hi), register r: dx[ls_acc_row(r, hi)][k0 + t]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mi = ls_acc_row(r, h);
      if (mi < m) *reinterpret_cast<f32x4*>(dx + (size_t)mi * k + k0) = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    }
  }
}

}  // namespace pv

using namespace pv;

extern "C" {

// 1 when the two entry points below take the shape
int pv_linear_f32_skinny_covers(int32_t m, int32_t n, int64_t k) {
  return (m > 0 && m <= 32 && n > 0 && n <= 128 && k >= (1 << 16) && k % 128 == 0) ? 1 : 0;
}

size_t pv_linear_fwd_f32_skinny_workspace_bytes(int32_t n) { return (size_t)2 * kNumCU * 32 * (size_t)n * sizeof(float); }

int pv_linear_fwd_f32_skinny(const float* x, const float* w, const float* bias, float* y, int32_t m, int32_t n, int64_t k, int32_t relu,
                             void* workspace, size_t workspace_bytes, void* stream) {
  PV_REQUIRE(x && w && y && workspace, PV_EINVAL, "pv_linear_fwd_f32_skinny: null pointer");
  PV_REQUIRE(pv_linear_f32_skinny_covers(m, n, k), PV_ESIZE, "pv_linear_fwd_f32_skinny: m <= 32, n <= 128, k %% 128 == 0, k >= 65 536");
  PV_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, PV_EINVAL, "pv_linear_fwd_f32_skinny: 16-byte aligned x and w");
  hipStream_t st0 = as_stream(stream);
  if (!getenv("PV_LINEAR_F32_SKINNY_DIRECT")) {      // (the direct-from-memory form below: the cross-check, 290 us on fc1)
    const long long n_tiles = k / LF_KT;
    const long long n_wg = std::min<long long>(n_tiles, kNumCU);
    PV_REQUIRE(workspace_bytes >= (size_t)n_wg * 2 * 32 * n * sizeof(float), PV_ESIZE, "pv_linear_fwd_f32_skinny: workspace too small");
    hipLaunchKernelGGL(linear_fwd_f32_skinny_lds_kernel, dim3((unsigned)n_wg), dim3(512), 0, st0, x, w, (float*)workspace, m, n,
                       (long long)k, n_tiles);
    hipLaunchKernelGGL(linear_fwd_f32_skinny_reduce_kernel, dim3((unsigned)((m * n + 31) / 32)), dim3(256), 0, st0,
                       (const float*)workspace, bias, y, m, n, (int)(2 * n_wg), relu ? 1 : 0);
    return check_launch("pv_linear_fwd_f32_skinny");
  }
  // k ranges of whole 64-column groups, at most two workgroups per CU
  const long long groups = k / (8 * LS_U);
  long long n_wg = std::min<long long>(groups, 2 * kNumCU);
  const long long per = (groups + n_wg - 1) / n_wg;
  n_wg = (groups + per - 1) / per;
  PV_REQUIRE(workspace_bytes >= (size_t)n_wg * 32 * n * sizeof(float), PV_ESIZE, "pv_linear_fwd_f32_skinny: workspace too small");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(linear_fwd_f32_skinny_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, x, w, (float*)workspace, m, n, (long long)k,
                     per * 8 * LS_U);
  hipLaunchKernelGGL(linear_fwd_f32_skinny_reduce_kernel, dim3((unsigned)((m * n + 31) / 32)), dim3(256), 0, st,
                     (const float*)workspace, bias, y, m, n, (int)n_wg, relu ? 1 : 0);
  return check_launch("pv_linear_fwd_f32_skinny");
}

int pv_linear_dx_f32_skinny(const float* g, const float* w, float* dx, int32_t m, int32_t n, int64_t k, void* stream) {
  PV_REQUIRE(g && w && dx, PV_EINVAL, "pv_linear_dx_f32_skinny: null pointer");
  PV_REQUIRE(pv_linear_f32_skinny_covers(m, n, k), PV_ESIZE, "pv_linear_dx_f32_skinny: m <= 32, n <= 128, k %% 128 == 0, k >= 65 536");
  PV_REQUIRE((((uintptr_t)dx | (uintptr_t)w) & 15) == 0, PV_EINVAL, "pv_linear_dx_f32_skinny: 16-byte aligned dx and w");
  const long long n_tiles = k / 128;
  const unsigned grid = (unsigned)std::min<long long>((n_tiles + 3) / 4, 2 * kNumCU);
  hipLaunchKernelGGL(linear_dx_f32_skinny_kernel, dim3(grid), dim3(256), 0, as_stream(stream), g, w, dx, m, n, (long long)k, n_tiles);
  return check_launch("pv_linear_dx_f32_skinny");
}

}  // extern "C"
