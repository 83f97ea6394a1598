// Row-wise f32 kernels of the Perceiver path (perceiver_pytorch.Perceiver as instantiated by
// predict_pv_yield/models/perceiver/perceiver.py:70-80): LayerNorm (PreNorm), softmax of the attention scores, GEGLU
// (FeedForward) and the mean over latents (to_logits' Reduce), each with its backward.  One wave (or workgroup) per row;
// all of them are single streaming passes (HBM-bound), the contraction work lives in gemm_f32.hip.
#include "pv_common.h"

namespace pv {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- LayerNorm over the last dimension (d <= 256), eps inside the square root, biased variance (F.layer_norm) ----------
constexpr int LN_MAXPL = 4;  // elements per lane

__global__ __launch_bounds__(256) void layernorm_fwd_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, long long rows,
                                                          int d, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * d;
  float v[LN_MAXPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < d ? xr[c] : 0.f;
    s += v[i];
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    const float t = c < d ? v[i] - mu : 0.f;
    q += t * t;
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    if (c < d) y[row * d + c] = (v[i] - mu) * rs * w[c] + b[c];
  }
  if (lane == 0) mean[row] = mu, rstd[row] = rs;
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;  per-block partial sums of dw = sum dy * xhat, db = sum dy
__global__ __launch_bounds__(256) void layernorm_bwd_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx,
                                                          float* __restrict__ part /* [grid][2][d] */, long long rows, int d,
                                                          int rows_per_block) {
  __shared__ float red[4][2][LN_MAXPL * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dwp[LN_MAXPL], dbp[LN_MAXPL], wv[LN_MAXPL];
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    dwp[i] = 0.f, dbp[i] = 0.f;
    wv[i] = lane + 64 * i < d ? w[lane + 64 * i] : 0.f;
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (long long row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAXPL], g[LN_MAXPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPL; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < d;
      const float dv = ok ? dy[row * d + c] : 0.f;
      xh[i] = ok ? (x[row * d + c] - mu) * rs : 0.f;
      g[i] = dv * wv[i];
      s1 += g[i];
      s2 += g[i] * xh[i];
      dwp[i] += dv * xh[i];
      dbp[i] += dv;
    }
    const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
    if (dx) {
#pragma unroll
      for (int i = 0; i < LN_MAXPL; ++i) {
        const int c = lane + 64 * i;
        if (c < d) dx[row * d + c] = rs * (g[i] - m1 - xh[i] * m2);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) red[wave][0][lane + 64 * i] = dwp[i], red[wave][1][lane + 64 * i] = dbp[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += blockDim.x) {
    const int which = i / d, c = i - which * d;
    part[((size_t)blockIdx.x * 2 + which) * d + c] = ((red[0][which][c] + red[1][which][c]) + red[2][which][c]) + red[3][which][c];
  }
}

// ---- softmax(scale * x) over rows of length len; one workgroup per row; len <= 4096 keeps the row in registers ----------
constexpr int SM_PT = 16;

__global__ __launch_bounds__(256) void softmax_fwd_f32(const float* __restrict__ x, float* __restrict__ y, int len, float scale) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* xr = x + row * len;
  float* yr = y + row * len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[SM_PT];
  const bool cached = len <= 256 * SM_PT;
  float mx = -INFINITY;
  if (cached) {
#pragma unroll
    for (int i = 0; i < SM_PT; ++i) {
      const int c = threadIdx.x + 256 * i;
      v[i] = c < len ? xr[c] * scale : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
  } else {
    for (int c = threadIdx.x; c < len; c += 256) mx = fmaxf(mx, xr[c] * scale);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  if (cached) {
#pragma unroll
    for (int i = 0; i < SM_PT; ++i) {
      v[i] = expf(v[i] - mx);   // exp(-inf) = 0 for the padding
      s += v[i];
    }
  } else {
    for (int c = threadIdx.x; c < len; c += 256) s += expf(xr[c] * scale - mx);
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (((red[0] + red[1]) + red[2]) + red[3]);
  if (cached) {
#pragma unroll
    for (int i = 0; i < SM_PT; ++i) {
      const int c = threadIdx.x + 256 * i;
      if (c < len) yr[c] = v[i] * inv;
    }
  } else {
    for (int c = threadIdx.x; c < len; c += 256) yr[c] = expf(xr[c] * scale - mx) * inv;
  }
}

// dx = scale * p * (dp - sum(dp * p))
__global__ __launch_bounds__(256) void softmax_bwd_f32(const float* __restrict__ p, const float* __restrict__ dp,
                                                        float* __restrict__ dx, int len, float scale) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* pr = p + row * len;
  const float* dr = dp + row * len;
  float* xr = dx + row * len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pv_[SM_PT], dv[SM_PT];
  const bool cached = len <= 256 * SM_PT;
  float s = 0.f;
  if (cached) {
#pragma unroll
    for (int i = 0; i < SM_PT; ++i) {
      const int c = threadIdx.x + 256 * i;
      pv_[i] = c < len ? pr[c] : 0.f;
      dv[i] = c < len ? dr[c] : 0.f;
      s += pv_[i] * dv[i];
    }
  } else {
    for (int c = threadIdx.x; c < len; c += 256) s += pr[c] * dr[c];
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float dot = ((red[0] + red[1]) + red[2]) + red[3];
  if (cached) {
#pragma unroll
    for (int i = 0; i < SM_PT; ++i) {
      const int c = threadIdx.x + 256 * i;
      if (c < len) xr[c] = scale * pv_[i] * (dv[i] - dot);
    }
  } else {
    for (int c = threadIdx.x; c < len; c += 256) xr[c] = scale * pr[c] * (dr[c] - dot);
  }
}

// ---- GEGLU: y[r, c] = x[r, c] * gelu(x[r, h + c]), exact (erf) gelu ---------------------------------------------------
__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float g) {
  const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * g * g);
  return cdf + g * pdf;
}

__global__ __launch_bounds__(256) void geglu_fwd_f32(const float* __restrict__ x, float* __restrict__ y, long long rows, int h) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * h) return;
  const long long r = i / h;
  const int c = (int)(i - r * h);
  y[i] = x[r * 2 * h + c] * gelu_erf(x[r * 2 * h + h + c]);
}

__global__ __launch_bounds__(256) void geglu_bwd_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long long rows, int h) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * h) return;
  const long long r = i / h;
  const int c = (int)(i - r * h);
  const float a = x[r * 2 * h + c], g = x[r * 2 * h + h + c], d = dy[i];
  dx[r * 2 * h + c] = d * gelu_erf(g);
  dx[r * 2 * h + h + c] = d * a * gelu_erf_grad(g);
}

// ---- mean over the middle axis of [b, n, d] (to_logits: Reduce('b n d -> b d', 'mean')) -------------------------------
__global__ __launch_bounds__(256) void mean_axis1_fwd_f32(const float* __restrict__ x, float* __restrict__ y, int b, int n, int d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * d) return;
  const int bi = i / d, c = i - bi * d;
  float s = 0.f;
  for (int j = 0; j < n; ++j) s += x[((size_t)bi * n + j) * d + c];
  y[i] = s / (float)n;
}
__global__ __launch_bounds__(256) void mean_axis1_bwd_f32(const float* __restrict__ dy, float* __restrict__ dx, int b, int n, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)b * n * d) return;
  const int c = (int)(i % d);
  const int bi = (int)(i / ((long long)n * d));
  dx[i] = dy[(size_t)bi * d + c] / (float)n;
}

void launch_sum_slabs(const float* slabs, float* out, long long n, int n_slabs, long long stride, long long offset,
                      hipStream_t st, int accumulate = 0);   // gemm_f32.hip

static int ln_blocks(long long rows, int* rows_per_block) {
  long long nb = (rows + 3) / 4;          // at least one row per wave
  if (nb > 1024) nb = 1024;
  long long per = ((rows + nb - 1) / nb + 3) / 4 * 4;
  *rows_per_block = (int)per;
  return (int)((rows + per - 1) / per);
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int64_t rows,
                         int32_t d, float eps, void* stream) {
  PV_REQUIRE(x && w && b && y && mean && rstd, PV_EINVAL, "pv_layernorm_fwd_f32: null pointer");
  PV_REQUIRE(rows > 0 && d > 0 && d <= 64 * LN_MAXPL, PV_ESIZE, "pv_layernorm_fwd_f32: d=%d must be in 1..%d", d, 64 * LN_MAXPL);
  hipLaunchKernelGGL(layernorm_fwd_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, w, b, y, mean, rstd,
                     (long long)rows, d, eps);
  return check_launch("pv_layernorm_fwd_f32");
}

int pv_layernorm_bwd_workspace_bytes(int64_t rows, int32_t d, size_t* bytes) {
  PV_REQUIRE(bytes && rows > 0 && d > 0, PV_EINVAL, "pv_layernorm_bwd_workspace_bytes: bad arguments");
  int per;
  *bytes = (size_t)ln_blocks(rows, &per) * 2 * d * sizeof(float);
  return PV_OK;
}

int pv_layernorm_bwd_f32(const float* x, const float* w, const float* dy, const float* mean, const float* rstd, float* dx,
                         float* dw, float* db, int64_t rows, int32_t d, void* ws, size_t ws_bytes, int32_t accumulate,
                         void* stream) {
  PV_REQUIRE(x && w && dy && mean && rstd && dw && db, PV_EINVAL, "pv_layernorm_bwd_f32: null pointer");
  PV_REQUIRE(rows > 0 && d > 0 && d <= 64 * LN_MAXPL, PV_ESIZE, "pv_layernorm_bwd_f32: d=%d must be in 1..%d", d, 64 * LN_MAXPL);
  int per;
  const int nb = ln_blocks(rows, &per);
  PV_REQUIRE(ws && ws_bytes >= (size_t)nb * 2 * d * sizeof(float), PV_EINVAL, "pv_layernorm_bwd_f32: workspace too small");
  hipStream_t st = as_stream(stream);
  float* part = (float*)ws;
  hipLaunchKernelGGL(layernorm_bwd_f32, dim3((unsigned)nb), dim3(256), 0, st, x, w, dy, mean, rstd, dx, part, (long long)rows, d, per);
  // part is [nb][2][d]: summed over blocks in index order, straight into dw and db
  if (db == dw + d) {   // one [2 d] vector: one launch
    launch_sum_slabs(part, dw, 2 * d, nb, 2 * d, 0, st, accumulate);
  } else {
    launch_sum_slabs(part, dw, d, nb, 2 * d, 0, st, accumulate);
    launch_sum_slabs(part, db, d, nb, 2 * d, d, st, accumulate);
  }
  return check_launch("pv_layernorm_bwd_f32");
}

int pv_softmax_fwd_f32(const float* x, float* y, int64_t rows, int32_t len, float scale, void* stream) {
  PV_REQUIRE(x && y && rows > 0 && len > 0, PV_EINVAL, "pv_softmax_fwd_f32: bad arguments");
  PV_REQUIRE(rows <= 0x7fffffffll, PV_ESIZE, "pv_softmax_fwd_f32: too many rows for one launch");
  hipLaunchKernelGGL(softmax_fwd_f32, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), x, y, len, scale);
  return check_launch("pv_softmax_fwd_f32");
}

int pv_softmax_bwd_f32(const float* p, const float* dp, float* dx, int64_t rows, int32_t len, float scale, void* stream) {
  PV_REQUIRE(p && dp && dx && rows > 0 && len > 0, PV_EINVAL, "pv_softmax_bwd_f32: bad arguments");
  PV_REQUIRE(rows <= 0x7fffffffll, PV_ESIZE, "pv_softmax_bwd_f32: too many rows for one launch");
  hipLaunchKernelGGL(softmax_bwd_f32, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), p, dp, dx, len, scale);
  return check_launch("pv_softmax_bwd_f32");
}

int pv_geglu_fwd_f32(const float* x, float* y, int64_t rows, int32_t h, void* stream) {
  PV_REQUIRE(x && y && rows > 0 && h > 0, PV_EINVAL, "pv_geglu_fwd_f32: bad arguments");
  hipLaunchKernelGGL(geglu_fwd_f32, dim3((unsigned)((rows * h + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, (long long)rows, h);
  return check_launch("pv_geglu_fwd_f32");
}

int pv_geglu_bwd_f32(const float* x, const float* dy, float* dx, int64_t rows, int32_t h, void* stream) {
  PV_REQUIRE(x && dy && dx && rows > 0 && h > 0, PV_EINVAL, "pv_geglu_bwd_f32: bad arguments");
  hipLaunchKernelGGL(geglu_bwd_f32, dim3((unsigned)((rows * h + 255) / 256)), dim3(256), 0, as_stream(stream), x, dy, dx, (long long)rows, h);
  return check_launch("pv_geglu_bwd_f32");
}

int pv_mean_axis1_fwd_f32(const float* x, float* y, int32_t b, int32_t n, int32_t d, void* stream) {
  PV_REQUIRE(x && y && b > 0 && n > 0 && d > 0, PV_EINVAL, "pv_mean_axis1_fwd_f32: bad arguments");
  hipLaunchKernelGGL(mean_axis1_fwd_f32, dim3((unsigned)((b * d + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, b, n, d);
  return check_launch("pv_mean_axis1_fwd_f32");
}

int pv_mean_axis1_bwd_f32(const float* dy, float* dx, int32_t b, int32_t n, int32_t d, void* stream) {
  PV_REQUIRE(dy && dx && b > 0 && n > 0 && d > 0, PV_EINVAL, "pv_mean_axis1_bwd_f32: bad arguments");
  const long long total = (long long)b * n * d;
  hipLaunchKernelGGL(mean_axis1_bwd_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dy, dx, b, n, d);
  return check_launch("pv_mean_axis1_bwd_f32");
}

}  // extern "C"
