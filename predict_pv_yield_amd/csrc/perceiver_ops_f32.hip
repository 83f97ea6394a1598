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
                                                          int rows_per_block, const float* __restrict__ dx_add) {
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
        if (c < d) dx[row * d + c] = rs * (g[i] - m1 - xh[i] * m2) + (dx_add ? dx_add[row * d + c] : 0.f);
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

// ---- LayerNorm of SHORT rows (d <= 64, d % 8 in {2, 4, 6}): one THREAD per row, rows staged through LDS -------------------
// The wave-per-row kernels above spend ~70 instructions on a 152-byte row (d = 38: the Perceiver's context of 11 + 27
// channels, 2.5 M rows per step in experiments/003) and ran at 1.1 TB/s.  Here a workgroup copies 256 consecutive rows
// (one contiguous 256 d-float stretch) into LDS with coalesced 16-byte loads, every lane then owns one row: it reads it
// with 8-byte LDS loads (row pitch d words with d/2 odd or d/4 odd: the 32 lanes of a read group fall on distinct bank
// pairs), does the whole row arithmetic alone, writes the result back in place, and the stretch leaves with coalesced
// 16-byte stores.  gamma / beta are wave-uniform (scalar loads).  The backward accumulates d gamma / d beta per lane over the
// rows it walks and reduces them once per workgroup through LDS, in a fixed order.
template <int NT>
__device__ __forceinline__ void ln_stage_in(float* lds, const float* __restrict__ src, long long n_valid) {
  // lds[i] = src[i] for i < n_valid (<= NT * DMAX), 16 bytes per lane; src is 16-byte aligned
  for (int i4 = threadIdx.x; i4 * 4 < n_valid; i4 += NT) {
    if ((long long)i4 * 4 + 3 < n_valid) {
      *reinterpret_cast<float4*>(lds + i4 * 4) = *reinterpret_cast<const float4*>(src + (long long)i4 * 4);
    } else {
      for (int e = 0; e < 4; ++e)
        if ((long long)i4 * 4 + e < n_valid) lds[i4 * 4 + e] = src[(long long)i4 * 4 + e];
    }
  }
}
template <int NT>
__device__ __forceinline__ void ln_stage_out(const float* lds, float* __restrict__ dst, long long n_valid,
                                             const float* __restrict__ add = nullptr) {
  // dst[i] = lds[i] (+ add[i]); add, if given, is laid out and aligned like dst
  for (int i4 = threadIdx.x; i4 * 4 < n_valid; i4 += NT) {
    if ((long long)i4 * 4 + 3 < n_valid) {
      float4 v = *reinterpret_cast<const float4*>(lds + i4 * 4);
      if (add) {
        const float4 a = *reinterpret_cast<const float4*>(add + (long long)i4 * 4);
        v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
      }
      *reinterpret_cast<float4*>(dst + (long long)i4 * 4) = v;
    } else {
      for (int e = 0; e < 4; ++e)
        if ((long long)i4 * 4 + e < n_valid) dst[(long long)i4 * 4 + e] = lds[i4 * 4 + e] + (add ? add[(long long)i4 * 4 + e] : 0.f);
    }
  }
}

template <int DMAX>
__global__ __launch_bounds__(256) void layernorm_fwd_rows_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd,
                                                               long long rows, int d, float eps) {
  __shared__ __attribute__((aligned(16))) float buf[256 * DMAX];
  const long long r0 = (long long)blockIdx.x * 256;
  const long long n_valid = (rows - r0 < 256 ? rows - r0 : 256) * d;
  ln_stage_in<256>(buf, x + r0 * d, n_valid);
  __syncthreads();
  const long long row = r0 + threadIdx.x;
  if (row < rows) {
    float2* rr = reinterpret_cast<float2*>(buf + threadIdx.x * d);
    float2 v[DMAX / 2];
#pragma unroll
    for (int j = 0; j < DMAX / 2; ++j) v[j] = 2 * j < d ? rr[j] : make_float2(0.f, 0.f);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < DMAX / 2; ++j) s += v[j].x + v[j].y;
    const float mu = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < DMAX / 2; ++j)
      if (2 * j < d) {
        const float t0 = v[j].x - mu, t1 = v[j].y - mu;
        q += t0 * t0 + t1 * t1;
      }
    const float rs = 1.0f / sqrtf(q / (float)d + eps);
#pragma unroll
    for (int j = 0; j < DMAX / 2; ++j)
      if (2 * j < d) rr[j] = make_float2((v[j].x - mu) * rs * w[2 * j] + b[2 * j], (v[j].y - mu) * rs * w[2 * j + 1] + b[2 * j + 1]);
    mean[row] = mu, rstd[row] = rs;
  }
  __syncthreads();
  ln_stage_out<256>(buf, y + r0 * d, n_valid);
}

constexpr int LNR_BT = 256;   // threads = rows per chunk of the backward (128 with four workgroups per CU and 4096 blocks measured slower: the per-block column reduction and the longer slab sum cost more than the overlap gains)
template <int DMAX>
__global__ __launch_bounds__(LNR_BT) void layernorm_bwd_rows_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ dy, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx,
                                                               float* __restrict__ part /* [grid][2][d] */, long long rows, int d,
                                                               int rows_per_block, const float* __restrict__ dx_add) {
  __shared__ __attribute__((aligned(16))) float bxg[2 * LNR_BT * DMAX];   // x rows | dy rows; reused by the final reduction
  float* bx = bxg;
  float* bg = bxg + LNR_BT * DMAX;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float2 dwp[DMAX / 2], dbp[DMAX / 2];
#pragma unroll
  for (int j = 0; j < DMAX / 2; ++j) dwp[j] = make_float2(0.f, 0.f), dbp[j] = make_float2(0.f, 0.f);
  const long long rb0 = (long long)blockIdx.x * rows_per_block;
  const long long rb1 = rb0 + rows_per_block < rows ? rb0 + rows_per_block : rows;
  for (long long r0 = rb0; r0 < rb1; r0 += LNR_BT) {   // rows_per_block is a multiple of LNR_BT: chunks stay 16-byte aligned
    const long long n_valid = (rb1 - r0 < LNR_BT ? rb1 - r0 : LNR_BT) * d;
    __syncthreads();                                    // the previous chunk has left the buffers
    ln_stage_in<LNR_BT>(bx, x + r0 * d, n_valid);
    ln_stage_in<LNR_BT>(bg, dy + r0 * d, n_valid);
    __syncthreads();
    const long long row = r0 + threadIdx.x;
    if (row < rb1) {
      const float mu = mean[row], rs = rstd[row];
      const float2* xr = reinterpret_cast<const float2*>(bx + threadIdx.x * d);
      float2* gr = reinterpret_cast<float2*>(bg + threadIdx.x * d);
      float2 xh[DMAX / 2], g[DMAX / 2];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < DMAX / 2; ++j) {
        if (2 * j < d) {
          const float2 xv = xr[j], dv = gr[j];
          xh[j] = make_float2((xv.x - mu) * rs, (xv.y - mu) * rs);
          g[j] = make_float2(dv.x * w[2 * j], dv.y * w[2 * j + 1]);
          s1 += g[j].x + g[j].y;
          s2 += g[j].x * xh[j].x + g[j].y * xh[j].y;
          dwp[j].x += dv.x * xh[j].x, dwp[j].y += dv.y * xh[j].y;
          dbp[j].x += dv.x, dbp[j].y += dv.y;
        } else {
          xh[j] = make_float2(0.f, 0.f), g[j] = make_float2(0.f, 0.f);
        }
      }
      if (dx) {
        const float m1 = s1 / (float)d, m2 = s2 / (float)d;
#pragma unroll
        for (int j = 0; j < DMAX / 2; ++j)
          if (2 * j < d) gr[j] = make_float2(rs * (g[j].x - m1 - xh[j].x * m2), rs * (g[j].y - m1 - xh[j].y * m2));
      }
    }
    if (dx) {
      __syncthreads();
      ln_stage_out<LNR_BT>(bg, dx + r0 * d, n_valid, dx_add ? dx_add + r0 * d : nullptr);
    }
  }
  // column sums over the lanes: wave by wave through LDS (lane order inside a wave, then wave order: fixed)
  float* red = bx;                                     // [2 * DMAX][65]
  float tot = 0.f;
  for (int wv = 0; wv < LNR_BT / 64; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int j = 0; j < DMAX / 2; ++j) {
        red[(2 * j) * 65 + lane] = dwp[j].x, red[(2 * j + 1) * 65 + lane] = dwp[j].y;
        red[(DMAX + 2 * j) * 65 + lane] = dbp[j].x, red[(DMAX + 2 * j + 1) * 65 + lane] = dbp[j].y;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * DMAX) {
      float t = 0.f;
      for (int l = 0; l < 64; ++l) t += red[threadIdx.x * 65 + l];
      tot += t;
    }
  }
  if (threadIdx.x < 2 * DMAX) {
    const int which = threadIdx.x / DMAX, c = threadIdx.x - which * DMAX;
    if (c < d) part[((size_t)blockIdx.x * 2 + which) * d + c] = tot;
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

constexpr int LN_MAX_BLOCKS = 1024;
static int ln_blocks(long long rows, int* rows_per_block) {
  long long nb = (rows + 3) / 4;          // at least one row per wave
  if (nb > LN_MAX_BLOCKS) nb = LN_MAX_BLOCKS;
  long long per = ((rows + nb - 1) / nb + 3) / 4 * 4;
  *rows_per_block = (int)per;
  return (int)((rows + per - 1) / per);
}

// ---- d(LayerNorm weight, bias) of a normalised context straight from the gradient of its bf16 projection -----------------
// A cross-attention's context goes  x -> LayerNorm -> ctx -> to_kv (bias-free Linear) -> K | V.  When x needs no gradient (the
// images of a Perceiver), the backward of that chain needs from d(K | V) only dW_kv and (d gamma, d beta) -- d ctx = dKV W is an
// intermediate of [rows, d] floats (0.38 GB for experiments/003's 2.5 M-row context) that the two-kernel form writes (GEMM) and
// reads back together with x (LayerNorm backward).  Here a wave forms d ctx for 32 rows on the matrix cores (A = the bf16
// gradient rows as they are, B = W rounded to bf16 once, in LDS for the whole kernel: the values the one-term GEMM uses) and
// folds it in the accumulators into the column sums  d gamma[n] += d ctx[r, n] xhat[r, n],  d beta[n] += d ctx[r, n].
// Reads: 2 kdim + 4 d + 8 bytes per row.  Sums: per lane over its rows in order, then halves, waves (LDS), workgroups
// (pv_sum_slabs order) -- a fixed order.
constexpr int LPB_KMAX = 128;                 // kdim <= 128 (eight 16-deep steps)
constexpr int LPB_BRS = 2 * LPB_KMAX + 16;    // bytes per n-row of W^T in LDS (the 16-byte pad walks the banks)
template <int KSTEPS>
__global__ __launch_bounds__(256, 2) void layernorm_bwd_params_from_proj_kernel(const uint16_t* __restrict__ g16,
                                                                                const float* __restrict__ wkv,
                                                                                const float* __restrict__ x,
                                                                                const float* __restrict__ mean,
                                                                                const float* __restrict__ rstd,
                                                                                float* __restrict__ part, long long rows, int d,
                                                                                long long n_rowblocks) {
  typedef __attribute__((ext_vector_type(16))) float v16f;
  __shared__ __attribute__((aligned(16))) unsigned char Bs[64 * LPB_BRS];
  __shared__ float red[4][2][64];
  constexpr int kdim = 16 * KSTEPS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // W^T as bf16: Bs[n][k] = bf16(W[k][n]) for the Linear weight W [kdim, d] (n >= d: zero)
  for (int i = tid; i < 64 * kdim; i += 256) {
    const int n = i / kdim, k = i - n * kdim;
    const float v = n < d ? wkv[(size_t)k * d + n] : 0.f;
    *reinterpret_cast<uint16_t*>(Bs + n * LPB_BRS + 2 * k) = f32_to_bf16_bits(v);
  }
  __syncthreads();
  const int row = lane & 31, half = lane >> 5;
  const int n0 = row, n1 = 32 + row;
  float dg0 = 0.f, db0 = 0.f, dg1 = 0.f, db1 = 0.f;
  const long long stride = (long long)gridDim.x * 4;
  for (long long rb = (long long)blockIdx.x * 4 + wave; rb < n_rowblocks; rb += stride) {
    const long long r0 = rb * 32;
    // A: lane (row, half) holds k = 16 ks + 8 half .. + 7 of its row: one 16-byte load per step
    const long long ar = r0 + row < rows ? r0 + row : rows - 1;
    const u32x4* ap = reinterpret_cast<const u32x4*>(g16 + ar * kdim + 8 * half);
    u32x4 a[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = ap[2 * ks];
    // the rows' statistics and x at the accumulators' positions: row 8 (i / 4) + 4 half + i % 4, columns n0 | n1
    float mu[16], rs[16], x0[16], x1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long r = r0 + 8 * (i >> 2) + 4 * half + (i & 3);
      const bool ok = r < rows;
      const long long rc = ok ? r : rows - 1;
      mu[i] = mean[rc];
      rs[i] = ok ? rstd[rc] : 0.f;             // (a row beyond the end: xhat = 0 and its d ctx is masked below)
      x0[i] = n0 < d ? x[rc * d + n0] : 0.f;
      x1[i] = n1 < d ? x[rc * d + n1] : 0.f;
    }
    v16f acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const bf16x8 av = __builtin_bit_cast(bf16x8, a[ks]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *reinterpret_cast<const bf16x8*>(Bs + n0 * LPB_BRS + 32 * ks + 16 * half), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *reinterpret_cast<const bf16x8*>(Bs + n1 * LPB_BRS + 32 * ks + 16 * half), acc1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool ok = r0 + 8 * (i >> 2) + 4 * half + (i & 3) < rows;
      const float g0 = ok ? acc0[i] : 0.f, g1 = ok ? acc1[i] : 0.f;
      dg0 += g0 * ((x0[i] - mu[i]) * rs[i]);
      db0 += g0;
      dg1 += g1 * ((x1[i] - mu[i]) * rs[i]);
      db1 += g1;
    }
  }
  // halves of a wave (the two lanes of a column), then the four waves
  dg0 += __shfl_xor(dg0, 32), db0 += __shfl_xor(db0, 32), dg1 += __shfl_xor(dg1, 32), db1 += __shfl_xor(db1, 32);
  if (half == 0) {
    red[wave][0][n0] = dg0, red[wave][1][n0] = db0;
    red[wave][0][n1] = dg1, red[wave][1][n1] = db1;
  }
  __syncthreads();
  if (tid < 2 * d) {
    const int which = tid / d, n = tid - which * d;
    part[(size_t)blockIdx.x * 2 * d + tid] = ((red[0][which][n] + red[1][which][n]) + red[2][which][n]) + red[3][which][n];
  }
}

static int lpb_blocks(long long rows) {
  const long long rbs = (rows + 31) / 32;
  return (int)std::min<long long>((rbs + 3) / 4, 2 * kNumCU);
}


// ---- the whole backward of  x -> LayerNorm -> to_kv  for a context that takes no gradient: dW_kv, d gamma, d beta ---------
// One pass over the bf16 gradient rows dKV [rows, 128] and x [rows, d]: a workgroup stages 32 rows of each in LDS (x as the
// normalised context ctx = xhat gamma + beta rounded to bf16 -- the operand the separate weight-gradient GEMM reads from memory --
// and as xhat in f32), then wave w of the four
//   * adds  ctx^T [d, 32 rows] dKV[32 rows, columns 32 w .. 32 w + 31]  to its block of dW^T (both operands read transposed from
//     their row-major tiles: ds_read_tr16_b64), and
//   * forms the part of  d ctx = dKV W  that comes from ITS 32 columns of dKV and folds it into its own partial column sums
//     d gamma += d ctx xhat, d beta += d ctx  (the sums are linear in d ctx: no exchange between the waves).
// The next block's rows are in flight (registers) under the current block's products.  Neither ctx nor d ctx exists in memory:
// the weight-gradient GEMM (dKV and ctx read: 1.0 GB for experiments/003's context) and the kernel above (dKV and x again) become
// one pass over 1.0 GB.  Sums over the row blocks of a workgroup in order, then workgroups in index order (pv_sum_slabs).
constexpr int CBK_G_RS = 320;      // bytes per row of the dKV tile [32][128] bf16 (4 rows x 64 B walk the banks for the transposed reads)
constexpr int CBK_C_RS = 192;      // ... of the ctx tile [32][64] bf16
constexpr int CBK_XLD = 66;        // floats per row of the xhat tile [32][64 + 2]
typedef __attribute__((address_space(3))) s16x4 cbk_lds_s16x4;
__global__ __launch_bounds__(256) void context_bwd_kernel(const uint16_t* __restrict__ g16, const float* __restrict__ wkv,
                                                              const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ lnw,
                                                              const float* __restrict__ lnb, float* __restrict__ part_dw,
                                                              float* __restrict__ part_ln, long long rows, int d,
                                                              long long n_rowblocks, const float* __restrict__ x2, int d1,
                                                              int period) {
  typedef __attribute__((ext_vector_type(16))) float v16f;
  __shared__ __attribute__((aligned(16))) unsigned char Gs[32 * CBK_G_RS];
  __shared__ __attribute__((aligned(16))) unsigned char Cs[32 * CBK_C_RS];
  __shared__ __attribute__((aligned(16))) unsigned char Ws[64 * LPB_BRS];
  __shared__ float Xh[32 * CBK_XLD];
  __shared__ float red[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * 128; i += 256) {      // W^T as bf16: Ws[n][k] = bf16(W[k][n]), zero for n >= d
    const int n = i >> 7, k = i & 127;
    *reinterpret_cast<uint16_t*>(Ws + n * LPB_BRS + 2 * k) = f32_to_bf16_bits(n < d ? wkv[(size_t)k * d + n] : 0.f);
  }
  for (int i = tid; i < 32 * CBK_C_RS / 4; i += 256) reinterpret_cast<uint32_t*>(Cs)[i] = 0u;   // columns >= d stay zero
  for (int i = tid; i < 32 * CBK_XLD; i += 256) Xh[i] = 0.f;
  const int row = lane & 31, half = lane >> 5;
  const int qi = (lane & 15) >> 2, pi = lane & 3, cb = 16 * ((lane >> 4) & 1);
  // transposed operand reads (two 4 x 16-bit pieces per 16-deep step): ctx^T tiles t = 0, 1 and this wave's block of dKV
  int c_rd[2][2], g_rd[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    g_rd[s2] = (8 * half + 4 * s2 + qi) * CBK_G_RS + (32 * wave + cb + 4 * pi) * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t) c_rd[t][s2] = (8 * half + 4 * s2 + qi) * CBK_C_RS + (32 * t + cb + 4 * pi) * 2;
  }
  auto fetch_tr = [&](const unsigned char* plane, const int (&rd)[2], int ks, int rs) -> bf16x8 {
    typedef __attribute__((ext_vector_type(8))) short s16x8c;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cbk_lds_s16x4*)(plane + rd[0] + 16 * ks * rs));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cbk_lds_s16x4*)(plane + rd[1] + 16 * ks * rs));
    const s16x8c v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // staging: two 16-byte pieces of the (contiguous) dKV tile per thread; wave w brings rows 8 w .. 8 w + 7 of x, a lane per
  // column (their mean / rstd are wave-uniform: scalar loads)
  u32x4 gq[2];
  float xq[8], muq[8], rsq[8];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const float gam = lane < d ? lnw[lane] : 0.f, bet = lane < d ? lnb[lane] : 0.f;
  auto prefetch = [&](long long rb) {
    const long long r0 = rb * 32;
    const unsigned pbase = x2 ? (unsigned)((unsigned long long)r0 % (unsigned)period) : 0u;      // (period >= 32 is checked by the host)
    const uint16_t* gp = g16 + r0 * 128 + (size_t)tid * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      gq[i] = r0 + (tid >> 4) + 16 * i < rows ? *reinterpret_cast<const u32x4*>(gp + 2048 * i) : (u32x4){0u, 0u, 0u, 0u};
    // this lane's column of the 8 rows: one base address and one stride per block (two sources: see pv_context_fwd_bf16 -- the
    // lane reads either the channels or the position features, whose row index wraps at most once inside a block)
    const long long rw = r0 + 8 * wave_u;
    unsigned rp0 = pbase + 8 * wave_u;
    rp0 = rp0 >= (unsigned)period && x2 ? rp0 - (unsigned)period : rp0;
    const bool second = x2 && lane >= d1;
    const float* pl = !x2 ? x + rw * d + lane : (second ? x2 + (size_t)rp0 * (d - d1) + (lane - d1) : x + rw * d1 + lane);
    const int stride_l = !x2 ? d : (second ? d - d1 : d1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long r = rw + i;
      const bool ok = r < rows;
      const long long rc = ok ? r : rows - 1;
      muq[i] = mean[rc];
      rsq[i] = ok ? rstd[rc] : 0.f;
      const long long wrap = (second && rp0 + i >= (unsigned)period) ? (long long)period * (d - d1) : 0;
      xq[i] = (ok && lane < d) ? pl[(long long)i * stride_l - wrap] : 0.f;
    }
  };
  v16f accW[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) accW[0][i] = 0.f, accW[1][i] = 0.f;
  float dg0 = 0.f, db0 = 0.f;
  long long rb = blockIdx.x;
  if (rb < n_rowblocks) prefetch(rb);
  __syncthreads();
  for (; rb < n_rowblocks; rb += gridDim.x) {
    // registers -> tiles
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(Gs + ((tid >> 4) + 16 * i) * CBK_G_RS + 16 * (tid & 15)) = gq[i];
    if (lane < d) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 8 * wave_u + i;
        Xh[r * CBK_XLD + lane] = (xq[i] - muq[i]) * rsq[i];
        // (the LayerNorm kernel's expression; a row beyond the end has rstd = 0 here: its ctx must be 0, not beta)
        const float c = rsq[i] != 0.f ? (xq[i] - muq[i]) * rsq[i] * gam + bet : 0.f;
        *reinterpret_cast<uint16_t*>(Cs + r * CBK_C_RS + 2 * lane) = f32_to_bf16_bits(c);
      }
    }
    __syncthreads();
    if (rb + gridDim.x < n_rowblocks) prefetch(rb + gridDim.x);
    // d ctx: wave (t = w & 1, kh = w >> 1) forms columns 32 t .. 32 t + 31 from the dKV columns 64 kh .. 64 kh + 63
    v16f accD;
#pragma unroll
    for (int i = 0; i < 16; ++i) accD[i] = 0.f;
    const int dt = wave_u & 1, dkh = wave_u >> 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 gb = fetch_tr(Gs, g_rd, ks, CBK_G_RS);
      accW[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fetch_tr(Cs, c_rd[0], ks, CBK_C_RS), gb, accW[0], 0, 0, 0);
      accW[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fetch_tr(Cs, c_rd[1], ks, CBK_C_RS), gb, accW[1], 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kb = (64 * dkh + 16 * ks + 8 * half) * 2;
      accD = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Gs + row * CBK_G_RS + kb),
                                                     *reinterpret_cast<const bf16x8*>(Ws + (32 * dt + row) * LPB_BRS + kb), accD, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 8 * (i >> 2) + 4 * half + (i & 3);
      dg0 += accD[i] * Xh[r * CBK_XLD + 32 * dt + row];
      db0 += accD[i];
    }
    __syncthreads();   // the tiles are overwritten by the next block
  }
  // dW^T of this workgroup: wave w holds [n = 32 t + row_i][j = 32 w + lane % 32]
  float* pw = part_dw + (size_t)blockIdx.x * 128 * d;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = 32 * t + 8 * (i >> 2) + 4 * half + (i & 3);
      if (n < d) pw[(size_t)(32 * wave + row) * d + n] = accW[t][i];
    }
  dg0 += __shfl_xor(dg0, 32), db0 += __shfl_xor(db0, 32);
  if (half == 0) red[wave][0][row] = dg0, red[wave][1][row] = db0;      // wave w: columns 32 (w & 1) + row, dKV half w >> 1
  __syncthreads();
  if (tid < 2 * d) {
    const int which = tid / d, n = tid - which * d;
    part_ln[(size_t)blockIdx.x * 2 * d + tid] = red[n >> 5][which][n & 31] + red[2 + (n >> 5)][which][n & 31];
  }
}

static int cbk_blocks(long long rows) { return (int)std::min<long long>((rows + 31) / 32, 3 * kNumCU); }   // three workgroups share a CU (LDS, registers): all resident

}  // namespace pv

using namespace pv;

extern "C" {

int pv_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int64_t rows,
                         int32_t d, float eps, void* stream) {
  PV_REQUIRE(x && w && b && y && mean && rstd, PV_EINVAL, "pv_layernorm_fwd_f32: null pointer");
  PV_REQUIRE(rows > 0 && d > 0 && d <= 64 * LN_MAXPL, PV_ESIZE, "pv_layernorm_fwd_f32: d=%d must be in 1..%d", d, 64 * LN_MAXPL);
  const bool short_rows = d <= 64 && d % 2 == 0 && d % 8 != 0 && rows >= 65536 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
  if (short_rows && d <= 40)
    hipLaunchKernelGGL(layernorm_fwd_rows_f32<40>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), x, w, b,
                       y, mean, rstd, (long long)rows, d, eps);
  else if (short_rows)
    hipLaunchKernelGGL(layernorm_fwd_rows_f32<64>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), x, w, b,
                       y, mean, rstd, (long long)rows, d, eps);
  else
    hipLaunchKernelGGL(layernorm_fwd_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, w, b, y, mean,
                       rstd, (long long)rows, d, eps);
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
                         const float* dx_add, void* stream) {
  PV_REQUIRE(x && w && dy && mean && rstd && dw && db, PV_EINVAL, "pv_layernorm_bwd_f32: null pointer");
  PV_REQUIRE(rows > 0 && d > 0 && d <= 64 * LN_MAXPL, PV_ESIZE, "pv_layernorm_bwd_f32: d=%d must be in 1..%d", d, 64 * LN_MAXPL);
  int per;
  const int nb = ln_blocks(rows, &per);
  PV_REQUIRE(ws && ws_bytes >= (size_t)nb * 2 * d * sizeof(float), PV_EINVAL, "pv_layernorm_bwd_f32: workspace too small");
  hipStream_t st = as_stream(stream);
  float* part = (float*)ws;
  PV_REQUIRE(!dx_add || dx, PV_EINVAL, "pv_layernorm_bwd_f32: dx_add without dx");
  const bool short_rows = d <= 64 && d % 2 == 0 && d % 8 != 0 && rows >= 65536 && ((uintptr_t)x % 16 == 0) &&
                          ((uintptr_t)dy % 16 == 0) && (!dx || (uintptr_t)dx % 16 == 0) && ((uintptr_t)dx_add % 16 == 0);
  int nbl = nb;
  if (short_rows) {   // one thread per row: whole multiples of 256 rows per block (never more blocks than ln_blocks sized)
    const long long per_r = ((rows + LN_MAX_BLOCKS - 1) / LN_MAX_BLOCKS + LNR_BT - 1) / LNR_BT * LNR_BT;
    nbl = (int)((rows + per_r - 1) / per_r);
    if (d <= 40)
      hipLaunchKernelGGL(layernorm_bwd_rows_f32<40>, dim3((unsigned)nbl), dim3(LNR_BT), 0, st, x, w, dy, mean, rstd, dx, part,
                         (long long)rows, d, (int)per_r, dx_add);
    else
      hipLaunchKernelGGL(layernorm_bwd_rows_f32<64>, dim3((unsigned)nbl), dim3(LNR_BT), 0, st, x, w, dy, mean, rstd, dx, part,
                         (long long)rows, d, (int)per_r, dx_add);
  } else {
    hipLaunchKernelGGL(layernorm_bwd_f32, dim3((unsigned)nb), dim3(256), 0, st, x, w, dy, mean, rstd, dx, part, (long long)rows, d, per,
                       dx_add);
  }
  // part is [nb][2][d]: summed over blocks in index order, straight into dw and db
  if (db == dw + d) {   // one [2 d] vector: one launch
    launch_sum_slabs(part, dw, 2 * d, nbl, 2 * d, 0, st, accumulate);
  } else {
    launch_sum_slabs(part, dw, d, nbl, 2 * d, 0, st, accumulate);
    launch_sum_slabs(part, db, d, nbl, 2 * d, d, st, accumulate);
  }
  return check_launch("pv_layernorm_bwd_f32");
}

int pv_layernorm_bwd_params_from_proj_workspace_bytes(int64_t rows, int32_t d, size_t* bytes) {
  PV_REQUIRE(bytes && rows > 0 && d > 0, PV_EINVAL, "pv_layernorm_bwd_params_from_proj_workspace_bytes: bad arguments");
  *bytes = (size_t)lpb_blocks(rows) * 2 * d * sizeof(float);
  return PV_OK;
}

int pv_layernorm_bwd_params_from_proj_bf16(const uint16_t* dkv16, const float* w_kv, const float* x, const float* mean,
                                           const float* rstd, float* dw, float* db, int64_t rows, int32_t d, int32_t kdim,
                                           void* ws, size_t ws_bytes, int32_t accumulate, void* stream) {
  PV_REQUIRE(dkv16 && w_kv && x && mean && rstd && dw && db, PV_EINVAL, "pv_layernorm_bwd_params_from_proj_bf16: null pointer");
  PV_REQUIRE(rows > 0 && d > 0 && d <= 64, PV_ESIZE, "pv_layernorm_bwd_params_from_proj_bf16: d=%d must be in 1..64", d);
  PV_REQUIRE(kdim == 64 || kdim == 128, PV_ESIZE, "pv_layernorm_bwd_params_from_proj_bf16: kdim=%d must be 64 or 128", kdim);
  PV_REQUIRE(((uintptr_t)dkv16 & 15) == 0, PV_EINVAL, "pv_layernorm_bwd_params_from_proj_bf16: gradient rows must be 16-byte aligned");
  const int nb = lpb_blocks(rows);
  PV_REQUIRE(ws && ws_bytes >= (size_t)nb * 2 * d * sizeof(float), PV_EINVAL, "pv_layernorm_bwd_params_from_proj_bf16: workspace too small");
  hipStream_t st = as_stream(stream);
  float* part = (float*)ws;
  const long long rbs = (rows + 31) / 32;
  if (kdim == 128)
    hipLaunchKernelGGL(layernorm_bwd_params_from_proj_kernel<8>, dim3((unsigned)nb), dim3(256), 0, st, dkv16, w_kv, x, mean, rstd,
                       part, (long long)rows, d, rbs);
  else
    hipLaunchKernelGGL(layernorm_bwd_params_from_proj_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, dkv16, w_kv, x, mean, rstd,
                       part, (long long)rows, d, rbs);
  if (db == dw + d) {
    launch_sum_slabs(part, dw, 2 * d, nb, 2 * d, 0, st, accumulate);
  } else {
    launch_sum_slabs(part, dw, d, nb, 2 * d, 0, st, accumulate);
    launch_sum_slabs(part, db, d, nb, 2 * d, d, st, accumulate);
  }
  return check_launch("pv_layernorm_bwd_params_from_proj_bf16");
}

int pv_context_bwd_workspace_bytes(int64_t rows, int32_t d, size_t* bytes) {
  PV_REQUIRE(bytes && rows > 0 && d > 0 && d <= 64, PV_EINVAL, "pv_context_bwd_workspace_bytes: bad arguments");
  *bytes = (size_t)cbk_blocks(rows) * (128 + 2) * d * sizeof(float);
  return PV_OK;
}

int pv_context_bwd_bf16(const uint16_t* dkv16, const float* w_kv, const float* x, const float* x2, int32_t d1, int64_t period,
                        const float* mean, const float* rstd, const float* ln_w, const float* ln_b, float* dw_kv, float* dln_w,
                        float* dln_b, int64_t rows, int32_t d, int32_t kdim, void* ws, size_t ws_bytes, int32_t accumulate_kv,
                        int32_t accumulate_ln, void* stream) {
  PV_REQUIRE(!x2 || (d1 > 0 && d1 < d && period >= 32 && period <= 0x7fffffffLL && rows <= 0x7fffffffLL), PV_EINVAL,
             "pv_context_bwd_bf16: two sources need 0 < d1 < d and a period >= 32");
  PV_REQUIRE(dkv16 && w_kv && x && mean && rstd && ln_w && ln_b && dw_kv && dln_w && dln_b, PV_EINVAL,
             "pv_context_bwd_bf16: null pointer");
  PV_REQUIRE(rows > 0 && d > 0 && d <= 64, PV_ESIZE, "pv_context_bwd_bf16: d=%d must be in 1..64", d);
  PV_REQUIRE(kdim == 128, PV_ESIZE, "pv_context_bwd_bf16: kdim=%d must be 128", kdim);
  PV_REQUIRE(((uintptr_t)dkv16 & 15) == 0, PV_EINVAL, "pv_context_bwd_bf16: gradient rows must be 16-byte aligned");
  const int nb = cbk_blocks(rows);
  PV_REQUIRE(ws && ws_bytes >= (size_t)nb * (128 + 2) * d * sizeof(float), PV_EINVAL, "pv_context_bwd_bf16: workspace too small");
  hipStream_t st = as_stream(stream);
  float* part_dw = (float*)ws;
  float* part_ln = part_dw + (size_t)nb * 128 * d;
  hipLaunchKernelGGL(context_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, dkv16, w_kv, x, mean, rstd, ln_w, ln_b, part_dw,
                     part_ln, (long long)rows, d, (long long)((rows + 31) / 32), x2, d1, (int)period);
  launch_sum_slabs(part_dw, dw_kv, 128LL * d, nb, 128LL * d, 0, st, accumulate_kv);
  if (dln_b == dln_w + d) {
    launch_sum_slabs(part_ln, dln_w, 2 * d, nb, 2 * d, 0, st, accumulate_ln);
  } else {
    launch_sum_slabs(part_ln, dln_w, d, nb, 2 * d, 0, st, accumulate_ln);
    launch_sum_slabs(part_ln, dln_b, d, nb, 2 * d, d, st, accumulate_ln);
  }
  return check_launch("pv_context_bwd_bf16");
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
