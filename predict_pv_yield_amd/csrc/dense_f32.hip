// fp32 fully-connected head, forecast losses and Adam.
// replaces: F.linear / torch.cat-fed FC stack (predict_pv_yield/models/conv3d/model.py:92-103,125-152),
//           F.mse_loss, (y_hat - y).abs().mean(), WeightedLosses (base_model.py:98-103),
//           torch.optim.Adam(lr=5e-4) (base_model.py:255-257).
#include "pv_common.h"

namespace pv {

// dst[i1][i0][:] = src[i0][i1][:] for contiguous segments of seg16 x 16 bytes: the chunk-major staging of the K-sharded fc1's two
// all-to-alls ([B][W][K / W] <-> [W][B][K / W], segments of 245 KB).  torch's strided copy moves these 2 x 128 MB at 2.8 TB/s.
__global__ __launch_bounds__(256) void swap01_segments_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long long n0,
                                                               long long n1, long long seg16) {
  const size_t total = (size_t)n0 * n1 * seg16, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t seg = i / seg16, c = i - seg * seg16;      // destination segment (i1, i0)
    const size_t i1 = seg / n0, i0 = seg - i1 * n0;
    dst[i] = src[(i0 * n1 + i1) * seg16 + c];
  }
}

constexpr int M_TILE = 8;
constexpr long long SMALL_K = 4096, SMALL_MN = 1 << 16;   // "small layer": single-launch forward / backward

// ---- forward: split-K partial products, one block = (n, k-chunk, m-tile) -----------------------
__global__ __launch_bounds__(256) void linear_fwd_partial_f32(const float* __restrict__ x,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ partial, int m, int n,
                                                               long long k, long long k_chunk) {
  const int col = blockIdx.x;      // output feature
  const int ks = blockIdx.y;       // k-chunk
  const int m0 = blockIdx.z * M_TILE;
  const long long k0 = (long long)ks * k_chunk;
  const long long k1 = k0 + k_chunk < k ? k0 + k_chunk : k;
  float acc[M_TILE];
#pragma unroll
  for (int i = 0; i < M_TILE; ++i) acc[i] = 0.f;
  const float* wr = w + (size_t)col * k;
  for (long long kk = k0 + threadIdx.x; kk < k1; kk += blockDim.x) {
    float wv = wr[kk];
#pragma unroll
    for (int i = 0; i < M_TILE; ++i) {
      if (m0 + i < m) acc[i] = fmaf(x[(size_t)(m0 + i) * k + kk], wv, acc[i]);
    }
  }
  __shared__ float red[4][M_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < M_TILE; ++i) {
    float v = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < M_TILE && m0 + threadIdx.x < m) {
    int i = threadIdx.x;
    float s = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    partial[((size_t)ks * m + (m0 + i)) * n + col] = s;
  }
}

// The same partial products for 16-byte aligned operands (k % 4 == 0): one block = 8 rows x 8 output features x k-chunk,
// every lane moves float4s (16 loads feed 256 FMAs), so the weight matrix is read once per 8 rows and x once per 8
// features (fc1 of the exact-f32 model, 514 MB of weights: 2.9 ms with the scalar kernel, 0.9 ms at 8 x 4; 16 x 8 needs 298
// registers and is slower again, 1.7 ms).
constexpr int FT_N = 8, FT_M = 8;
__global__ __launch_bounds__(256) void linear_fwd_tile_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ partial, int m, int n, long long k,
                                                            long long k_chunk) {
  const int n0 = blockIdx.x * FT_N, ks = blockIdx.y, m0 = blockIdx.z * FT_M;
  const long long k0 = (long long)ks * k_chunk;
  const long long k1 = k0 + k_chunk < k ? k0 + k_chunk : k;
  float acc[FT_M][FT_N];
#pragma unroll
  for (int i = 0; i < FT_M; ++i)
#pragma unroll
    for (int j = 0; j < FT_N; ++j) acc[i][j] = 0.f;
  for (long long kk = k0 + 4 * threadIdx.x; kk < k1; kk += 4 * blockDim.x) {
    f32x4 wv[FT_N], xv[FT_M];
#pragma unroll
    for (int j = 0; j < FT_N; ++j)
      wv[j] = n0 + j < n ? *reinterpret_cast<const f32x4*>(w + (size_t)(n0 + j) * k + kk) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FT_M; ++i)
      xv[i] = m0 + i < m ? *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + i) * k + kk) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < FT_M; ++i)
#pragma unroll
        for (int j = 0; j < FT_N; ++j) acc[i][j] = fmaf(xv[i][e], wv[j][e], acc[i][j]);
  }
  __shared__ float red[4][FT_M * FT_N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < FT_M; ++i)
#pragma unroll
    for (int j = 0; j < FT_N; ++j) {
      float v = acc[i][j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][i * FT_N + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < FT_M * FT_N) {
    const int i = threadIdx.x / FT_N, j = threadIdx.x % FT_N;
    if (m0 + i < m && n0 + j < n)
      partial[((size_t)ks * m + (m0 + i)) * n + n0 + j] =
          ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
}

// out = dy where y > 0, else 0: the backward of F.relu (model.py:117-120) as its own pass, for consumers whose kernel cannot
// gate while it stages (the f32 matrix-core dgrad reads its operand global -> LDS directly)
// maxbits (may be NULL): receives, by atomicMax, the bits of the largest |out| -- the scale the two-term half-float split of
// the gated gradient needs (pv_pack_split2_..., have_max): this pass streams the values anyway, a pass of its own costs 62 us
__global__ __launch_bounds__(256) void relu_gate_f32_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             float* __restrict__ out, size_t n4, uint32_t* __restrict__ maxbits) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 d = reinterpret_cast<const f32x4*>(dy)[i], g = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = g[e] > 0.f ? d[e] : 0.f;
      m = max(m, __builtin_bit_cast(uint32_t, o[e]) & 0x7fffffffu);
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
  if (maxbits) {      // one atomic per workgroup (same-address atomics serialise at ~15 ns apiece)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    __shared__ uint32_t wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
      if (m) atomicMax(maxbits, m);
    }
  }
}

// y[m,n] = relu?(bias[n] + sum_ks partial[ks][m][n]) (fixed order: deterministic)
__global__ __launch_bounds__(256) void linear_reduce_f32(const float* __restrict__ partial,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int m, int n, int k_splits, int relu) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * n) return;
  int col = i % n;
  float s = 0.f;
  for (int ks = 0; ks < k_splits; ++ks) s += partial[(size_t)ks * m * n + i];
  if (bias) s += bias[col];
  if (relu) s = s > 0.f ? s : 0.f;
  y[i] = s;
}

// ---- backward ------------------------------------------------------------------------------------
// dx[mi, kk] = sum_n g[mi, n] * w[n, kk]; thread = one kk, M_TILE rows; g tile broadcast from LDS
__device__ __forceinline__ void linear_bwd_dx_body(float* g /* LDS [n][M_TILE] */, const float* __restrict__ w,
                                                   const float* __restrict__ dy, const float* __restrict__ ymask,
                                                   float* __restrict__ dx, int m, int n, long long k, int bx, int by) {
  const int m0 = by * M_TILE;
  for (int i = threadIdx.x; i < n * M_TILE; i += blockDim.x) {
    int col = i / M_TILE, r = i % M_TILE;
    float v = 0.f;
    if (m0 + r < m) {
      size_t off = (size_t)(m0 + r) * n + col;
      v = dy[off];
      if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    }
    g[i] = v;
  }
  __syncthreads();
  long long kk = (long long)bx * blockDim.x + threadIdx.x;
  if (kk >= k) return;
  float acc[M_TILE];
#pragma unroll
  for (int i = 0; i < M_TILE; ++i) acc[i] = 0.f;
  int col = 0;
  // thirty-two weight rows in flight where there are as many (same order of additions): with eight, the 128-feature layers of
  // the head were a chain of sixteen dependent memory latencies (10 us for fc2's 64 KB)
  for (; col + 32 <= n; col += 32) {
    float wv[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) wv[j] = w[(size_t)(col + j) * k + kk];
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int i = 0; i < M_TILE; ++i) acc[i] = fmaf(g[(col + j) * M_TILE + i], wv[j], acc[i]);
  }
  for (; col + 8 <= n; col += 8) {       // eight weight rows in flight (same order of additions)
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[(size_t)(col + j) * k + kk];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < M_TILE; ++i) acc[i] = fmaf(g[(col + j) * M_TILE + i], wv[j], acc[i]);
  }
  for (; col < n; ++col) {
    float wv = w[(size_t)col * k + kk];
#pragma unroll
    for (int i = 0; i < M_TILE; ++i) acc[i] = fmaf(g[col * M_TILE + i], wv, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < M_TILE; ++i)
    if (m0 + i < m) dx[(size_t)(m0 + i) * k + kk] = acc[i];
}

__global__ __launch_bounds__(256) void linear_bwd_dx_f32(const float* __restrict__ w, const float* __restrict__ dy,
                                                          const float* __restrict__ ymask, float* __restrict__ dx,
                                                          int m, int n, long long k) {
  extern __shared__ float g[];  // [n][M_TILE]
  linear_bwd_dx_body(g, w, dy, ymask, dx, m, n, k, blockIdx.x, blockIdx.y);
}

// dw[n0+j, kk] = sum_mi g[mi, n0+j] * x[mi, kk]; thread = one kk, N_TILE output features
constexpr int N_TILE = 8;
__device__ __forceinline__ void linear_bwd_dw_body(float* g /* LDS [m][N_TILE] */, const float* __restrict__ x,
                                                   const float* __restrict__ dy, const float* __restrict__ ymask,
                                                   float* __restrict__ dw, int m, int n, long long k, int bx, int by) {
  const int n0 = by * N_TILE;
  for (int i = threadIdx.x; i < m * N_TILE; i += blockDim.x) {
    int r = i / N_TILE, j = i % N_TILE;
    float v = 0.f;
    if (n0 + j < n) {
      size_t off = (size_t)r * n + n0 + j;
      v = dy[off];
      if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    }
    g[i] = v;
  }
  __syncthreads();
  long long kk = (long long)bx * blockDim.x + threadIdx.x;
  if (kk >= k) return;
  float acc[N_TILE];
#pragma unroll
  for (int j = 0; j < N_TILE; ++j) acc[j] = 0.f;
  int r = 0;
  for (; r + 32 <= m; r += 32) {         // thirty-two rows of x in flight where there are as many (same order of additions)
    float xv[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) xv[q] = x[(size_t)(r + q) * k + kk];
#pragma unroll
    for (int q = 0; q < 32; ++q)
#pragma unroll
      for (int j = 0; j < N_TILE; ++j) acc[j] = fmaf(g[(r + q) * N_TILE + j], xv[q], acc[j]);
  }
  for (; r + 8 <= m; r += 8) {           // eight rows of x in flight (same order of additions)
    float xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) xv[q] = x[(size_t)(r + q) * k + kk];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int j = 0; j < N_TILE; ++j) acc[j] = fmaf(g[(r + q) * N_TILE + j], xv[q], acc[j]);
  }
  for (; r < m; ++r) {
    float xv = x[(size_t)r * k + kk];
#pragma unroll
    for (int j = 0; j < N_TILE; ++j) acc[j] = fmaf(g[r * N_TILE + j], xv, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < N_TILE; ++j)
    if (n0 + j < n) dw[(size_t)(n0 + j) * k + kk] = acc[j];
}

__global__ __launch_bounds__(256) void linear_bwd_dw_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ ymask, float* __restrict__ dw,
                                                          int m, int n, long long k) {
  extern __shared__ float g[];  // [m][N_TILE]
  linear_bwd_dw_body(g, x, dy, ymask, dw, m, n, k, blockIdx.x, blockIdx.y);
}

__device__ __forceinline__ void linear_bwd_db_body(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                   float* __restrict__ db, int m, int n, int bx);

__global__ __launch_bounds__(256) void linear_bwd_db_f32(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                          float* __restrict__ db, int m, int n) {
  linear_bwd_db_body(dy, ymask, db, m, n, blockIdx.x);
}

__device__ __forceinline__ void linear_bwd_db_body(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                   float* __restrict__ db, int m, int n, int bx) {
  int col = bx * blockDim.x + threadIdx.x;
  if (col >= n) return;
  float s = 0.f;
  for (int r = 0; r < m; ++r) {
    size_t off = (size_t)r * n + col;
    float v = dy[off];
    if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    s += v;
  }
  db[col] = s;
}

// The small layers of the head (fc2..fc4, decoder_fc*: k <= a few thousand): the whole backward in ONE launch -- the
// dx, dw and db tiles are the same device functions, selected by block index -- and the forward without the split-K
// round trip: one thread per output, k-loop in registers (fixed order).
__global__ __launch_bounds__(256) void linear_bwd_small_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dy, const float* __restrict__ ymask,
                                                             float* __restrict__ dx, float* __restrict__ dw,
                                                             float* __restrict__ db, int m, int n, long long k, int kb,
                                                             int n_dx, int n_dw) {
  extern __shared__ float g[];  // max(n * M_TILE, m * N_TILE) floats
  int blk = blockIdx.x;
  if (blk < n_dx) {
    linear_bwd_dx_body(g, w, dy, ymask, dx, m, n, k, blk % kb, blk / kb);
  } else if (blk < n_dx + n_dw) {
    blk -= n_dx;
    linear_bwd_dw_body(g, x, dy, ymask, dw, m, n, k, blk % kb, blk / kb);
  } else {
    linear_bwd_db_body(dy, ymask, db, m, n, blk - n_dx - n_dw);
  }
}

__global__ __launch_bounds__(256) void linear_fwd_small_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int m,
                                                             int n, int k, int relu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * n) return;
  const int r = i / n, col = i - r * n;
  const float* xr = x + (size_t)r * k;
  const float* wr = w + (size_t)col * k;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int kk = 0;
  // 32 values of x and of w per trip, loaded before the first multiply-add (one value at a time this launch was a chain
  // of k / 4 dependent cache latencies: 7 us for 128 x 128); the additions keep their order
  if ((((uintptr_t)xr | (uintptr_t)wr) & 15) == 0) {
    for (; kk + 64 <= k; kk += 64) {      // 64 values of x and of w per trip where there are as many (two trips for k = 128)
      f32x4 xv[16], wv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xr + kk + 4 * j), wv[j] = *reinterpret_cast<const f32x4*>(wr + kk + 4 * j);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        s0 = fmaf(xv[j][0], wv[j][0], s0);
        s1 = fmaf(xv[j][1], wv[j][1], s1);
        s2 = fmaf(xv[j][2], wv[j][2], s2);
        s3 = fmaf(xv[j][3], wv[j][3], s3);
      }
    }
    for (; kk + 32 <= k; kk += 32) {
      f32x4 xv[8], wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xr + kk + 4 * j), wv[j] = *reinterpret_cast<const f32x4*>(wr + kk + 4 * j);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s0 = fmaf(xv[j][0], wv[j][0], s0);
        s1 = fmaf(xv[j][1], wv[j][1], s1);
        s2 = fmaf(xv[j][2], wv[j][2], s2);
        s3 = fmaf(xv[j][3], wv[j][3], s3);
      }
    }
  }
  for (; kk + 4 <= k; kk += 4) {
    s0 = fmaf(xr[kk], wr[kk], s0);
    s1 = fmaf(xr[kk + 1], wr[kk + 1], s1);
    s2 = fmaf(xr[kk + 2], wr[kk + 2], s2);
    s3 = fmaf(xr[kk + 3], wr[kk + 3], s3);
  }
  for (; kk < k; ++kk) s0 = fmaf(xr[kk], wr[kk], s0);
  float s = (s0 + s1) + (s2 + s3);
  if (bias) s += bias[col];
  if (relu) s = s > 0.f ? s : 0.f;
  y[i] = s;
}

// ---- forecast losses (single block; [B, forecast_len] is tiny) ----------------------------------
struct LossWeights { float w[64]; };

__global__ __launch_bounds__(256) void forecast_losses_f32(const float* __restrict__ y_hat, const float* __restrict__ y,
                                                            long long y_row_stride, long long y_col_stride, int m,
                                                            int n, float grad_scale, LossWeights lw,
                                                            float* __restrict__ out4, float* __restrict__ grad,
                                                            float* __restrict__ per_horizon) {
  float s_se = 0.f, s_ae = 0.f, s_wse = 0.f, s_wae = 0.f;
  const int total = m * n;
  const float inv = 1.0f / (float)total;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    int r = i / n, c = i - r * n;
    float d = y_hat[i] - y[(long long)r * y_row_stride + (long long)c * y_col_stride];
    float a = fabsf(d);
    s_se += d * d;
    s_ae += a;
    s_wse += lw.w[c] * (d * d);
    s_wae += lw.w[c] * a;
    if (grad) grad[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv * grad_scale;
  }
  __shared__ float red[4][4];
  float v[4] = {s_se, s_ae, s_wse, s_wae};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float t = v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    int q = threadIdx.x;
    out4[q] = (((red[0][q] + red[1][q]) + red[2][q]) + red[3][q]) * inv;
  }
  // mse_each_forecast_horizon / mae_each_forecast_horizon (base_model.py:123-124): mean over the batch axis for every
  // forecast step; one thread per step walks the rows in index order (deterministic; m * n is a few hundred values)
  if (per_horizon && (int)threadIdx.x < n) {
    const int c = threadIdx.x;
    float se = 0.f, ae = 0.f;
    for (int r = 0; r < m; ++r) {
      float d = y_hat[r * n + c] - y[(long long)r * y_row_stride + (long long)c * y_col_stride];
      se += d * d;
      ae += fabsf(d);
    }
    per_horizon[c] = se / (float)m;
    per_horizon[n + c] = ae / (float)m;
  }
}

// ---- Adam: torch._single_tensor_adam order of operations, f32 ------------------------------------
template <typename GT>  // gradient type: float, or uint16_t holding bf16 (data-parallel wire format)
__global__ __launch_bounds__(256) void adam_step_f32(float* __restrict__ p, const GT* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      uint16_t* __restrict__ shadow, size_t n, float one_minus_b1,
                                                      float beta2, float one_minus_b2, float bc2_sqrt, float eps,
                                                      float neg_step_size, float grad_scale) {
  size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f32x4 pv4 = *reinterpret_cast<const f32x4*>(p + i);
      f32x4 gv4;
      if constexpr (sizeof(GT) == 4) {
        gv4 = *reinterpret_cast<const f32x4*>(g + i);
      } else {
        const u32x2 raw = *reinterpret_cast<const u32x2*>(g + i);
        gv4[0] = __builtin_bit_cast(float, raw[0] << 16); gv4[1] = __builtin_bit_cast(float, raw[0] & 0xffff0000u);
        gv4[2] = __builtin_bit_cast(float, raw[1] << 16); gv4[3] = __builtin_bit_cast(float, raw[1] & 0xffff0000u);
      }
      f32x4 mv4 = *reinterpret_cast<const f32x4*>(m + i);
      f32x4 vv4 = *reinterpret_cast<const f32x4*>(v + i);
      uint16_t sh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gr = gv4[j] * grad_scale;
        float mm = mv4[j] + one_minus_b1 * (gr - mv4[j]);        // exp_avg.lerp_(grad, 1 - beta1)
        float vv = vv4[j] * beta2 + (one_minus_b2 * gr) * gr;     // mul_(beta2).addcmul_(g, g, 1 - beta2)
        float denom = sqrtf(vv) / bc2_sqrt + eps;
        float pp = pv4[j] + neg_step_size * (mm / denom);        // addcdiv_(exp_avg, denom, -step_size)
        mv4[j] = mm; vv4[j] = vv; pv4[j] = pp;
        sh[j] = f32_to_bf16_bits(pp);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv4;
      *reinterpret_cast<f32x4*>(m + i) = mv4;
      *reinterpret_cast<f32x4*>(v + i) = vv4;
      if (shadow) {
        u32x2 o = {(uint32_t)sh[0] | ((uint32_t)sh[1] << 16), (uint32_t)sh[2] | ((uint32_t)sh[3] << 16)};
        *reinterpret_cast<u32x2*>(shadow + i) = o;
      }
    } else {
      for (size_t q = i; q < n; ++q) {
        float gq;
        if constexpr (sizeof(GT) == 4) gq = (float)g[q]; else gq = bf16_bits_to_f32((uint16_t)g[q]);
        float gr = gq * grad_scale;
        float mm = m[q] + one_minus_b1 * (gr - m[q]);
        float vv = v[q] * beta2 + (one_minus_b2 * gr) * gr;
        float denom = sqrtf(vv) / bc2_sqrt + eps;
        float pp = p[q] + neg_step_size * (mm / denom);
        m[q] = mm; v[q] = vv; p[q] = pp;
        if (shadow) shadow[q] = f32_to_bf16_bits(pp);
      }
    }
  }
}

__global__ __launch_bounds__(256) void cast_f32_to_bf16_kernel(const float* __restrict__ src,
                                                                uint16_t* __restrict__ dst, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
      u32x2 o = {pack_bf16_pair(v[0], v[1]),
                 pack_bf16_pair(v[2], v[3])};
      *reinterpret_cast<u32x2*>(dst + i) = o;
    } else {
      for (size_t q = i; q < n; ++q) dst[q] = f32_to_bf16_bits(src[q]);
    }
  }
}

// y[i][j] = act(alpha x[i][j] + bias[j]): the epilogue of a Linear whose product was summed elsewhere (K-sharded fc1: partial
// products of the ranks' column shards), and the 1 / world scaling of an output gradient
__global__ __launch_bounds__(256) void scale_bias_relu_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                               float* __restrict__ y, size_t total, int n, float alpha, int relu) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    float v = __fmul_rn(alpha, x[i]);
    if (bias) v = __fadd_rn(v, bias[i % (size_t)n]);
    y[i] = (relu && !(v > 0.f)) ? (v != v ? v : 0.f) : v;      // (a NaN stays a NaN, as torch's relu keeps it)
  }
}

// ---- embedding (nn.Embedding(940, 16) on pv_system_row_number / gsp_id, model_sat_nwp.py:149-151,251-260) ------------
__global__ __launch_bounds__(256) void embedding_fwd_f32(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                          float* __restrict__ out, int n_ids, int dim, int n_rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ids * dim) return;
  int r = i / dim, c = i - r * dim;
  long long id = ids[r];
  out[i] = (id >= 0 && id < n_rows) ? table[(size_t)id * dim + c] : 0.f;
}
// dtable[v][c] = sum over the batch rows whose id == v (fixed order: deterministic, no atomics; the batch is tiny)
__global__ __launch_bounds__(256) void embedding_bwd_f32(const float* __restrict__ dout, const int64_t* __restrict__ ids,
                                                          float* __restrict__ dtable, int n_ids, int dim, int n_rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * dim) return;
  int v = i / dim, c = i - v * dim;
  float s = 0.f;
  for (int r = 0; r < n_ids; ++r)
    if (ids[r] == v) s += dout[(size_t)r * dim + c];
  dtable[i] = s;
}

// multi-tensor Adam: block -> (tensor, 1024-element chunk) through a table passed by value in the kernel arguments
struct AdamTable {
  pv_adam_tensor t[PV_ADAM_MAX_TENSORS];
  int blk0[PV_ADAM_MAX_TENSORS + 1];  // first block of each tensor
  int n_tensors;
};

// one thread: step <- step + 1 and the six scalars every Adam kernel of that step reads (the host computes the same numbers
// for the by-value form; here they live in device memory so that a captured HIP graph can be replayed step after step)
__global__ void adam_scalars_advance_kernel(float* __restrict__ scalars, int* __restrict__ step, double lr, double beta1,
                                            double beta2, double eps) {
  const int s = *step + 1;
  *step = s;
  const double bc1 = 1.0 - pow(beta1, (double)s), bc2 = 1.0 - pow(beta2, (double)s);
  scalars[0] = (float)(1.0 - beta1);
  scalars[1] = (float)beta2;
  scalars[2] = (float)(1.0 - beta2);
  scalars[3] = (float)sqrt(bc2);
  scalars[4] = (float)eps;
  scalars[5] = (float)(-(lr / bc1));
}

__global__ __launch_bounds__(256) void adam_step_multi_f32(AdamTable tab, float one_minus_b1, float beta2, float one_minus_b2,
                                                            float bc2_sqrt, float eps, float neg_step_size, float grad_scale,
                                                            const float* __restrict__ ad_dev) {
  if (ad_dev) {   // scalars of this step from device memory (pv_adam_scalars_advance): the graph-replayable form
    one_minus_b1 = ad_dev[0], beta2 = ad_dev[1], one_minus_b2 = ad_dev[2], bc2_sqrt = ad_dev[3], eps = ad_dev[4];
    neg_step_size = ad_dev[5];
  }
  int ti = 0;
  while (ti + 1 < tab.n_tensors && (int)blockIdx.x >= tab.blk0[ti + 1]) ++ti;
  const pv_adam_tensor T = tab.t[ti];
  const size_t i0 = ((size_t)(blockIdx.x - tab.blk0[ti]) * 256 + threadIdx.x) * 4;
  if (i0 >= T.n) return;
  const int cnt = T.n - i0 < 4 ? (int)(T.n - i0) : 4;
  float pv[4], gv[4], mv[4], vv[4];
  if (cnt == 4) {
    *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(T.param + i0);
    *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(T.grad + i0);
    *reinterpret_cast<f32x4*>(mv) = *reinterpret_cast<const f32x4*>(T.exp_avg + i0);
    *reinterpret_cast<f32x4*>(vv) = *reinterpret_cast<const f32x4*>(T.exp_avg_sq + i0);
  } else {
    for (int j = 0; j < 4; ++j) {
      const bool ok = j < cnt;
      pv[j] = ok ? T.param[i0 + j] : 0.f, gv[j] = ok ? T.grad[i0 + j] : 0.f;
      mv[j] = ok ? T.exp_avg[i0 + j] : 0.f, vv[j] = ok ? T.exp_avg_sq[i0 + j] : 1.f;
    }
  }
  uint16_t sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // same operation order as adam_step_f32
    const float gr = gv[j] * grad_scale;
    const float mm = mv[j] + one_minus_b1 * (gr - mv[j]);
    const float v2 = vv[j] * beta2 + (one_minus_b2 * gr) * gr;
    const float denom = sqrtf(v2) / bc2_sqrt + eps;
    const float pp = pv[j] + neg_step_size * (mm / denom);
    mv[j] = mm, vv[j] = v2, pv[j] = pp;
    sh[j] = f32_to_bf16_bits(pp);
  }
  if (cnt == 4) {
    *reinterpret_cast<f32x4*>(T.param + i0) = *reinterpret_cast<const f32x4*>(pv);
    *reinterpret_cast<f32x4*>(T.exp_avg + i0) = *reinterpret_cast<const f32x4*>(mv);
    *reinterpret_cast<f32x4*>(T.exp_avg_sq + i0) = *reinterpret_cast<const f32x4*>(vv);
    if (T.bf16_shadow) {
      const u32x2 o = {(uint32_t)sh[0] | ((uint32_t)sh[1] << 16), (uint32_t)sh[2] | ((uint32_t)sh[3] << 16)};
      *reinterpret_cast<u32x2*>(T.bf16_shadow + i0) = o;
    }
  } else {
    for (int j = 0; j < cnt; ++j) {
      T.param[i0 + j] = pv[j], T.exp_avg[i0 + j] = mv[j], T.exp_avg_sq[i0 + j] = vv[j];
      if (T.bf16_shadow) T.bf16_shadow[i0 + j] = sh[j];
    }
  }
}

// k from which the forward goes to pv_gemm_f32 (gemm_f32.hip: bf16 x 3 on the matrix cores, f32-accurate to ~1e-6): fc1 of
// the f32 model, 32 x 1 003 520 -> 128, 920 us on the register-tile kernel below against 305 us
constexpr long long GEMM_K = 1 << 16;

static long long fwd_k_chunk(long long k, int* k_splits) {
  // ~8k elements of K per block (2k for the matrix-core path: two output tiles per chunk), at most 512 splits
  long long chunk = k >= GEMM_K ? 2048 : 8192;
  long long splits = (k + chunk - 1) / chunk;
  if (splits > 512) { splits = 512; chunk = (k + splits - 1) / splits; }
  if (splits < 1) splits = 1;
  splits = (k + chunk - 1) / chunk;
  *k_splits = (int)splits;
  return chunk;
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_linear_workspace_bytes(int32_t m, int32_t n, int64_t k, size_t* bytes) {
  PV_REQUIRE(bytes && m > 0 && n > 0 && k > 0, PV_EINVAL, "pv_linear_workspace_bytes: bad arguments");
  int ks;
  fwd_k_chunk(k, &ks);
  *bytes = (size_t)ks * m * n * sizeof(float);
  return PV_OK;
}

int pv_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, int32_t m, int32_t n, int64_t k,
                      int relu, void* workspace, size_t workspace_bytes, void* stream) {
  PV_REQUIRE(x && w && y && workspace, PV_EINVAL, "pv_linear_fwd_f32: null pointer");
  PV_REQUIRE(m > 0 && n > 0 && k > 0, PV_EINVAL, "pv_linear_fwd_f32: bad sizes");
  int ks;
  long long chunk = fwd_k_chunk(k, &ks);
  PV_REQUIRE(workspace_bytes >= (size_t)ks * m * n * sizeof(float), PV_ESIZE, "pv_linear_fwd_f32: workspace too small");
  PV_REQUIRE(n <= 65535 && ks <= 65535, PV_ESIZE, "pv_linear_fwd_f32: n too large");
  hipStream_t st = as_stream(stream);
  if (k <= SMALL_K && (long long)m * n <= SMALL_MN) {
    hipLaunchKernelGGL(linear_fwd_small_f32, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, st, x, w, bias, y, m, n, (int)k,
                       relu ? 1 : 0);
    return check_launch("pv_linear_fwd_f32");
  }
  if (k >= GEMM_K && k < (1ll << 31)) {
    // partial[s] = x[:, chunk s] W[:, chunk s]^T : A = x (row stride k), B = W^T as a strided view (contraction index contiguous)
    pv_gemm_desc d = {m, n, (int32_t)k, k, 1, 1, k, n, 1, 1, 0, 0, 0, 0, 0, 0, ks, (int64_t)m * n};
    const int rc = pv_gemm_f32(x, w, nullptr, (float*)workspace, &d, 0, stream);
    if (rc != PV_OK) return rc;
  } else if (k % 4 == 0 && chunk % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0) {
    dim3 grid((unsigned)((n + FT_N - 1) / FT_N), (unsigned)ks, (unsigned)((m + FT_M - 1) / FT_M));
    hipLaunchKernelGGL(linear_fwd_tile_f32, grid, dim3(256), 0, st, x, w, (float*)workspace, m, n, (long long)k, chunk);
  } else {
    dim3 grid((unsigned)n, (unsigned)ks, (unsigned)((m + M_TILE - 1) / M_TILE));
    hipLaunchKernelGGL(linear_fwd_partial_f32, grid, dim3(256), 0, st, x, w, (float*)workspace, m, n, (long long)k, chunk);
  }
  hipLaunchKernelGGL(linear_reduce_f32, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, st,
                     (const float*)workspace, bias, y, m, n, ks, relu ? 1 : 0);
  return check_launch("pv_linear_fwd_f32");
}

int pv_linear_bwd_f32(const float* x, const float* w, const float* dy, const float* y_relu_mask, float* dx, float* dw,
                      float* db, int32_t m, int32_t n, int64_t k, void* stream) {
  PV_REQUIRE(dy, PV_EINVAL, "pv_linear_bwd_f32: null dy");
  PV_REQUIRE(m > 0 && n > 0 && k > 0, PV_EINVAL, "pv_linear_bwd_f32: bad sizes");
  hipStream_t st = as_stream(stream);
  unsigned kb = (unsigned)((k + 255) / 256);
  if (dx && dw && db && k <= SMALL_K && (long long)m * n <= SMALL_MN) {
    PV_REQUIRE(w && x, PV_EINVAL, "pv_linear_bwd_f32: dx needs w, dw needs x");
    const size_t lds = (size_t)std::max(n * M_TILE, m * N_TILE) * sizeof(float);
    PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_bwd_f32: m=%d / n=%d too large", m, n);
    const int n_dx = (int)kb * ((m + M_TILE - 1) / M_TILE), n_dw = (int)kb * ((n + N_TILE - 1) / N_TILE);
    const int n_db = (n + 255) / 256;
    hipLaunchKernelGGL(linear_bwd_small_f32, dim3((unsigned)(n_dx + n_dw + n_db)), dim3(256), lds, st, x, w, dy, y_relu_mask,
                       dx, dw, db, m, n, (long long)k, (int)kb, n_dx, n_dw);
    return check_launch("pv_linear_bwd_f32");
  }
  if (dx) {
    PV_REQUIRE(w, PV_EINVAL, "pv_linear_bwd_f32: dx needs w");
    size_t lds = (size_t)n * M_TILE * sizeof(float);
    PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_bwd_f32: n=%d too large", n);
    hipLaunchKernelGGL(linear_bwd_dx_f32, dim3(kb, (unsigned)((m + M_TILE - 1) / M_TILE)), dim3(256), lds, st, w, dy,
                       y_relu_mask, dx, m, n, (long long)k);
  }
  if (dw) {
    PV_REQUIRE(x, PV_EINVAL, "pv_linear_bwd_f32: dw needs x");
    size_t lds = (size_t)m * N_TILE * sizeof(float);
    PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_bwd_f32: m=%d too large", m);
    hipLaunchKernelGGL(linear_bwd_dw_f32, dim3(kb, (unsigned)((n + N_TILE - 1) / N_TILE)), dim3(256), lds, st, x, dy,
                       y_relu_mask, dw, m, n, (long long)k);
  }
  if (db) {
    hipLaunchKernelGGL(linear_bwd_db_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, y_relu_mask, db, m, n);
  }
  return check_launch("pv_linear_bwd_f32");
}

int pv_relu_gate_f32(const float* dy, const float* y, float* out, size_t n, void* stream) {
  PV_REQUIRE(dy && y && out, PV_EINVAL, "pv_relu_gate_f32: null pointer");
  PV_REQUIRE(n % 4 == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)out % 16) == 0, PV_EINVAL,
             "pv_relu_gate_f32: 16-byte aligned buffers of a multiple of 4 elements");
  if (n == 0) return PV_OK;
  hipLaunchKernelGGL(relu_gate_f32_kernel, dim3(stream_grid(n / 4, 256)), dim3(256), 0, as_stream(stream), dy, y, out, n / 4,
                     (uint32_t*)nullptr);
  return check_launch("pv_relu_gate_f32");
}

int pv_relu_gate_max_f32(const float* dy, const float* y, float* out, size_t n, float* state, void* stream) {
  PV_REQUIRE(dy && y && out && state, PV_EINVAL, "pv_relu_gate_max_f32: null pointer");
  PV_REQUIRE(n % 4 == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                 ((uintptr_t)state % 4) == 0, PV_EINVAL,
             "pv_relu_gate_max_f32: 16-byte aligned buffers of a multiple of 4 elements");
  hipStream_t st = as_stream(stream);
  PV_REQUIRE(hipMemsetAsync(state, 0, sizeof(uint32_t), st) == hipSuccess, PV_ELAUNCH, "pv_relu_gate_max_f32: memset failed");
  if (n == 0) return PV_OK;
  hipLaunchKernelGGL(relu_gate_f32_kernel, dim3(std::min<unsigned>(stream_grid(n / 4, 256), 4 * kNumCU)), dim3(256), 0, st, dy, y, out,
                     n / 4, reinterpret_cast<uint32_t*>(state));
  return check_launch("pv_relu_gate_max_f32");
}

int pv_forecast_losses_f32(const float* y_hat, const float* y, int64_t y_row_stride, int64_t y_col_stride, int32_t m,
                           int32_t n, float grad_scale, float* out4, float* grad, float* per_horizon, void* stream) {
  PV_REQUIRE(y_hat && y && out4, PV_EINVAL, "pv_forecast_losses_f32: null pointer");
  PV_REQUIRE(m > 0 && n > 0 && n <= 64, PV_ESIZE, "pv_forecast_losses_f32: forecast length %d outside 1..64", n);
  // WeightedLosses(forecast_length=n): w_i = exp(-ln2 * i), normalised to mean 1 (f32 like torch.FloatTensor)
  LossWeights lw;
  float tmp[64];
  float sum = 0.f;
  for (int i = 0; i < n; ++i) { tmp[i] = (float)exp(-0.6931471805599453 * i); sum += tmp[i]; }
  for (int i = 0; i < 64; ++i) lw.w[i] = i < n ? tmp[i] / sum * (float)n : 0.f;
  hipLaunchKernelGGL(forecast_losses_f32, dim3(1), dim3(256), 0, as_stream(stream), y_hat, y, (long long)y_row_stride,
                     (long long)y_col_stride, m, n, grad_scale, lw, out4, grad, per_horizon);
  return check_launch("pv_forecast_losses_f32");
}

int pv_swap01_segments(const void* src, void* dst, int64_t n0, int64_t n1, int64_t seg_bytes, void* stream) {
  PV_REQUIRE(src && dst && src != dst, PV_EINVAL, "pv_swap01_segments: null pointer / in place");
  PV_REQUIRE(n0 > 0 && n1 > 0 && seg_bytes > 0 && seg_bytes % 16 == 0, PV_ESIZE,
             "pv_swap01_segments: n0, n1 > 0 and segments of a multiple of 16 bytes, got %lld x %lld x %lld", (long long)n0,
             (long long)n1, (long long)seg_bytes);
  PV_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, PV_EINVAL, "pv_swap01_segments: buffers must be 16-byte aligned");
  const size_t total = (size_t)n0 * n1 * (seg_bytes / 16);
  hipLaunchKernelGGL(swap01_segments_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream),
                     static_cast<const u32x4*>(src), static_cast<u32x4*>(dst), (long long)n0, (long long)n1, (long long)(seg_bytes / 16));
  return check_launch("pv_swap01_segments");
}

int pv_cast_f32_to_bf16(const float* src, uint16_t* dst, size_t n, void* stream) {
  PV_REQUIRE(src && dst, PV_EINVAL, "pv_cast_f32_to_bf16: null pointer");
  PV_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), PV_EINVAL, "pv_cast_f32_to_bf16: unaligned");
  if (n == 0) return PV_OK;
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), src,
                     dst, n);
  return check_launch("pv_cast_f32_to_bf16");
}

int pv_scale_bias_relu_f32(const float* x, const float* bias, float* y, int32_t m, int32_t n, float alpha, int32_t relu,
                           void* stream) {
  PV_REQUIRE(x && y, PV_EINVAL, "pv_scale_bias_relu_f32: null pointer");
  PV_REQUIRE(m >= 0 && n > 0, PV_ESIZE, "pv_scale_bias_relu_f32: bad sizes");
  const size_t total = (size_t)m * (size_t)n;
  if (total == 0) return PV_OK;
  hipLaunchKernelGGL(scale_bias_relu_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x, bias, y, total, n,
                     alpha, relu);
  return check_launch("pv_scale_bias_relu_f32");
}

int pv_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint16_t* bf16_shadow,
                     size_t n, double lr, double beta1, double beta2, double eps, int32_t step, float grad_scale,
                     void* stream) {
  PV_REQUIRE(param && grad && exp_avg && exp_avg_sq, PV_EINVAL, "pv_adam_step_f32: null pointer");
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_adam_step_f32: step must be >= 1");
  PV_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                 ((uintptr_t)exp_avg_sq % 16 == 0) && ((uintptr_t)bf16_shadow % 8 == 0),
             PV_EINVAL, "pv_adam_step_f32: buffers must be 16-byte aligned");
  if (n == 0) return PV_OK;
  // python-float (double) scalars exactly as torch computes them, then narrowed to f32 at the kernel boundary
  double bc1 = 1.0 - pow(beta1, (double)step);
  double bc2 = 1.0 - pow(beta2, (double)step);
  double step_size = lr / bc1;
  double bc2_sqrt = sqrt(bc2);
  hipLaunchKernelGGL(adam_step_f32<float>, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), param, grad,
                     exp_avg, exp_avg_sq, bf16_shadow, n, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)bc2_sqrt, (float)eps, (float)(-step_size), grad_scale);
  return check_launch("pv_adam_step_f32");
}

static int adam_multi_table(const pv_adam_tensor* tensors, int32_t n_tensors, AdamTable* tab, int* blocks_out, const char* who) {
  PV_REQUIRE(tensors && n_tensors > 0 && n_tensors <= PV_ADAM_MAX_TENSORS, PV_EINVAL, "%s: 1..%d tensors per call", who,
             PV_ADAM_MAX_TENSORS);
  int blocks = 0;
  tab->n_tensors = 0;
  for (int i = 0; i < n_tensors; ++i) {
    const pv_adam_tensor& t = tensors[i];
    PV_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, PV_EINVAL, "%s: null pointer in tensor %d", who, i);
    PV_REQUIRE(((uintptr_t)t.param % 16 == 0) && ((uintptr_t)t.grad % 16 == 0) && ((uintptr_t)t.exp_avg % 16 == 0) &&
                   ((uintptr_t)t.exp_avg_sq % 16 == 0) && ((uintptr_t)t.bf16_shadow % 8 == 0),
               PV_EINVAL, "%s: buffers of tensor %d must be 16-byte aligned", who, i);
    PV_REQUIRE(t.n < (1ull << 31), PV_ESIZE, "%s: tensor %d too large for the multi-tensor path", who, i);
    if (t.n == 0) continue;
    tab->t[tab->n_tensors] = t;
    tab->blk0[tab->n_tensors] = blocks;
    blocks += (int)((t.n + 1023) / 1024);
    ++tab->n_tensors;
  }
  tab->blk0[tab->n_tensors] = blocks;
  *blocks_out = blocks;
  return PV_OK;
}

int pv_adam_step_multi_f32(const pv_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                           double eps, int32_t step, float grad_scale, void* stream) {
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_adam_step_multi_f32: step must be >= 1");
  AdamTable tab;
  int blocks = 0;
  int rc = adam_multi_table(tensors, n_tensors, &tab, &blocks, "pv_adam_step_multi_f32");
  if (rc) return rc;
  if (tab.n_tensors == 0) return PV_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_step_multi_f32, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tab, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(-(lr / bc1)), grad_scale,
                     (const float*)nullptr);
  return check_launch("pv_adam_step_multi_f32");
}

int pv_adam_scalars_advance(float* scalars_dev, int32_t* step_dev, double lr, double beta1, double beta2, double eps,
                            void* stream) {
  PV_REQUIRE(scalars_dev && step_dev, PV_EINVAL, "pv_adam_scalars_advance: null pointer");
  hipLaunchKernelGGL(adam_scalars_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), scalars_dev, step_dev, lr, beta1, beta2,
                     eps);
  return check_launch("pv_adam_scalars_advance");
}

int pv_adam_step_multi_dev_f32(const pv_adam_tensor* tensors, int32_t n_tensors, const float* scalars_dev, float grad_scale,
                               void* stream) {
  PV_REQUIRE(scalars_dev, PV_EINVAL, "pv_adam_step_multi_dev_f32: null scalars");
  AdamTable tab;
  int blocks = 0;
  int rc = adam_multi_table(tensors, n_tensors, &tab, &blocks, "pv_adam_step_multi_dev_f32");
  if (rc) return rc;
  if (tab.n_tensors == 0) return PV_OK;
  hipLaunchKernelGGL(adam_step_multi_f32, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tab, 0.f, 0.f, 0.f, 1.f, 0.f,
                     0.f, grad_scale, scalars_dev);
  return check_launch("pv_adam_step_multi_dev_f32");
}

int pv_adam_step_bf16grad(float* param, const uint16_t* grad_bf16, float* exp_avg, float* exp_avg_sq,
                          uint16_t* bf16_shadow, size_t n, double lr, double beta1, double beta2, double eps, int32_t step,
                          float grad_scale, void* stream) {
  PV_REQUIRE(param && grad_bf16 && exp_avg && exp_avg_sq, PV_EINVAL, "pv_adam_step_bf16grad: null pointer");
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_adam_step_bf16grad: step must be >= 1");
  PV_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad_bf16 % 8 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                 ((uintptr_t)exp_avg_sq % 16 == 0) && ((uintptr_t)bf16_shadow % 8 == 0),
             PV_EINVAL, "pv_adam_step_bf16grad: buffers must be 16-byte aligned");
  if (n == 0) return PV_OK;
  double bc1 = 1.0 - pow(beta1, (double)step);
  double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_step_f32<uint16_t>, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), param,
                     grad_bf16, exp_avg, exp_avg_sq, bf16_shadow, n, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(-(lr / bc1)), grad_scale);
  return check_launch("pv_adam_step_bf16grad");
}

int pv_embedding_fwd_f32(const float* table, const int64_t* ids, float* out, int32_t n_ids, int32_t dim, int32_t n_rows,
                         void* stream) {
  PV_REQUIRE(table && ids && out, PV_EINVAL, "pv_embedding_fwd_f32: null pointer");
  PV_REQUIRE(n_ids > 0 && dim > 0 && n_rows > 0, PV_EINVAL, "pv_embedding_fwd_f32: bad sizes");
  hipLaunchKernelGGL(embedding_fwd_f32, dim3((unsigned)((n_ids * dim + 255) / 256)), dim3(256), 0, as_stream(stream), table,
                     ids, out, n_ids, dim, n_rows);
  return check_launch("pv_embedding_fwd_f32");
}

int pv_embedding_bwd_f32(const float* dout, const int64_t* ids, float* dtable, int32_t n_ids, int32_t dim, int32_t n_rows,
                         void* stream) {
  PV_REQUIRE(dout && ids && dtable, PV_EINVAL, "pv_embedding_bwd_f32: null pointer");
  PV_REQUIRE(n_ids > 0 && dim > 0 && n_rows > 0, PV_EINVAL, "pv_embedding_bwd_f32: bad sizes");
  hipLaunchKernelGGL(embedding_bwd_f32, dim3((unsigned)((n_rows * dim + 255) / 256)), dim3(256), 0, as_stream(stream), dout,
                     ids, dtable, n_ids, dim, n_rows);
  return check_launch("pv_embedding_bwd_f32");
}

}  // extern "C"
