// bf16 kernels for the one big fully-connected layer, fc1 (in_features = cnn_output_size ~ 1.0e6,
// out_features = 128; predict_pv_yield/models/conv3d/model.py:74-78,92,125).  fc1 holds 99.9 % of the
// model's parameters: every kernel here is a single streaming pass over the [N, K] weight (or gradient)
// matrix and is HBM-bound; the contraction runs on MFMA only so that the VALU stays out of the way.
//
//   fwd : y[M,N]  = x[M,K] . w[N,K]^T   split-K over workgroups, f32 slabs + fixed-order reduce
//   bwd : dx[M,K] = g[M,N] . w[N,K]     (bf16 out);  dw[N,K] = g^T . x (f32 out);  db[N] = colsum(g)
//         with g = dy ⊙ (y > 0)
#include "pv_common.h"
#include <string.h>

namespace pv {

// ---------------------------------------------------------------------------------------------
// forward: one workgroup = one K-range, 4 waves = 4 n-tiles of 32 output features, MT m-tiles
// of 32 rows.  k-permutation: inside a 64-deep k-block lane (r, h) owns k = 32*h + 8*s + j for
// k-step s, so each lane streams 64 contiguous bytes of its row; A and B use the same map.
// ---------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void linear_fwd_bf16_kernel(const uint16_t* __restrict__ x,
                                                               const uint16_t* __restrict__ w,
                                                               float* __restrict__ partial, int m, int n,
                                                               long long k, int kblocks_per_wg) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nt = blockIdx.y * 4 + wave;
  const long long total_kb = (k + 63) / 64;
  const long long kb0 = (long long)blockIdx.x * kblocks_per_wg;
  long long kb1 = kb0 + kblocks_per_wg;
  if (kb1 > total_kb) kb1 = total_kb;

  int wrow = nt * 32 + r;
  if (wrow > n - 1) wrow = n - 1;
  const uint16_t* wp = w + (size_t)wrow * k + 32 * hh;
  const uint16_t* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int xrow = mt * 32 + r;
    if (xrow > m - 1) xrow = m - 1;
    xp[mt] = x + (size_t)xrow * k + 32 * hh;
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[mt][j] = 0.f;

  const bf16x8 zero8 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f,
                        (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  // full 64-deep k-blocks: unconditional loads (4 x 16 B of W and MT x 4 x 16 B of x per lane in flight)
  long long kb_full = kb1;
  if (kb_full * 64 > k) kb_full = k / 64;
  if (kb_full < kb0) kb_full = kb0;
#pragma unroll 2
  for (long long kb = kb0; kb < kb_full; ++kb) {
    const long long kbase = kb * 64;
    bf16x8 wv[4];
    bf16x8 xv[MT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wv[s] = *reinterpret_cast<const bf16x8*>(wp + kbase + 8 * s);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xv[mt][s] = *reinterpret_cast<const bf16x8*>(xp[mt] + kbase + 8 * s);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv[mt][s], wv[s], acc[mt], 0, 0, 0);
  }
  // ragged tail block (k % 64 != 0): clamped address + select, never a branch around a load
  for (long long kb = kb_full; kb < kb1; ++kb) {
    const long long kbase = kb * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long long kk = kbase + 32 * hh + 8 * s;
      const bool ok = kk + 8 <= k;
      const long long kc = ok ? kbase + 8 * s : 0;
      bf16x8 wv = *reinterpret_cast<const bf16x8*>(wp + kc);
      wv = ok ? wv : zero8;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x8 xv = *reinterpret_cast<const bf16x8*>(xp[mt] + kc);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ok ? xv : zero8, wv, acc[mt], 0, 0, 0);
      }
    }
  }
  // C layout: col = lane&31 = output feature, row = sample
  const int col = nt * 32 + r;
  if (col < n) {
    float* dst = partial + (size_t)blockIdx.x * m * n + col;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (row < m) dst[(size_t)row * n] = acc[mt][j];
      }
  }
}

// y = relu?(bias + sum of split-K slabs), fixed order.  block = 64 outputs x 4 slab groups (one wave each)
// (blockIdx.y: row block of 32 rows whose slabs start blk_stride floats further on -- the LDS-staged forward takes 32 rows per
// launch, a call with more rows launches it per block and reduces ALL blocks here, once)
__global__ __launch_bounds__(256) void linear_reduce_bf16path(const float* __restrict__ partial,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int m, int n, int k_splits, int relu, size_t blk_stride = 0,
                                                               int rows_per_blk = 32) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  if (gridDim.y > 1) {
    partial += (size_t)blockIdx.y * blk_stride;
    y += (size_t)blockIdx.y * rows_per_blk * n;
    m = m - rows_per_blk * (int)blockIdx.y < rows_per_blk ? m - rows_per_blk * (int)blockIdx.y : rows_per_blk;
  }
  const size_t mn = (size_t)m * n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < m * n) {
    int k = grp;
    for (; k + 60 < k_splits; k += 64) {     // sixteen loads in flight (four trips of the loop below; same order of additions)
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = partial[(size_t)(k + 4 * j) * mn + i];
#pragma unroll
      for (int j = 0; j < 16; j += 4) s0 += v[j], s1 += v[j + 1], s2 += v[j + 2], s3 += v[j + 3];
    }
    for (; k + 12 < k_splits; k += 16) {
      s0 += partial[(size_t)k * mn + i];
      s1 += partial[(size_t)(k + 4) * mn + i];
      s2 += partial[(size_t)(k + 8) * mn + i];
      s3 += partial[(size_t)(k + 12) * mn + i];
    }
    for (; k < k_splits; k += 4) s0 += partial[(size_t)k * mn + i];
  }
  part[grp][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && i < m * n) {
    float s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (bias) s += bias[i % n];
    if (relu) s = s > 0.f ? s : 0.f;
    y[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// backward (f32 VALU outer products; both kernels stream [N,K] once per tile row and are
// bandwidth-bound on the 0.5 GB gradient / 0.26 GB weight matrices)
// ---------------------------------------------------------------------------------------------
constexpr int BT = 16;  // tile of the small dimension held in registers

// XCD-aware tile order for the two backward kernels.  Workgroups of a 1-D grid go round-robin over the 8 XCDs (each
// with its own L2); the `inner` tiles that re-read the same k-block of the big matrix are given consecutive slots of
// ONE XCD, so that block comes from HBM once and from that XCD's L2 afterwards.
constexpr int N_XCD = 8;
__device__ __forceinline__ bool xcd_tile(int inner, int n_kblocks, int* kblock, int* tile) {
  const int xcd = blockIdx.x % N_XCD, slot = blockIdx.x / N_XCD;
  *kblock = (slot / inner) * N_XCD + xcd;
  *tile = slot % inner;
  return *kblock < n_kblocks;
}
static unsigned xcd_grid(unsigned n_kblocks, unsigned inner) { return ((n_kblocks + N_XCD - 1) / N_XCD) * N_XCD * inner; }

// dx[m0..m0+7][k..k+7] = sum_n g[m][n] * w[n][k..k+7]
__global__ __launch_bounds__(256) void linear_bwd_dx_bf16_kernel(const uint16_t* __restrict__ w,
                                                                  const float* __restrict__ dy,
                                                                  const float* __restrict__ ymask,
                                                                  uint16_t* __restrict__ dx, int m, int n, long long k) {
  extern __shared__ float g[];  // [n][BT]
  int kblock, mtile;
  if (!xcd_tile((m + BT - 1) / BT, (int)((k / 8 + 255) / 256), &kblock, &mtile)) return;
  const int m0 = mtile * BT;
  for (int i = threadIdx.x; i < n * BT; i += blockDim.x) {
    int col = i / BT, rr = i % BT;
    float v = 0.f;
    if (m0 + rr < m) {
      size_t off = (size_t)(m0 + rr) * n + col;
      v = dy[off];
      if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    }
    g[i] = v;
  }
  __syncthreads();
  const long long k8 = ((long long)kblock * blockDim.x + threadIdx.x) * 8;
  if (k8 >= k) return;
  float acc[BT][8];
#pragma unroll
  for (int i = 0; i < BT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  for (int col = 0; col < n; ++col) {
    u32x4 raw = *reinterpret_cast<const u32x4*>(w + (size_t)col * k + k8);
    float wv[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wv[2 * q] = __builtin_bit_cast(float, raw[q] << 16);
      wv[2 * q + 1] = __builtin_bit_cast(float, raw[q] & 0xffff0000u);
    }
    f32x4 gq[BT / 4];
#pragma unroll
    for (int q = 0; q < BT / 4; ++q) gq[q] = *reinterpret_cast<const f32x4*>(g + col * BT + 4 * q);
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const float gv = gq[i >> 2][i & 3];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(gv, wv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < BT; ++i) {
    if (m0 + i < m) {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        o[q] = pack_bf16_pair(acc[i][2 * q], acc[i][2 * q + 1]);
      *reinterpret_cast<u32x4*>(dx + (size_t)(m0 + i) * k + k8) = o;
    }
  }
}

// dw[n0..n0+7][k..k+7] = sum_m g[m][n] * x[m][k..k+7]
// FUSE_ADAM: the gradient tile never leaves registers -- the Adam update of the same 8 x 8 weights (and of the bf16
// shadow) is applied in place: one pass over p, m, v instead of writing 0.5 GB of dw and reading it back.
struct AdamScalars {
  float one_minus_b1, beta2, one_minus_b2, bc2_sqrt, eps, neg_step_size;
};

// XT: uint16_t = x in bf16 (the bf16 model); float = x in f32 (precision="fp32": exact f32 products, f32 accumulation in batch
// order -- the fused form of that model's fc1 weight gradient + Adam, pv_linear_wgrad_adam_f32)
template <int MODE, typename XT = uint16_t>  // 0: write dw f32; 1: fused Adam update (dw = parameter); 2: write dw bf16 (dw reinterpreted)
__global__ __launch_bounds__(256) void linear_bwd_dw_bf16_kernel(const XT* __restrict__ x,
                                                                  const float* __restrict__ dy,
                                                                  const float* __restrict__ ymask,
                                                                  float* __restrict__ dw, int m, int n, long long k,
                                                                  float* __restrict__ exp_avg,
                                                                  float* __restrict__ exp_avg_sq,
                                                                  uint16_t* __restrict__ shadow, AdamScalars ad) {
  extern __shared__ float g[];  // [m][BT]
  int kblock, ntile;
  if (!xcd_tile((n + BT - 1) / BT, (int)((k / 8 + 255) / 256), &kblock, &ntile)) return;
  const int n0 = ntile * BT;
  for (int i = threadIdx.x; i < m * BT; i += blockDim.x) {
    int rr = i / BT, j = i % BT;
    float v = 0.f;
    if (n0 + j < n) {
      size_t off = (size_t)rr * n + n0 + j;
      v = dy[off];
      if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    }
    g[i] = v;
  }
  __syncthreads();
  const long long k8 = ((long long)kblock * blockDim.x + threadIdx.x) * 8;
  if (k8 >= k) return;
  // the 32 x (BT x 8) FMAs per thread run as packed f32 (v_pk_fma_f32, the gradient value broadcast to both halves):
  // as plain v_fma_f32 they cost ~125 us of SIMD time per pass over the matrix, a fifth of this HBM-bound kernel
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc2[BT][4];
#pragma unroll
  for (int i = 0; i < BT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x2_t){0.f, 0.f};
#pragma unroll 4
  for (int rr = 0; rr < m; ++rr) {
    f32x2_t xv2[4];
    if constexpr (sizeof(XT) == 4) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(x + (size_t)rr * k + k8), hi = *reinterpret_cast<const f32x4*>(x + (size_t)rr * k + k8 + 4);
      xv2[0] = (f32x2_t){lo[0], lo[1]}, xv2[1] = (f32x2_t){lo[2], lo[3]}, xv2[2] = (f32x2_t){hi[0], hi[1]}, xv2[3] = (f32x2_t){hi[2], hi[3]};
    } else {
      u32x4 raw = *reinterpret_cast<const u32x4*>(x + (size_t)rr * k + k8);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xv2[q] = (f32x2_t){__builtin_bit_cast(float, raw[q] << 16), __builtin_bit_cast(float, raw[q] & 0xffff0000u)};
    }
    f32x4 gq[BT / 4];
#pragma unroll
    for (int q = 0; q < BT / 4; ++q) gq[q] = *reinterpret_cast<const f32x4*>(g + rr * BT + 4 * q);
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      const float gv = gq[i >> 2][i & 3];
      const f32x2_t g2 = {gv, gv};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[i][j] = __builtin_elementwise_fma(g2, xv2[j], acc2[i][j]);
    }
  }
  float acc[BT][8];
#pragma unroll
  for (int i = 0; i < BT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][2 * j] = acc2[i][j][0], acc[i][2 * j + 1] = acc2[i][j][1];
  if constexpr (MODE != 1) {
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      if (n0 + i < n) {
        const size_t off = (size_t)(n0 + i) * k + k8;
        if constexpr (MODE == 0) {
          f32x4 o0 = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
          f32x4 o1 = {acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
          *reinterpret_cast<f32x4*>(dw + off) = o0;
          *reinterpret_cast<f32x4*>(dw + off + 4) = o1;
        } else {
          u32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            o[q] = pack_bf16_pair(acc[i][2 * q], acc[i][2 * q + 1]);
          *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(dw) + off) = o;
        }
      }
    }
  } else {
    // dw here is the PARAMETER (updated in place); same operation order as adam_step_f32 / torch.  The correctly rounded
    // sqrt and the two divisions expand to ~40 VALU instructions per element (a few thousand cycles per row of the
    // tile), so the NEXT row's p / m / v are fetched before the current row's arithmetic: without that a wave has its
    // 6 loads in flight only between rows and the HBM pipe drains while it computes.
    f32x4 nxt[6];
    auto fetch = [&](int i) {
      const size_t off = (size_t)min(n0 + i, n - 1) * k + k8;
      nxt[0] = *reinterpret_cast<const f32x4*>(dw + off);
      nxt[1] = *reinterpret_cast<const f32x4*>(dw + off + 4);
      nxt[2] = *reinterpret_cast<const f32x4*>(exp_avg + off);
      nxt[3] = *reinterpret_cast<const f32x4*>(exp_avg + off + 4);
      nxt[4] = *reinterpret_cast<const f32x4*>(exp_avg_sq + off);
      nxt[5] = *reinterpret_cast<const f32x4*>(exp_avg_sq + off + 4);
    };
    fetch(0);
#pragma unroll
    for (int i = 0; i < BT; ++i) {
      float pv[8], mv[8], vv[8];
      *reinterpret_cast<f32x4*>(pv) = nxt[0];
      *reinterpret_cast<f32x4*>(pv + 4) = nxt[1];
      *reinterpret_cast<f32x4*>(mv) = nxt[2];
      *reinterpret_cast<f32x4*>(mv + 4) = nxt[3];
      *reinterpret_cast<f32x4*>(vv) = nxt[4];
      *reinterpret_cast<f32x4*>(vv + 4) = nxt[5];
      if (i + 1 < BT) fetch(i + 1);
      if (n0 + i < n) {
        const size_t off = (size_t)(n0 + i) * k + k8;
        uint32_t sh[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gr = acc[i][j];
          const float mm = mv[j] + ad.one_minus_b1 * (gr - mv[j]);
          const float v2 = vv[j] * ad.beta2 + (ad.one_minus_b2 * gr) * gr;
          const float denom = sqrtf(v2) / ad.bc2_sqrt + ad.eps;
          const float pp = pv[j] + ad.neg_step_size * (mm / denom);
          mv[j] = mm; vv[j] = v2; pv[j] = pp;
          if (j & 1) sh[j >> 1] |= (uint32_t)f32_to_bf16_bits(pp) << 16; else sh[j >> 1] = f32_to_bf16_bits(pp);
        }
        *reinterpret_cast<f32x4*>(dw + off) = *reinterpret_cast<const f32x4*>(pv);
        *reinterpret_cast<f32x4*>(dw + off + 4) = *reinterpret_cast<const f32x4*>(pv + 4);
        *reinterpret_cast<f32x4*>(exp_avg + off) = *reinterpret_cast<const f32x4*>(mv);
        *reinterpret_cast<f32x4*>(exp_avg + off + 4) = *reinterpret_cast<const f32x4*>(mv + 4);
        *reinterpret_cast<f32x4*>(exp_avg_sq + off) = *reinterpret_cast<const f32x4*>(vv);
        *reinterpret_cast<f32x4*>(exp_avg_sq + off + 4) = *reinterpret_cast<const f32x4*>(vv + 4);
        if (shadow) {
          u32x4 so = {sh[0], sh[1], sh[2], sh[3]};
          *reinterpret_cast<u32x4*>(shadow + off) = so;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// fc1 backward in ONE pass over the matrix (single GPU): weight gradient + Adam update (as linear_bwd_dw_bf16_kernel<1>)
// AND dx = g . W from the very weights the pass streams anyway.  The separate dx kernel re-reads the 257 MB bf16 operand
// copy of the matrix (84-93 us of the 1.83 ms step) shortly before the update streams the f32 master of the same matrix.
// Tile = ALL n <= 128 rows x 128 k-columns per workgroup (dx needs every row of a k-range): thread (kq = tid % 16,
// rg = tid / 16) owns rows 8 rg .. +7 x columns 8 kq .. +7.  x [m <= 32][128] (bf16) and g [32][128] (f32) sit in LDS; the
// 8 x 8 gradient tile is formed in registers and applied row by row like the two-kernel path (same operation order:
// bit-identical parameters / moments / operand copy); the PRE-update weights are rounded to bf16 -- the values the operand
// copy held during the forward -- and parked transposed in LDS ([k][n]) as the B operand of dx = g W on the bf16 matrix
// cores, g as a bf16 hi + lo pair (two MFMAs per step, ~16 mantissa bits) exactly like linear_bwd_dx_bf16_v2_kernel; the dx
// tile leaves through LDS as 256-byte runs.  The x tile and the outgoing dx tile share bytes with the weight tile: 50 KB
// of LDS and 152 registers = three workgroups per CU (the update pass needs the occupancy to keep HBM busy while a
// workgroup sits in its prologue / matrix tail).  Measured alone: 684 us against 620 us for the update-only kernel plus
// 85-99 us for the dx kernel it replaces.
// ---------------------------------------------------------------------------------------------
constexpr int FD_KT = 128;                 // k-columns per workgroup
constexpr int FD_XLD = FD_KT + 8;          // bf16 row stride of the x / dx tile
constexpr int FD_WLD = 128 + 8;            // bf16 row stride of the transposed weight tile [k][n]

// AHEAD rows of p / m / v in flight per thread; EARLY: the first of them requested at the top of the kernel
// GR: rows of the gradient tile formed at a time (8: all at once; 4 / 2: in halves / quarters, each right before its Adam rows)
template <int AHEAD, bool EARLY, int GR = 8>
__global__ __launch_bounds__(256, 3) void linear_bwd_dw_dx_adam_kernel(
    const uint16_t* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ ymask, float* __restrict__ w,
    int m, int n, long long k, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, uint16_t* __restrict__ shadow,
    uint16_t* __restrict__ dx, float* __restrict__ db, AdamScalars ad, int gate_dx, const float* __restrict__ ad_dev,
    int mv_tiled) {
  // ad_dev (may be NULL): the six Adam scalars in device memory (pv_adam_scalars_advance) -- the form a captured HIP graph
  // replays, where the bias corrections must change from replay to replay and kernel arguments cannot
  if (ad_dev) ad = AdamScalars{ad_dev[0], ad_dev[1], ad_dev[2], ad_dev[3], ad_dev[4], ad_dev[5]};
  __shared__ __attribute__((aligned(16))) float gs[32 * 128];               // g = dy (.) relu'  [b][n], zero padded
  __shared__ __attribute__((aligned(16))) uint16_t wt[FD_KT * FD_WLD];      // pre-update weights, bf16, [k][n]
  // the x tile [b][k] lives in the first 8.5 KB of wt until the gradient tile is formed (wt is written after that), and the
  // dx tile [b][k] goes through the same bytes on its way out
  uint16_t* xs = wt;
  static_assert(32 * FD_XLD <= FD_KT * FD_WLD, "x tile does not fit inside the weight tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = tid & 15, rg = tid >> 4;
  const long long k0 = (long long)blockIdx.x * FD_KT;
  const long long k8 = k0 + 8 * kq;
  const bool k_ok = k8 < k;
  // the first rows' p / m / v are requested before anything else (round 5): they used to be requested after the gradient
  // tile was formed, so a workgroup had no load of the 3.4 GB stream in flight during its prologue and its ~4 000 cycles of
  // gradient FMAs
  f32x4 nxt[AHEAD][6];      // the rows in flight (EARLY = false, AHEAD = 1: one row, requested after the gradient tile -- rounds 3-4)
  // mv_tiled: the two moment arrays are stored tile by tile, [k / 128][n][128]: a workgroup's share is ONE contiguous
  // n x 512-byte block per array instead of n segments 4 MB apart
  const size_t mv_tile_base = (size_t)blockIdx.x * (size_t)n * FD_KT + 8 * kq;
  auto fetch = [&](int i) {
    const int r = min(8 * rg + i, n - 1);
    const size_t off = (size_t)r * k + (k_ok ? k8 : 0);
    const size_t moff = mv_tiled ? mv_tile_base + (size_t)r * FD_KT : off;
    f32x4 (&d)[6] = nxt[i % AHEAD];
    d[0] = *reinterpret_cast<const f32x4*>(w + off);
    d[1] = *reinterpret_cast<const f32x4*>(w + off + 4);
    d[2] = *reinterpret_cast<const f32x4*>(exp_avg + moff);
    d[3] = *reinterpret_cast<const f32x4*>(exp_avg + moff + 4);
    d[4] = *reinterpret_cast<const f32x4*>(exp_avg_sq + moff);
    d[5] = *reinterpret_cast<const f32x4*>(exp_avg_sq + moff + 4);
  };
  if constexpr (EARLY) {
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) fetch(i);
  }
  for (int i = tid; i < 32 * 128; i += 256) {
    const int b = i >> 7, nn = i & 127;
    float v = 0.f;
    if (b < m && nn < n) {
      const size_t off = (size_t)b * n + nn;
      v = dy[off];
      if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    }
    gs[i] = v;
  }
  for (int i = tid; i < 32 * (FD_KT / 8); i += 256) {          // 16-byte chunks of the x tile
    const int b = i / (FD_KT / 8), c = i - b * (FD_KT / 8);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (b < m && k0 + 8 * c < k) v = *reinterpret_cast<const u32x4*>(x + (size_t)b * k + k0 + 8 * c);
    *reinterpret_cast<u32x4*>(xs + b * FD_XLD + 8 * c) = v;
  }
  __syncthreads();
  if (db && blockIdx.x == 0 && tid < n) {    // db[n] = column sums of g (index order), by the first workgroup
    float sacc = 0.f;
    for (int b = 0; b < m; ++b) sacc += gs[b * 128 + tid];
    db[tid] = sacc;
  }
  // ---- weight gradient tile: rows 8 rg .. +7, columns k8 .. +7 (packed f32 FMAs, gradient value broadcast) ----------------
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc2[GR][4];
  auto form_gradient = [&](int r0) __attribute__((always_inline)) {      // rows 8 rg + r0 .. + GR - 1
#pragma unroll
    for (int i = 0; i < GR; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x2_t){0.f, 0.f};
#pragma unroll 4
    for (int b = 0; b < m; ++b) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(xs + b * FD_XLD + 8 * kq);
      f32x2_t xv2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        xv2[q] = (f32x2_t){__builtin_bit_cast(float, raw[q] << 16), __builtin_bit_cast(float, raw[q] & 0xffff0000u)};
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gs + b * 128 + 8 * rg + (r0 & 4));
      f32x4 g1 = g0;
      if constexpr (GR == 8) g1 = *reinterpret_cast<const f32x4*>(gs + b * 128 + 8 * rg + 4);
#pragma unroll
      for (int i = 0; i < GR; ++i) {
        const float gv = GR == 8 ? (i < 4 ? g0[i] : g1[i - 4]) : g0[(r0 & 3) + i];
        const f32x2_t g2 = {gv, gv};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = __builtin_elementwise_fma(g2, xv2[j], acc2[i][j]);
      }
    }
  };
  form_gradient(0);
  // ---- Adam, row by row (next rows' p / m / v in flight under this row's arithmetic); the old weights are kept as bf16 ------
  uint32_t wold[8][4];     // bf16 pairs of the PRE-update weights: [row i][column pair]
  if constexpr (!EARLY) {
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) fetch(i);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if constexpr (GR < 8) {
      if (i > 0 && i % GR == 0) form_gradient(i);
    }
    float pv[8], mv[8], vv[8];
    *reinterpret_cast<f32x4*>(pv) = nxt[i % AHEAD][0];
    *reinterpret_cast<f32x4*>(pv + 4) = nxt[i % AHEAD][1];
    *reinterpret_cast<f32x4*>(mv) = nxt[i % AHEAD][2];
    *reinterpret_cast<f32x4*>(mv + 4) = nxt[i % AHEAD][3];
    *reinterpret_cast<f32x4*>(vv) = nxt[i % AHEAD][4];
    *reinterpret_cast<f32x4*>(vv + 4) = nxt[i % AHEAD][5];
    if (i + AHEAD < 8) fetch(i + AHEAD);
    const bool row_ok = 8 * rg + i < n && k_ok;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      wold[i][q] = row_ok ? (pack_bf16_pair(pv[2 * q], pv[2 * q + 1])) : 0u;
    if (row_ok) {
      const size_t off = (size_t)(8 * rg + i) * k + k8;
      uint32_t sh[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gr = acc2[i % GR][j >> 1][j & 1];
        const float mm = mv[j] + ad.one_minus_b1 * (gr - mv[j]);
        const float v2 = vv[j] * ad.beta2 + (ad.one_minus_b2 * gr) * gr;
        const float denom = sqrtf(v2) / ad.bc2_sqrt + ad.eps;
        const float pp = pv[j] + ad.neg_step_size * (mm / denom);
        mv[j] = mm; vv[j] = v2; pv[j] = pp;
        if (j & 1) sh[j >> 1] |= (uint32_t)f32_to_bf16_bits(pp) << 16; else sh[j >> 1] = f32_to_bf16_bits(pp);
      }
      *reinterpret_cast<f32x4*>(w + off) = *reinterpret_cast<const f32x4*>(pv);
      *reinterpret_cast<f32x4*>(w + off + 4) = *reinterpret_cast<const f32x4*>(pv + 4);
      const size_t moff = mv_tiled ? mv_tile_base + (size_t)(8 * rg + i) * FD_KT : off;
      *reinterpret_cast<f32x4*>(exp_avg + moff) = *reinterpret_cast<const f32x4*>(mv);
      *reinterpret_cast<f32x4*>(exp_avg + moff + 4) = *reinterpret_cast<const f32x4*>(mv + 4);
      *reinterpret_cast<f32x4*>(exp_avg_sq + moff) = *reinterpret_cast<const f32x4*>(vv);
      *reinterpret_cast<f32x4*>(exp_avg_sq + moff + 4) = *reinterpret_cast<const f32x4*>(vv + 4);
      if (shadow) {
        u32x4 so = {sh[0], sh[1], sh[2], sh[3]};
        *reinterpret_cast<u32x4*>(shadow + off) = so;
      }
    }
  }
  if (!dx) return;
  __syncthreads();   // every thread is done reading the x tile: its bytes now become part of wt
  // transposed into wt: column kk = 8 kq + j holds rows 8 rg .. +7 as one 16-byte piece
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u32x4 piece;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t lo = (wold[2 * q][j >> 1] >> (16 * (j & 1))) & 0xffffu;
      const uint32_t hi = (wold[2 * q + 1][j >> 1] >> (16 * (j & 1))) & 0xffffu;
      piece[q] = lo | (hi << 16);
    }
    *reinterpret_cast<u32x4*>(wt + (8 * kq + j) * FD_WLD + 8 * rg) = piece;
  }
  __syncthreads();
  // ---- dx[b][k0 + 32 wave + col] = sum_n g[b][n] w_old[n][k]: A = g rows b (hi + lo), B = wt rows k ---------------------------
  const int col = lane & 31, half = lane >> 5;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(gs + col * 128 + 16 * st + 8 * half);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(gs + col * 128 + 16 * st + 8 * half + 4);
    const float gx[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    u32x4 hw, lw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint16_t h0 = f32_to_bf16_bits(gx[2 * q]), h1 = f32_to_bf16_bits(gx[2 * q + 1]);
      const uint16_t l0 = f32_to_bf16_bits(gx[2 * q] - bf16_bits_to_f32(h0)), l1 = f32_to_bf16_bits(gx[2 * q + 1] - bf16_bits_to_f32(h1));
      hw[q] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      lw[q] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    const bf16x8 bw = *reinterpret_cast<const bf16x8*>(wt + (32 * wave + col) * FD_WLD + 16 * st + 8 * half);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, lw), bw, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hw), bw, o, 0, 0, 0);
  }
  __syncthreads();   // all four waves are done reading wt / gs: the dx tile [b][k] is assembled in wt's first bytes
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int b = (r & 3) + 8 * (r >> 2) + 4 * half;
    xs[b * FD_XLD + 32 * wave + col] = f32_to_bf16_bits(o[r]);
  }
  __syncthreads();
  for (int i = tid; i < 32 * (FD_KT / 8); i += 256) {          // 16-byte chunks: 256 contiguous bytes per row of dx
    const int b = i / (FD_KT / 8), c = i - b * (FD_KT / 8);
    if (b < m && k0 + 8 * c < k) {
      u32x4 o = *reinterpret_cast<const u32x4*>(xs + b * FD_XLD + 8 * c);
      if (gate_dx) {
        // x is a ReLU output: (x > 0) is its producer's ReLU derivative.  Applying it here (x of this tile is L2-resident)
        // hands the producer an already-gated gradient: the NCDHW -> NDHWC repack that follows reads dx only, not dx + x
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + (size_t)b * k + k0 + 8 * c);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          o[q] &= __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, relu_pair01(xv[q])) * (u16x2_t){0xffff, 0xffff});
      }
      *reinterpret_cast<u32x4*>(dx + (size_t)b * k + k0 + 8 * c) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same pass for the K-SHARDED fc1 (distributed.py, HipAdam large_grad_mode "ksharded"): this rank's COLUMN shard of the
// weight, [n <= 128][k = K / W], and ALL samples of the global batch: m = W x per-GPU batch rows (256 at 8 x 32), taken in
// blocks of 32.  A workgroup owns 128 k-columns as above and walks the row blocks twice: once to accumulate the gradient tile
// on the matrix cores (x blocks staged in LDS and read transposed, g^T as three-term bf16 fragments), then -- after the Adam
// rows, with the pre-update weights parked transposed in LDS -- once more for dx, block by block.  The outgoing dx block has LDS
// bytes of its own (the weight tile must survive all blocks): 59 KB, two workgroups per CU.  dy arrives already gated (the
// all-gathered g = dy (.) relu' of every rank) and is turned into operand fragments once per call (tall_split_g_kernel); the
// gradient is scaled by grad_scale (1 / world: the mean of DDP's all-reduce) on its way into Adam, dx is not.
// History (512 rows x 125 440 columns, one rank of 8 at 64 samples per GPU): eight dx launches + the register-tiled weight gradient
// + Adam 0.9 ms; one pass with the gradient tile on the vector ALU 655 us; gradient on the matrix cores with g converted per
// workgroup 408 us; with finished fragments: see profiles/r06/NOTES.md section 4.
// ---------------------------------------------------------------------------------------------
constexpr int TL_XS = 320;      // bytes per row of an x block [32][128] bf16 (rows 64 B apart mod 256: conflict-free transposed reads)
constexpr int TL_TS = 64 + 4;   // floats per row of a wave's transpose patch [32 rows][64 columns]
constexpr int TL_GA = 2 * 4 * 3 * 64, TL_GD = 2 * 8 * 64;      // 16-byte operand fragments per block of 32 rows: gradient / dx
typedef __attribute__((address_space(3))) s16x4 tl_lds_s16x4_t;
static_assert(32 * TL_XS <= FD_KT * FD_WLD * 2 && 4 * 32 * TL_TS * 4 <= FD_KT * FD_WLD * 2, "x block / transpose patches live in the weight tile's bytes");

// g [m][n] (f32, already gated) as ready-made matrix-core operands, ONCE per call: every one of the shard's ~1 000 workgroups needs
// the same fragments, and converting them per workgroup was most of the first matrix-core form's time (408 us at 512 rows, the
// dx walk's column-strided f32 reads of g at 16-way bank conflicts the rest).  Per block mb of 32 rows:
//   gradient (A = g^T): fragment ((mb, ks, wave), term t, lane (r, hh)) = bf16 term t of g[32 mb + 16 ks + 8 hh + j][32 wave + r],
//     j = 0 .. 7 -- THREE terms (an f32 splits exactly into three bf16);
//   dx (A = g): fragment ((mb, term, st), lane (col, half)) = g[32 mb + col][16 st + 8 half + j] as hi / lo (two terms, ~16 bits,
//     as linear_bwd_dx_bf16_v2_kernel).
// Rows beyond m and outputs beyond n are zeros.
__global__ __launch_bounds__(256) void tall_split_g_kernel(const float* __restrict__ dy, int m, int n, u32x4* __restrict__ ga,
                                                            u32x4* __restrict__ gd, int nblk) {
  const int total = nblk * (TL_GA + TL_GD);
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < total; f += gridDim.x * blockDim.x) {
    const int mb = f / (TL_GA + TL_GD), i = f - mb * (TL_GA + TL_GD);
    float v[8];
    int term;
    if (i < TL_GA) {      // ((ks * 4 + wave) * 3 + term) * 64 + lane
      const int lane = i & 63, q = i >> 6, ks = q / 12, wave = (q / 3) & 3, r = lane & 31, hh = lane >> 5;
      term = q % 3;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = 32 * mb + 16 * ks + 8 * hh + j, nn = 32 * wave + r;
        v[j] = (b < m && nn < n) ? dy[(size_t)b * n + nn] : 0.f;
      }
    } else {              // (term * 8 + st) * 64 + lane
      const int i2 = i - TL_GA, lane = i2 & 63, q = i2 >> 6, st = q & 7, col = lane & 31, half = lane >> 5;
      term = q >> 3;
      const int b = 32 * mb + col;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int nn = 16 * st + 8 * half + j;
        v[j] = (b < m && nn < n) ? dy[(size_t)b * n + nn] : 0.f;
      }
    }
    u32x4 o;
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      uint16_t e[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float rem = v[2 * q2 + h];
        uint16_t bits = f32_to_bf16_bits(rem);
        for (int t = 0; t < term; ++t) {
          rem -= bf16_bits_to_f32(bits);
          bits = f32_to_bf16_bits(rem);
        }
        e[h] = bits;
      }
      o[q2] = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
    }
    (i < TL_GA ? ga + (size_t)mb * TL_GA + i : gd + (size_t)mb * TL_GD + (i - TL_GA))[0] = o;
  }
}

template <int AHEAD>
__global__ __launch_bounds__(256, 2) void linear_bwd_dw_dx_adam_tall_kernel(
    const uint16_t* __restrict__ x, const u32x4* __restrict__ ga, const u32x4* __restrict__ gd, float* __restrict__ w, int m, int n,
    long long k, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, uint16_t* __restrict__ shadow,
    uint16_t* __restrict__ dx, AdamScalars ad, int gate_dx, float grad_scale, int mv_tiled) {
  __shared__ __attribute__((aligned(16))) u32x4 gfr[TL_GD];                 // the dx walk's A fragments of a block (16 KB)
  __shared__ __attribute__((aligned(16))) uint16_t wt[FD_KT * FD_WLD];      // pre-update weights, bf16, [k][n]
  __shared__ __attribute__((aligned(16))) uint16_t dxs[32 * FD_XLD];        // the outgoing dx block [b][k]
  // wt's bytes hold the x blocks [b][k] (rows of TL_XS bytes) while the gradient is accumulated, then the waves' transpose patches;
  // wt itself is written after the Adam rows
  unsigned char* xsb = reinterpret_cast<unsigned char*>(wt);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = tid & 15, rg = tid >> 4;
  const long long k0 = (long long)blockIdx.x * FD_KT;
  const long long k8 = k0 + 8 * kq;
  const bool k_ok = k8 < k;
  f32x4 nxt[AHEAD][6];
  // mv_tiled: the moments in the [k / 128][n][128] tile layout of the single-GPU one-pass kernel (the same optimiser state serves
  // both kernels: per-GPU batches beyond 32 rows take this one)
  const size_t mv_tile_base = (size_t)blockIdx.x * (size_t)n * FD_KT + 8 * kq;
  auto fetch = [&](int i) {
    const int r = min(8 * rg + i, n - 1);
    const size_t off = (size_t)r * k + (k_ok ? k8 : 0);
    const size_t moff = mv_tiled ? mv_tile_base + (size_t)r * FD_KT : off;
    f32x4 (&d)[6] = nxt[i % AHEAD];
    d[0] = *reinterpret_cast<const f32x4*>(w + off);
    d[1] = *reinterpret_cast<const f32x4*>(w + off + 4);
    d[2] = *reinterpret_cast<const f32x4*>(exp_avg + moff);
    d[3] = *reinterpret_cast<const f32x4*>(exp_avg + moff + 4);
    d[4] = *reinterpret_cast<const f32x4*>(exp_avg_sq + moff);
    d[5] = *reinterpret_cast<const f32x4*>(exp_avg_sq + moff + 4);
  };
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) fetch(i);
  const int nblk = (m + 31) / 32;
  // ---- weight gradient tile on the matrix cores (round 6; the vector-ALU form took 655 us for 512 rows where the shard's bytes need
  // ~150): dW[n][k] += g^T[n][b] x[b][k] per block of 32 rows -- A = g^T rows 32 wave .. + 31 as three bf16 terms (exact: only the
  // order of the f32 additions differs from the register-tiled kernels), fetched as finished fragments; B = the x block read
  // transposed from LDS (ds_read_b64_tr_b16), four 32-column tiles per wave.  The next block's x and fragments are requested
  // before this block's products. ---------------------------------------------------------------------------------------------
  const int G = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int tr_off = (8 * (G >> 1) + tq) * TL_XS + (16 * (G & 1) + 4 * tp) * 2;
  u32x4 xr[2], ar[2][3];
  auto prefetch = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i, bl = id >> 4, c = id & 15, b = 32 * mb + bl;
      xr[i] = (u32x4){0u, 0u, 0u, 0u};
      if (mb < nblk && b < m && k0 + 8 * c < k) xr[i] = *reinterpret_cast<const u32x4*>(x + (size_t)b * k + k0 + 8 * c);
    }
    const u32x4* src = ga + (size_t)min(mb, nblk - 1) * TL_GA + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 3; ++t) ar[ks][t] = src[((ks * 4 + wave) * 3 + t) * 64];
  };
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
  prefetch(0);
  for (int mb = 0; mb < nblk; ++mb) {
    if (mb) __syncthreads();      // every wave is through with the previous block's x
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i;
      *reinterpret_cast<u32x4*>(xsb + (id >> 4) * TL_XS + 16 * (id & 15)) = xr[i];
    }
    bf16x8 a3[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 3; ++t) a3[ks][t] = __builtin_bit_cast(bf16x8, ar[ks][t]);
    __syncthreads();
    prefetch(mb + 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned char* bp = xsb + tr_off + ks * 16 * TL_XS + c * 64;
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tl_lds_s16x4_t*)(bp));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tl_lds_s16x4_t*)(bp + 4 * TL_XS));
        const __attribute__((ext_vector_type(8))) short b8s = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const bf16x8 b8 = __builtin_bit_cast(bf16x8, b8s);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[ks][2], b8, acc[c], 0, 0, 0);      // smallest term first
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[ks][1], b8, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[ks][0], b8, acc[c], 0, 0, 0);
      }
    }
  }
  // the tile reaches the Adam threads -- thread (kq, rg) owns rows 8 rg .. + 7 x columns 8 kq .. + 7, and rows 32 wave .. + 31 are
  // this wave's own -- through a patch of [32][64] floats per wave, one column half at a time
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t acc2[8][4];
  {
    const int r = lane & 31, hh = lane >> 5;
    float* patch = reinterpret_cast<float*>(wt) + wave * (32 * TL_TS);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      __syncthreads();      // the x block's last readers / the previous half's readers are done
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) patch[((j & 3) + 8 * (j >> 2) + 4 * hh) * TL_TS + 32 * t + r] = acc[2 * cc + t][j];
      __syncthreads();
      if ((kq >> 3) == cc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float* src = patch + (8 * (rg & 3) + i) * TL_TS + 8 * (kq & 7);
          const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
          acc2[i][0] = (f32x2_t){lo[0], lo[1]};
          acc2[i][1] = (f32x2_t){lo[2], lo[3]};
          acc2[i][2] = (f32x2_t){hi[0], hi[1]};
          acc2[i][3] = (f32x2_t){hi[2], hi[3]};
        }
      }
    }
  }
  // ---- Adam, row by row; the old weights are kept as bf16 ----------------------------------------------------------------------
  uint32_t wold[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float pv[8], mv[8], vv[8];
    *reinterpret_cast<f32x4*>(pv) = nxt[i % AHEAD][0];
    *reinterpret_cast<f32x4*>(pv + 4) = nxt[i % AHEAD][1];
    *reinterpret_cast<f32x4*>(mv) = nxt[i % AHEAD][2];
    *reinterpret_cast<f32x4*>(mv + 4) = nxt[i % AHEAD][3];
    *reinterpret_cast<f32x4*>(vv) = nxt[i % AHEAD][4];
    *reinterpret_cast<f32x4*>(vv + 4) = nxt[i % AHEAD][5];
    if (i + AHEAD < 8) fetch(i + AHEAD);
    const bool row_ok = 8 * rg + i < n && k_ok;
#pragma unroll
    for (int q = 0; q < 4; ++q) wold[i][q] = row_ok ? (pack_bf16_pair(pv[2 * q], pv[2 * q + 1])) : 0u;
    if (row_ok) {
      const size_t off = (size_t)(8 * rg + i) * k + k8;
      uint32_t sh[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gr = __fmul_rn(acc2[i][j >> 1][j & 1], grad_scale);
        const float mm = mv[j] + ad.one_minus_b1 * (gr - mv[j]);
        const float v2 = vv[j] * ad.beta2 + (ad.one_minus_b2 * gr) * gr;
        const float denom = sqrtf(v2) / ad.bc2_sqrt + ad.eps;
        const float pp = pv[j] + ad.neg_step_size * (mm / denom);
        mv[j] = mm; vv[j] = v2; pv[j] = pp;
        if (j & 1) sh[j >> 1] |= (uint32_t)f32_to_bf16_bits(pp) << 16; else sh[j >> 1] = f32_to_bf16_bits(pp);
      }
      *reinterpret_cast<f32x4*>(w + off) = *reinterpret_cast<const f32x4*>(pv);
      *reinterpret_cast<f32x4*>(w + off + 4) = *reinterpret_cast<const f32x4*>(pv + 4);
      const size_t moff = mv_tiled ? mv_tile_base + (size_t)(8 * rg + i) * FD_KT : off;
      *reinterpret_cast<f32x4*>(exp_avg + moff) = *reinterpret_cast<const f32x4*>(mv);
      *reinterpret_cast<f32x4*>(exp_avg + moff + 4) = *reinterpret_cast<const f32x4*>(mv + 4);
      *reinterpret_cast<f32x4*>(exp_avg_sq + moff) = *reinterpret_cast<const f32x4*>(vv);
      *reinterpret_cast<f32x4*>(exp_avg_sq + moff + 4) = *reinterpret_cast<const f32x4*>(vv + 4);
      if (shadow) {
        u32x4 so = {sh[0], sh[1], sh[2], sh[3]};
        *reinterpret_cast<u32x4*>(shadow + off) = so;
      }
    }
  }
  if (!dx) return;
  __syncthreads();   // every thread is done reading its transpose patch: the bytes now become wt
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u32x4 piece;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t lo = (wold[2 * q][j >> 1] >> (16 * (j & 1))) & 0xffffu;
      const uint32_t hi = (wold[2 * q + 1][j >> 1] >> (16 * (j & 1))) & 0xffffu;
      piece[q] = lo | (hi << 16);
    }
    *reinterpret_cast<u32x4*>(wt + (8 * kq + j) * FD_WLD + 8 * rg) = piece;
  }
  // ---- dx block by block: A = g rows (hi + lo) as finished fragments, staged through LDS once per workgroup (all four waves
  // multiply the same g rows); B = wt rows k.  The next block's fragments and this block's gate words are requested a block ahead. ----
  const int col = lane & 31, half = lane >> 5;
  u32x4 gdr[4], gx[2];
  auto prefetch_dx = [&](int mb) {
    const u32x4* src = gd + (size_t)min(mb, nblk - 1) * TL_GD + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) gdr[i] = src[256 * i];
  };
  auto prefetch_gate = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i, bl = id >> 4, c = id & 15, b = 32 * mb + bl;
      gx[i] = (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (gate_dx && b < m && k0 + 8 * c < k) gx[i] = *reinterpret_cast<const u32x4*>(x + (size_t)b * k + k0 + 8 * c);
    }
  };
  prefetch_dx(0);
  for (int mb = 0; mb < nblk; ++mb) {
    if (mb) __syncthreads();             // every wave is through with the previous block's fragments; its dxs rows have left
#pragma unroll
    for (int i = 0; i < 4; ++i) gfr[tid + 256 * i] = gdr[i];
    __syncthreads();                     // wt (first trip) and the fragments are complete
    prefetch_dx(mb + 1);
    prefetch_gate(mb);
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const bf16x8 hw = __builtin_bit_cast(bf16x8, gfr[st * 64 + lane]);
      const bf16x8 lw = __builtin_bit_cast(bf16x8, gfr[(8 + st) * 64 + lane]);
      const bf16x8 bw = *reinterpret_cast<const bf16x8*>(wt + (32 * wave + col) * FD_WLD + 16 * st + 8 * half);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lw, bw, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hw, bw, o, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int b = (r & 3) + 8 * (r >> 2) + 4 * half;
      dxs[b * FD_XLD + 32 * wave + col] = f32_to_bf16_bits(o[r]);
    }
    __syncthreads();                     // the dx block is assembled
#pragma unroll
    for (int i = 0; i < 2; ++i) {        // 16-byte chunks: 256 contiguous bytes per row of dx
      const int id = tid + 256 * i, bl = id >> 4, c = id & 15, b = 32 * mb + bl;
      if (b < m && k0 + 8 * c < k) {
        u32x4 ov = *reinterpret_cast<const u32x4*>(dxs + bl * FD_XLD + 8 * c);
        if (gate_dx) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ov[q] &= __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, relu_pair01(gx[i][q])) * (u16x2_t){0xffff, 0xffff});
        }
        *reinterpret_cast<u32x4*>(dx + (size_t)b * k + k0 + 8 * c) = ov;
      }
    }
  }
}

__global__ __launch_bounds__(256) void linear_bwd_db_bf16path(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                               float* __restrict__ db, int m, int n) {
  int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= n) return;
  float s = 0.f;
  for (int rr = 0; rr < m; ++rr) {
    size_t off = (size_t)rr * n + col;
    float v = dy[off];
    if (ymask && !(ymask[off] > 0.f)) v = 0.f;
    s += v;
  }
  db[col] = s;
}

// ---------------------------------------------------------------------------------------------
// v2 forward / dx for n <= 128, m <= 32 (the fc1 shapes): the [n, KC] weight panel of a k-tile is fetched with
// row-contiguous 512-byte runs (32 lanes x 16 B per row), staged in LDS once per workgroup and consumed by MFMA from
// there -- every weight byte crosses HBM once for the forward and once for dx.  The next tile's global loads are in
// flight while the current one is multiplied.
// ---------------------------------------------------------------------------------------------
#ifndef PV_V2_KC
#define PV_V2_KC 256
#endif
constexpr int V2_KC = PV_V2_KC;          // k-columns per tile
constexpr int V2_WS = V2_KC * 2 + 64;    // LDS row stride in bytes (rows 64 B apart mod 256: conflict-free 8-row reads)
constexpr int V2_CH = V2_KC / 8;         // 16-byte chunks per row
constexpr int V2_RPP = 256 / V2_CH;      // rows per pass of the 256 threads
constexpr int V2_WPASS = 128 / V2_RPP, V2_XPASS = 32 / V2_RPP;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

// rows [0, nrows) of a [*, k] bf16 matrix, columns [k0, k0 + KC): thread t owns 16-byte chunk t % CH of rows
// t / CH + RPP * i -- a row is fetched as one contiguous KC*2-byte run
template <int NPASS>
__device__ __forceinline__ void v2_load(const uint16_t* __restrict__ base, int nrows, long long k, long long k0,
                                        u32x4 (&r)[NPASS]) {
  const int chunk = threadIdx.x % V2_CH, rg = threadIdx.x / V2_CH;
  const long long col = k0 + chunk * 8;
  const bool c_ok = col < k;   // k % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int row = rg + V2_RPP * i;
    const bool ok = c_ok && row < nrows;
    const u32x4 v = *reinterpret_cast<const u32x4*>(base + (ok ? (size_t)row * k + col : 0));
    r[i] = ok ? v : (u32x4){0u, 0u, 0u, 0u};
  }
}
template <int NPASS>
__device__ __forceinline__ void v2_store(unsigned char* lds, const u32x4 (&r)[NPASS]) {
  const int chunk = threadIdx.x % V2_CH, rg = threadIdx.x / V2_CH;
#pragma unroll
  for (int i = 0; i < NPASS; ++i) *reinterpret_cast<u32x4*>(lds + (rg + V2_RPP * i) * V2_WS + chunk * 16) = r[i];
}

__global__ __launch_bounds__(256) void linear_fwd_bf16_v2_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                  float* __restrict__ partial, int m, int n, long long k,
                                                                  int n_tiles, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) unsigned char wt[128 * V2_WS];
  __shared__ __attribute__((aligned(16))) unsigned char xt[32 * V2_WS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int t0 = blockIdx.x * tiles_per_wg;
  const int t1 = t0 + tiles_per_wg < n_tiles ? t0 + tiles_per_wg : n_tiles;
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  u32x4 wr[V2_WPASS], xr[V2_XPASS];
  if (t0 < t1) {
    v2_load<V2_WPASS>(w, n, k, (long long)t0 * V2_KC, wr);
    v2_load<V2_XPASS>(x, m, k, (long long)t0 * V2_KC, xr);
  }
  const unsigned char* ap = xt + r * V2_WS + hh * 16;
  const unsigned char* bp = wt + (wave * 32 + r) * V2_WS + hh * 16;
  for (int t = t0; t < t1; ++t) {
    __syncthreads();  // the previous tile's LDS reads are done
    v2_store<V2_WPASS>(wt, wr);
    v2_store<V2_XPASS>(xt, xr);
    __syncthreads();
    if (t + 1 < t1) {
      v2_load<V2_WPASS>(w, n, k, (long long)(t + 1) * V2_KC, wr);
      v2_load<V2_XPASS>(x, m, k, (long long)(t + 1) * V2_KC, xr);
    }
    if (wave * 32 < n) {
#pragma unroll
      for (int ks = 0; ks < V2_KC / 16; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(bp + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    }
  }
  const int col = wave * 32 + r;
  if (col < n) {
    float* dst = partial + (size_t)blockIdx.x * m * n + col;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = (j & 3) + 8 * (j >> 2) + 4 * hh;
      if (row < m) dst[(size_t)row * n] = acc[j];
    }
  }
}

// dx[m][k] = sum_n g[m][n] w[n][k]: g (f32) enters the MFMA as a bf16 hi + lo pair (two MFMAs per step), so the
// gradient keeps ~16 mantissa bits; the [n][k] panel is read transposed from LDS (ds_read_b64_tr_b16).
// Wave w owns the KC/4 columns [w*KC/4, (w+1)*KC/4) of the tile (KC/128 column tiles of 32).
__global__ __launch_bounds__(256) void linear_bwd_dx_bf16_v2_kernel(const uint16_t* __restrict__ w, const float* __restrict__ dy,
                                                                     const float* __restrict__ ymask, uint16_t* __restrict__ dx,
                                                                     int m, int n, long long k, int n_tiles, int tiles_per_wg,
                                                                     float* __restrict__ db) {
  __shared__ __attribute__((aligned(16))) unsigned char wt[128 * V2_WS];
  // the bias gradient (column sums of the gated dy, rows added in index order like linear_bwd_db_bf16path) rides along
  // in workgroup 0: one launch less on a path where every launch costs ~5 us
  if (db && blockIdx.x == 0 && (int)threadIdx.x < n) {
    float sb = 0.f;
    for (int rr = 0; rr < m; ++rr) {
      const size_t off = (size_t)rr * n + threadIdx.x;
      float v = dy[off];
      if (ymask && !(ymask[off] > 0.f)) v = 0.f;
      sb += v;
    }
    db[threadIdx.x] = sb;
  }
  constexpr int CPW = V2_KC / 4;   // columns per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int nks = (n + 15) / 16;
  bf16x8 ahi[8], alo[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    s16x4 h0, h1, l0, l1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int nn = 16 * ks + 8 * hh + j;
      float v = 0.f;
      if (r < m && nn < n) {
        v = dy[(size_t)r * n + nn];
        if (ymask && !(ymask[(size_t)r * n + nn] > 0.f)) v = 0.f;
      }
      const uint16_t hb = f32_to_bf16_bits(v);
      const float hf = __builtin_bit_cast(float, (uint32_t)hb << 16);
      const uint16_t lb = f32_to_bf16_bits(v - hf);
      if (j < 4) h0[j] = (short)hb, l0[j] = (short)lb; else h1[j - 4] = (short)hb, l1[j - 4] = (short)lb;
    }
    const __attribute__((ext_vector_type(8))) short h8 = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    const __attribute__((ext_vector_type(8))) short l8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    ahi[ks] = __builtin_bit_cast(bf16x8, h8);
    alo[ks] = __builtin_bit_cast(bf16x8, l8);
  }
  // transposed-read address of this lane: 16-lane group G covers columns 16*(G&1).., rows 8*(G>>1) + 4*s + q
  const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int tr_off = (8 * (G >> 1) + q) * V2_WS + (wave * CPW + 16 * (G & 1) + 4 * p) * 2;

  const int t0 = blockIdx.x * tiles_per_wg;
  const int t1 = t0 + tiles_per_wg < n_tiles ? t0 + tiles_per_wg : n_tiles;
  u32x4 wr[V2_WPASS];
  if (t0 < t1) v2_load<V2_WPASS>(w, n, k, (long long)t0 * V2_KC, wr);
  for (int t = t0; t < t1; ++t) {
    __syncthreads();
    v2_store<V2_WPASS>(wt, wr);
    __syncthreads();
    if (t + 1 < t1) v2_load<V2_WPASS>(w, n, k, (long long)(t + 1) * V2_KC, wr);
    const long long k0 = (long long)t * V2_KC;
#pragma unroll
    for (int c = 0; c < CPW / 32; ++c) {
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < nks) {
          const unsigned char* bp = wt + tr_off + ks * 16 * V2_WS + c * 64;
          const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(bp));
          const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(bp + 4 * V2_WS));
          const __attribute__((ext_vector_type(8))) short b8s = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          const bf16x8 b8 = __builtin_bit_cast(bf16x8, b8s);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], b8, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ks], b8, acc, 0, 0, 0);
        }
      }
      const long long kcol = k0 + wave * CPW + c * 32 + r;
      if (kcol < k) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int row = (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (row < m) dx[(size_t)row * k + kcol] = f32_to_bf16_bits(acc[j]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fc1's weight gradient written once as bf16 (the data-parallel wire format, pv_linear_wgrad_bf16out) on the matrix cores.
// dW[n][k] = sum_m g[m][n] x[m][k], g = dy (.) relu'.  The register-tiled form (linear_bwd_dw_bf16_kernel<2>) multiplies on the
// vector ALU: 8.2 GFLOP at m = 32 for 0.32 GB of traffic -- 202 us where the bytes need 75 (345 us at m = 64: every rank of the
// 8-GPU run pays it per step).  Here a workgroup owns all n <= 128 rows x 128 k-columns: the x tile [m][128] (bf16) sits in LDS
// and is read transposed (ds_read_b64_tr_b16) as the B operand, g^T is the A operand as a bf16 hi + lo pair (two MFMAs per
// 16 batch rows: ~16 mantissa bits of g, as linear_bwd_dx_bf16_v2_kernel), wave w multiplies rows 32 w .. + 31; the finished
// tile goes through LDS and leaves as 256-byte runs of a row.  f32 accumulation; the result is the bf16 rounding of a sum that
// differs from the register-tiled kernel's in the last bits of f32 (tests: <= 1 bf16 ulp apart, > 99 % of the elements equal).
// ---------------------------------------------------------------------------------------------
constexpr int WB_KC = 128, WB_XS = WB_KC * 2 + 64, WB_OS = WB_KC + 8, WB_MAXM = 64;

__global__ __launch_bounds__(256, 2) void linear_wgrad_bf16out_mfma_kernel(const uint16_t* __restrict__ x, const float* __restrict__ dy,
                                                                            const float* __restrict__ ymask, uint16_t* __restrict__ dw,
                                                                            int m, int n, long long k, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[WB_MAXM * WB_XS];     // x tile [m][128] bf16, rows of 320 bytes
  __shared__ __attribute__((aligned(16))) uint16_t ot[128 * WB_OS];               // dW tile [n][128] bf16 on its way out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int mk = (m + 15) >> 4;      // k-steps of 16 batch rows
  // ---- A operand of this wave, ONCE per workgroup (g does not depend on the tile): g^T rows 32 wave .. + 31, bf16 hi + lo ------
  bf16x8 ahi[WB_MAXM / 16], alo[WB_MAXM / 16];
  const int nn = 32 * wave + r;
#pragma unroll
  for (int ks = 0; ks < WB_MAXM / 16; ++ks) {
    s16x4 h0, h1, l0, l1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 16 * ks + 8 * hh + j;
      float v = 0.f;
      if (ks < mk && row < m && nn < n) {
        v = dy[(size_t)row * n + nn];
        if (ymask && !(ymask[(size_t)row * n + nn] > 0.f)) v = 0.f;
      }
      const uint16_t hb = f32_to_bf16_bits(v);
      const float hf = __builtin_bit_cast(float, (uint32_t)hb << 16);
      const uint16_t lb = f32_to_bf16_bits(v - hf);
      if (j < 4) h0[j] = (short)hb, l0[j] = (short)lb; else h1[j - 4] = (short)hb, l1[j - 4] = (short)lb;
    }
    const __attribute__((ext_vector_type(8))) short h8 = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    const __attribute__((ext_vector_type(8))) short l8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    ahi[ks] = __builtin_bit_cast(bf16x8, h8);
    alo[ks] = __builtin_bit_cast(bf16x8, l8);
  }
  // transposed-read address of this lane (as in linear_bwd_dx_bf16_v2_kernel): 16-lane group G covers columns 16 (G & 1) ..,
  // contraction rows 8 (G >> 1) + 4 s + q
  const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int tr_off = (8 * (G >> 1) + q) * WB_XS + (16 * (G & 1) + 4 * p) * 2;
  // the x tile as 16-byte chunks: thread t holds chunks t, t + 256, ... (row = chunk / 16), requested one tile ahead
  constexpr int XCH = WB_MAXM * (WB_KC / 8) / 256;      // 4
  u32x4 xr[XCH];
  auto fetch_x = [&](int t) {
    const long long k0 = (long long)t * WB_KC;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int id = tid + 256 * i, row = id >> 4, c = id & 15;
      xr[i] = (u32x4){0u, 0u, 0u, 0u};
      if (t < n_tiles && row < m && k0 + 8 * c < k) xr[i] = *reinterpret_cast<const u32x4*>(x + (size_t)row * k + k0 + 8 * c);
    }
  };
  fetch_x(blockIdx.x);
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const long long k0 = (long long)t * WB_KC;
    __syncthreads();      // the previous tile's transposed reads and its copy-out are done
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int id = tid + 256 * i, row = id >> 4, c = id & 15;
      if (row < 16 * mk) *reinterpret_cast<u32x4*>(xs + row * WB_XS + 16 * c) = xr[i];      // (zero beyond the rows / columns that exist)
    }
    __syncthreads();
    fetch_x(t + gridDim.x);
    if (32 * wave < n) {
#pragma unroll
      for (int c = 0; c < WB_KC / 32; ++c) {
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
        for (int ks = 0; ks < WB_MAXM / 16; ++ks) {
          if (ks < mk) {      // wave-uniform
            const unsigned char* bp = xs + tr_off + ks * 16 * WB_XS + c * 64;
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(bp));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(bp + 4 * WB_XS));
            const __attribute__((ext_vector_type(8))) short b8s = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            const bf16x8 b8 = __builtin_bit_cast(bf16x8, b8s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ks], b8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], b8, acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int row = 32 * wave + (j & 3) + 8 * (j >> 2) + 4 * hh;
          ot[row * WB_OS + 32 * c + r] = f32_to_bf16_bits(acc[j]);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < 128 * (WB_KC / 8); i += 256) {          // 16-byte chunks: 256 contiguous bytes per row of dW
      const int row = i >> 4, c = i & 15;
      if (row < n && k0 + 8 * c < k)
        *reinterpret_cast<u32x4*>(dw + (size_t)row * k + k0 + 8 * c) = *reinterpret_cast<const u32x4*>(ot + row * WB_OS + 8 * c);
    }
  }
}

// dx ⊙ (x > 0) in place, 8 bf16 per thread: pv_linear_bwd_bf16(gate_dx_by_x) -- x is a ReLU output, (x > 0) the ReLU derivative of
// the layer that produced it (what the one-pass kernel's gate_dx does; the consumer then needs no gate operand of its own).  A
// pass of its own (192 MB at B = 32: ~38 us): inside the dx kernel, whose accumulator layout gives a lane one 2-byte element
// per row, the gate was 2-byte loads -- 95 -> 231 us.
__global__ __launch_bounds__(256) void gate_bf16_by_relu_kernel(uint16_t* __restrict__ dx, const uint16_t* __restrict__ x, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 d = reinterpret_cast<const u32x4*>(dx)[i];
    const u32x4 xv = reinterpret_cast<const u32x4*>(x)[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] &= __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, relu_pair01(xv[q])) * (u16x2_t){0xffff, 0xffff});
    reinterpret_cast<u32x4*>(dx)[i] = d;
  }
}

// ---------------------------------------------------------------------------------------------
// v3 forward for the fc1 shapes: the same k-tile march, but the tile ([128 weight rows + 32 x rows] x 128 k-columns,
// 40 KB) goes global -> LDS directly (buffer_load_dwordx4 ... lds) into a 3-stage ring: two tiles (80 KB per CU) are
// always in flight, no staging registers, no ds_write, one barrier per tile.  A wave instruction fills 1 KB = 4 rows x
// 256 B; the LDS image of a row is rotated by its row number (16-byte chunk c of row r sits at position (c + r) & 15)
// so the 16 rows an MFMA operand read touches fall into 16 different bank slots; the rotation is applied to the SOURCE
// chunk each lane fetches, the LDS side of the instruction being linear by lane.
// ---------------------------------------------------------------------------------------------
constexpr int V3_KC = 128;

// MB: row blocks of 32 x rows a tile carries beside the 128 weight rows (round 6: two -- 64 rows per stream over the weights, 144 KB
// of ring -- for calls with more than 32 rows: a per-GPU batch of 64, the K-sharded fc1's 256 / 512 rows)
template <int MB>
__global__ __launch_bounds__(256) void linear_fwd_bf16_v3_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                  float* __restrict__ partial, int m, int n, long long k,
                                                                  int n_tiles, int tiles_per_wg) {
  // MB = 4 (128 rows per stream over the weights: the K-sharded fc1's 256 / 512 rows): 64 KB per tile, TWO stages (one tile in
  // flight while one is multiplied: 16 MB in flight chip-wide, what the memory system's latency asks for)
  constexpr int V3_ROWS = 128 + 32 * MB, V3_TILEB = V3_ROWS * 256, LOADS = V3_ROWS / 16, V3_STAGES = MB <= 2 ? 3 : 2;
  __shared__ __attribute__((aligned(1024))) unsigned char ring[V3_STAGES * V3_TILEB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int t0 = blockIdx.x * tiles_per_wg;
  const int t1 = t0 + tiles_per_wg < n_tiles ? t0 + tiles_per_wg : n_tiles;
  if (t0 >= t1) return;
  constexpr uint32_t INVALID = 0xfffffff0u;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((size_t)n * k * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)m * k * 2), 0x00020000);
  // staging role of this lane: row (lane >> 4) of a 4-row block, position lane & 15 -> source chunk (pos - row) & 15
  const int srow = lane >> 4, spos = lane & 15;
  auto load_tile = [&](int t) {
    unsigned char* dst = ring + (t % V3_STAGES) * V3_TILEB;
    const long long k0 = (long long)t * V3_KC;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int blk = wave + 4 * i;  // 4-row block: 0..31 weight rows, 32.. x rows
      const int row = 4 * blk + srow;
      const int c = (spos - row) & 15;
      const bool c_ok = k0 + c * 8 < k;
      if (blk < 32) {
        const uint32_t off = (c_ok && row < n) ? (uint32_t)(((size_t)row * k + k0 + c * 8) * 2) : INVALID;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + blk * 1024), 16, off, 0, 0, 0);
      } else {
        const int xr = row - 128;
        const uint32_t off = (c_ok && xr < m) ? (uint32_t)(((size_t)xr * k + k0 + c * 8) * 2) : INVALID;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + blk * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  f32x16 acc[MB];
#pragma unroll
  for (int q = 0; q < MB; ++q)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
  // operand reads: x row r / weight row 32*wave + r, k-chunk 2*ks + hh, both rotated by r & 15
  int pos[V3_KC / 16];
#pragma unroll
  for (int ks = 0; ks < V3_KC / 16; ++ks) pos[ks] = ((2 * ks + hh + r) & 15) * 16;
  const int a_row = (128 + r) * 256, b_row = (wave * 32 + r) * 256;

  load_tile(t0);
  if (V3_STAGES == 3 && t0 + 1 < t1) load_tile(t0 + 1);
  for (int t = t0; t < t1; ++t) {
    // this wave issues LOADS loads per tile, in order: all but the newest tile's have landed (two stages: all have)
    if (V3_STAGES == 3 && t + 1 < t1) __builtin_amdgcn_s_waitcnt(0x0f70 | (LOADS & 15));
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();  // tile t complete for every wave; every wave is done with tile t-1 (whose stage the next request reuses)
    if (t + V3_STAGES - 1 < t1) load_tile(t + V3_STAGES - 1);
    const unsigned char* tile = ring + (t % V3_STAGES) * V3_TILEB;
    if (wave * 32 < n) {
#pragma unroll
      for (int ks = 0; ks < V3_KC / 16; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(tile + b_row + pos[ks]);
#pragma unroll
        for (int q = 0; q < MB; ++q) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile + a_row + q * (32 * 256) + pos[ks]);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
        }
      }
    }
  }
  const int col = wave * 32 + r;
  if (col < n) {
    float* dst = partial + (size_t)blockIdx.x * m * n + col;
#pragma unroll
    for (int q = 0; q < MB; ++q)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = 32 * q + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (row < m) dst[(size_t)row * n] = acc[q][j];
      }
  }
}

static int v3_split(long long k, int* tiles_per_wg, int* n_tiles) {
  const long long nt = (k + V3_KC - 1) / V3_KC;
  long long nwg = nt < 256 ? nt : 256;  // one 120 KB ring per CU
  const long long per = (nt + nwg - 1) / nwg;
  nwg = (nt + per - 1) / per;
  *tiles_per_wg = (int)per;
  *n_tiles = (int)nt;
  return (int)nwg;
}

static bool v2_shapes(int m, int n) { return m <= 32 && n <= 128 && n % 16 == 0 && n >= 32; }
static int v2_split(long long k, int* tiles_per_wg, int* n_tiles) {
  const long long nt = (k + V2_KC - 1) / V2_KC;
  long long nwg = nt < 512 ? nt : 512;  // 1-2 workgroups per CU (LDS), each with one tile in flight and one in LDS
  const long long per = (nt + nwg - 1) / nwg;
  nwg = (nt + per - 1) / per;
  *tiles_per_wg = (int)per;
  *n_tiles = (int)nt;
  return (int)nwg;
}

static int bf16_fwd_split(long long k, int* kblocks_per_wg) {
  long long total_kb = (k + 63) / 64;
  long long nwg = total_kb < 1024 ? total_kb : 1024;  // 4 workgroups per CU keep ~64 KB of loads in flight per CU
  long long per = (total_kb + nwg - 1) / nwg;
  nwg = (total_kb + per - 1) / per;
  *kblocks_per_wg = (int)per;
  return (int)nwg;
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_linear_bf16_workspace_bytes(int32_t m, int32_t n, int64_t k, size_t* bytes) {
  PV_REQUIRE(bytes && m > 0 && n > 0 && k > 0, PV_EINVAL, "pv_linear_bf16_workspace_bytes: bad arguments");
  int per;
  int nwg = bf16_fwd_split(k, &per);   // >= the v2 / v3 splits (<= 512 workgroups)
  const size_t rows = (size_t)((m + 127) / 128) * 128;   // (the LDS-staged forward keeps 32, 64 or 128 rows of slabs per launch)
  *bytes = (size_t)nwg * rows * n * sizeof(float);
  return PV_OK;
}

int pv_linear_fwd_bf16(const uint16_t* x, const uint16_t* w, const float* bias, float* y, int32_t m, int32_t n,
                       int64_t k, int relu, void* workspace, size_t workspace_bytes, void* stream) {
  PV_REQUIRE(x && w && y && workspace, PV_EINVAL, "pv_linear_fwd_bf16: null pointer");
  PV_REQUIRE(m > 0 && n > 0 && k > 0, PV_EINVAL, "pv_linear_fwd_bf16: bad sizes");
  PV_REQUIRE(k % 8 == 0, PV_ESIZE, "pv_linear_fwd_bf16: k=%lld must be a multiple of 8", (long long)k);
  PV_REQUIRE(m <= 1024, PV_ESIZE, "pv_linear_fwd_bf16: m=%d > 1024 rows per call", m);
  PV_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0), PV_EINVAL, "pv_linear_fwd_bf16: unaligned operand");
  int per;
  int nwg = bf16_fwd_split(k, &per);
  hipStream_t st = as_stream(stream);
  float* part = (float*)workspace;
  if (v2_shapes(m < 32 ? m : 32, n)) {
    // the LDS-staged kernels take 32 rows of x: more rows (a per-GPU batch of 64 in the strong-scaling run) go in blocks of 32, each
    // a pass over the weights at the memory system's rate (150 us for 64 rows where the register-tiled kernel took 292)
    int tiles, n_tiles;
    const bool fits32 = (size_t)n * k * 2 < 0xfffffff0ull;  // the v3 kernel's raw-buffer offsets are 32-bit
    const int nwg2 = !fits32 ? v2_split(k, &tiles, &n_tiles) : v3_split(k, &tiles, &n_tiles);
    // rows per launch: two row blocks beside the weight tile when there are that many, four for the tall calls
    const int rpb = !fits32 ? 32 : (m > 128 ? 128 : (m > 32 ? 64 : 32));
    const int nblk = (m + rpb - 1) / rpb;
    const size_t blk_stride = (size_t)nwg2 * rpb * n;      // floats between two launches' slabs
    PV_REQUIRE(workspace_bytes >= (size_t)nblk * blk_stride * sizeof(float), PV_ESIZE, "pv_linear_fwd_bf16: workspace too small");
    for (int m0 = 0; m0 < m; m0 += rpb) {
      const int mb = m - m0 < rpb ? m - m0 : rpb;
      const uint16_t* xb = x + (size_t)m0 * k;
      float* pb = part + (size_t)(m0 / rpb) * blk_stride;
      if (!fits32)
        hipLaunchKernelGGL(linear_fwd_bf16_v2_kernel, dim3((unsigned)nwg2), dim3(256), 0, st, xb, w, pb, mb, n, (long long)k,
                           n_tiles, tiles);
      else if (rpb == 128)
        hipLaunchKernelGGL(linear_fwd_bf16_v3_kernel<4>, dim3((unsigned)nwg2), dim3(256), 0, st, xb, w, pb, mb, n, (long long)k,
                           n_tiles, tiles);
      else if (rpb == 64)
        hipLaunchKernelGGL(linear_fwd_bf16_v3_kernel<2>, dim3((unsigned)nwg2), dim3(256), 0, st, xb, w, pb, mb, n, (long long)k,
                           n_tiles, tiles);
      else
        hipLaunchKernelGGL(linear_fwd_bf16_v3_kernel<1>, dim3((unsigned)nwg2), dim3(256), 0, st, xb, w, pb, mb, n, (long long)k,
                           n_tiles, tiles);
    }
    // one reduce for all launches (eight of ~5 us for the K-sharded fc1's 256 rows before)
    hipLaunchKernelGGL(linear_reduce_bf16path, dim3((unsigned)((rpb * n + 63) / 64), (unsigned)nblk), dim3(256), 0, st,
                       (const float*)part, bias, y, m, n, nwg2, relu ? 1 : 0, blk_stride, rpb);
    return check_launch("pv_linear_fwd_bf16");
  }
  PV_REQUIRE(m <= 128, PV_ESIZE, "pv_linear_fwd_bf16: m=%d > 128 rows per call for this shape (n=%d: the register-tiled kernel)", m, n);
  PV_REQUIRE(workspace_bytes >= (size_t)nwg * m * n * sizeof(float), PV_ESIZE, "pv_linear_fwd_bf16: workspace too small");
  dim3 grid((unsigned)nwg, (unsigned)((n + 127) / 128));
  const int mt = (m + 31) / 32;
  switch (mt) {
    case 1: hipLaunchKernelGGL(linear_fwd_bf16_kernel<1>, grid, dim3(256), 0, st, x, w, part, m, n, (long long)k, per); break;
    case 2: hipLaunchKernelGGL(linear_fwd_bf16_kernel<2>, grid, dim3(256), 0, st, x, w, part, m, n, (long long)k, per); break;
    case 3: hipLaunchKernelGGL(linear_fwd_bf16_kernel<3>, grid, dim3(256), 0, st, x, w, part, m, n, (long long)k, per); break;
    default: hipLaunchKernelGGL(linear_fwd_bf16_kernel<4>, grid, dim3(256), 0, st, x, w, part, m, n, (long long)k, per); break;
  }
  hipLaunchKernelGGL(linear_reduce_bf16path, dim3((unsigned)((m * n + 63) / 64)), dim3(256), 0, st, (const float*)part,
                     bias, y, m, n, nwg, relu ? 1 : 0);
  return check_launch("pv_linear_fwd_bf16");
}

int pv_linear_bwd_bf16(const uint16_t* x, const uint16_t* w, const float* dy, const float* y_relu_mask, uint16_t* dx,
                       float* dw, float* db, int32_t m, int32_t n, int64_t k, int32_t gate_dx_by_x, void* stream) {
  PV_REQUIRE(dy, PV_EINVAL, "pv_linear_bwd_bf16: null dy");
  PV_REQUIRE(!gate_dx_by_x || (x && dx), PV_EINVAL, "pv_linear_bwd_bf16: gate_dx_by_x needs x and dx");
  PV_REQUIRE(m > 0 && n > 0 && k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_bwd_bf16: bad sizes (k must be a multiple of 8)");
  hipStream_t st = as_stream(stream);
  unsigned kb = (unsigned)((k / 8 + 255) / 256);
  if (dx) {
    PV_REQUIRE(w, PV_EINVAL, "pv_linear_bwd_bf16: dx needs w");
    if (v2_shapes(m < 32 ? m : 32, n)) {
      int tiles, n_tiles;
      const int nwg2 = v2_split(k, &tiles, &n_tiles);
      if (m <= 32) {
        hipLaunchKernelGGL(linear_bwd_dx_bf16_v2_kernel, dim3((unsigned)nwg2), dim3(256), 0, st, w, dy, y_relu_mask, dx, m, n,
                           (long long)k, n_tiles, tiles, db);
        db = nullptr;  // done by the dx kernel
      } else {
        // more than 32 rows: blocks of 32 through the same kernel (the bias gradient, a sum over ALL rows, by its own kernel below)
        for (int m0 = 0; m0 < m; m0 += 32)
          hipLaunchKernelGGL(linear_bwd_dx_bf16_v2_kernel, dim3((unsigned)nwg2), dim3(256), 0, st, w, dy + (size_t)m0 * n,
                             y_relu_mask ? y_relu_mask + (size_t)m0 * n : (const float*)nullptr, dx + (size_t)m0 * k,
                             m - m0 < 32 ? m - m0 : 32, n, (long long)k, n_tiles, tiles, (float*)nullptr);
      }
    } else {
    size_t lds = (size_t)n * BT * sizeof(float);
    PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_bwd_bf16: n=%d too large", n);
    hipLaunchKernelGGL(linear_bwd_dx_bf16_kernel, dim3(xcd_grid(kb, (unsigned)((m + BT - 1) / BT))), dim3(256), lds, st, w, dy,
                       y_relu_mask, dx, m, n, (long long)k);
    }
    if (gate_dx_by_x) {
      PV_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dx % 16) == 0, PV_EINVAL, "pv_linear_bwd_bf16: gate_dx_by_x needs 16-byte aligned x and dx");
      const size_t n8 = (size_t)m * (size_t)k / 8;
      hipLaunchKernelGGL(gate_bf16_by_relu_kernel, dim3(stream_grid(n8, 256)), dim3(256), 0, st, dx, x, n8);
    }
  }
  if (dw) {
    PV_REQUIRE(x, PV_EINVAL, "pv_linear_bwd_bf16: dw needs x");
    size_t lds = (size_t)m * BT * sizeof(float);
    PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_bwd_bf16: m=%d too large", m);
    hipLaunchKernelGGL(linear_bwd_dw_bf16_kernel<0>, dim3(xcd_grid(kb, (unsigned)((n + BT - 1) / BT))), dim3(256), lds, st, x, dy,
                       y_relu_mask, dw, m, n, (long long)k, (float*)nullptr, (float*)nullptr, (uint16_t*)nullptr,
                       AdamScalars{});
  }
  if (db) {
    hipLaunchKernelGGL(linear_bwd_db_bf16path, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, y_relu_mask, db, m, n);
  }
  return check_launch("pv_linear_bwd_bf16");
}

int pv_linear_wgrad_adam_bf16(const uint16_t* x, const float* dy, const float* y_relu_mask, float* param, float* exp_avg,
                              float* exp_avg_sq, uint16_t* bf16_shadow, int32_t m, int32_t n, int64_t k, double lr,
                              double beta1, double beta2, double eps, int32_t step, void* stream) {
  PV_REQUIRE(x && dy && param && exp_avg && exp_avg_sq, PV_EINVAL, "pv_linear_wgrad_adam_bf16: null pointer");
  PV_REQUIRE(m > 0 && n > 0 && k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_wgrad_adam_bf16: bad sizes (k must be a multiple of 8)");
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_linear_wgrad_adam_bf16: step must be >= 1");
  size_t lds = (size_t)m * BT * sizeof(float);
  PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_wgrad_adam_bf16: m=%d too large", m);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamScalars ad{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                 (float)(-(lr / bc1))};
  unsigned kb = (unsigned)((k / 8 + 255) / 256);
  hipLaunchKernelGGL(linear_bwd_dw_bf16_kernel<1>, dim3(xcd_grid(kb, (unsigned)((n + BT - 1) / BT))), dim3(256), lds,
                     as_stream(stream), x, dy, y_relu_mask, param, m, n, (long long)k, exp_avg, exp_avg_sq, bf16_shadow, ad);
  return check_launch("pv_linear_wgrad_adam_bf16");
}

int pv_linear_wgrad_adam_f32(const float* x, const float* dy, const float* y_relu_mask, float* param, float* exp_avg,
                             float* exp_avg_sq, int32_t m, int32_t n, int64_t k, double lr, double beta1, double beta2,
                             double eps, int32_t step, void* stream) {
  PV_REQUIRE(x && dy && param && exp_avg && exp_avg_sq, PV_EINVAL, "pv_linear_wgrad_adam_f32: null pointer");
  PV_REQUIRE(m > 0 && n > 0 && k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_wgrad_adam_f32: bad sizes (k must be a multiple of 8)");
  PV_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)param % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 && ((uintptr_t)exp_avg_sq % 16) == 0,
             PV_EINVAL, "pv_linear_wgrad_adam_f32: 16-byte aligned buffers expected");
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_linear_wgrad_adam_f32: step must be >= 1");
  size_t lds = (size_t)m * BT * sizeof(float);
  PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_wgrad_adam_f32: m=%d too large", m);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamScalars ad{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                 (float)(-(lr / bc1))};
  unsigned kb = (unsigned)((k / 8 + 255) / 256);
  hipLaunchKernelGGL((linear_bwd_dw_bf16_kernel<1, float>), dim3(xcd_grid(kb, (unsigned)((n + BT - 1) / BT))), dim3(256), lds,
                     as_stream(stream), x, dy, y_relu_mask, param, m, n, (long long)k, exp_avg, exp_avg_sq, (uint16_t*)nullptr, ad);
  return check_launch("pv_linear_wgrad_adam_f32");
}

// the one-pass fc1 backward by schedule (same arithmetic, same bits).  Default (round 5): TWO rows of p / m / v in flight per thread,
// the first two requested at the top of the kernel -- 147 KB in flight per CU at three workgroups -- and the gradient tile formed in
// two halves of four rows, each right before its Adam rows.  Same-process A/Bs of the whole train step on a fast and a slow box:
// one row -> two rows in flight 1.5340 -> 1.5206 ms and 1.6858 -> 1.6722 ms (-13.5 us); gradient in halves another -4 .. -8 us
// (three runs).  PV_FC1_FETCH=early2: two rows, gradient at once; =early1: one row in flight, requested at the top (-0 .. -7 us
// against late); =late: rounds 3-4 (one row, first request after the gradient tile).  Measured and dropped: three or four rows at
// two workgroups per CU (-6 us), three rows with the gradient in quarters (no spill, +5 us), a persistent form (88 spills, +124 us).
#define PV_LAUNCH_FC1_ONE_PASS(grid, st, ...)                                                                         \
  do {                                                                                                                \
    const char* fs_ = getenv("PV_FC1_FETCH");                                                                         \
    if (fs_ && !strcmp(fs_, "late")) hipLaunchKernelGGL((linear_bwd_dw_dx_adam_kernel<1, false>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); \
    else if (fs_ && !strcmp(fs_, "early1")) hipLaunchKernelGGL((linear_bwd_dw_dx_adam_kernel<1, true>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); \
    else if (fs_ && !strcmp(fs_, "early2")) hipLaunchKernelGGL((linear_bwd_dw_dx_adam_kernel<2, true>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((linear_bwd_dw_dx_adam_kernel<2, true, 4>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);    \
  } while (0)

int pv_linear_wgrad_dx_adam_bf16(const uint16_t* x, const float* dy, const float* y_relu_mask, float* param, float* exp_avg,
                                 float* exp_avg_sq, uint16_t* bf16_shadow, uint16_t* dx, float* db, int32_t m, int32_t n,
                                 int64_t k, double lr, double beta1, double beta2, double eps, int32_t step,
                                 int32_t gate_dx_by_x, int32_t moments_tiled, void* stream) {
  PV_REQUIRE(x && dy && param && exp_avg && exp_avg_sq, PV_EINVAL, "pv_linear_wgrad_dx_adam_bf16: null pointer");
  PV_REQUIRE(m > 0 && m <= 32 && n > 0 && n <= 128 && n % 8 == 0, PV_ESIZE,
             "pv_linear_wgrad_dx_adam_bf16: built for m <= 32 rows of x and n <= 128 (multiple of 8) outputs, got m=%d n=%d", m, n);
  PV_REQUIRE(k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_wgrad_dx_adam_bf16: k must be a multiple of 8");
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_linear_wgrad_dx_adam_bf16: step must be >= 1");
  PV_REQUIRE(!moments_tiled || k % FD_KT == 0, PV_ESIZE, "pv_linear_wgrad_dx_adam_bf16: tiled moments need k %% %d == 0", FD_KT);
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamScalars ad{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                 (float)(-(lr / bc1))};
  const unsigned grid = (unsigned)((k + FD_KT - 1) / FD_KT);
  PV_LAUNCH_FC1_ONE_PASS(grid, as_stream(stream), x, dy, y_relu_mask, param, m, n,
                     (long long)k, exp_avg, exp_avg_sq, bf16_shadow, dx, db, ad, gate_dx_by_x, (const float*)nullptr,
                     moments_tiled ? 1 : 0);
  return check_launch("pv_linear_wgrad_dx_adam_bf16");
}

int pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes(int32_t m, size_t* bytes) {
  PV_REQUIRE(m > 0 && bytes, PV_EINVAL, "pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes: bad arguments");
  *bytes = (size_t)((m + 31) / 32) * (TL_GA + TL_GD) * 16;      // the output gradients as matrix-core operand fragments
  return PV_OK;
}

int pv_linear_wgrad_dx_adam_tall_bf16(const uint16_t* x, const float* dy, float* param, float* exp_avg, float* exp_avg_sq,
                                      uint16_t* bf16_shadow, uint16_t* dx, int32_t m, int32_t n, int64_t k, double lr,
                                      double beta1, double beta2, double eps, int32_t step, float grad_scale,
                                      int32_t gate_dx_by_x, int32_t moments_tiled, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  PV_REQUIRE(x && dy && param && exp_avg && exp_avg_sq && workspace, PV_EINVAL, "pv_linear_wgrad_dx_adam_tall_bf16: null pointer");
  PV_REQUIRE(!moments_tiled || k % FD_KT == 0, PV_ESIZE, "pv_linear_wgrad_dx_adam_tall_bf16: tiled moments need k %% 128 == 0");
  PV_REQUIRE(m > 0 && n > 0 && n <= 128 && n % 8 == 0, PV_ESIZE,
             "pv_linear_wgrad_dx_adam_tall_bf16: n <= 128 (multiple of 8) outputs, got m=%d n=%d", m, n);
  PV_REQUIRE(k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_wgrad_dx_adam_tall_bf16: k must be a multiple of 8");
  PV_REQUIRE(step >= 1, PV_EINVAL, "pv_linear_wgrad_dx_adam_tall_bf16: step must be >= 1");
  PV_REQUIRE((((uintptr_t)x | (uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)bf16_shadow | (uintptr_t)dx |
               (uintptr_t)workspace) & 15) == 0,
             PV_EINVAL, "pv_linear_wgrad_dx_adam_tall_bf16: buffers must be 16-byte aligned");
  const int nblk = (m + 31) / 32;
  PV_REQUIRE(workspace_bytes >= (size_t)nblk * (TL_GA + TL_GD) * 16, PV_ESIZE,
             "pv_linear_wgrad_dx_adam_tall_bf16: workspace too small (pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes)");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamScalars ad{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                 (float)(-(lr / bc1))};
  u32x4* ga = static_cast<u32x4*>(workspace);
  u32x4* gd = ga + (size_t)nblk * TL_GA;
  hipLaunchKernelGGL(tall_split_g_kernel, dim3((unsigned)((nblk * (TL_GA + TL_GD) + 255) / 256)), dim3(256), 0, as_stream(stream), dy,
                     m, n, ga, gd, nblk);
  const unsigned grid = (unsigned)((k + FD_KT - 1) / FD_KT);
  hipLaunchKernelGGL((linear_bwd_dw_dx_adam_tall_kernel<2>), dim3(grid), dim3(256), 0, as_stream(stream), x, ga, gd, param, m, n,
                     (long long)k, exp_avg, exp_avg_sq, bf16_shadow, dx, ad, gate_dx_by_x, grad_scale, moments_tiled ? 1 : 0);
  return check_launch("pv_linear_wgrad_dx_adam_tall_bf16");
}

int pv_linear_wgrad_dx_adam_dev_bf16(const uint16_t* x, const float* dy, const float* y_relu_mask, float* param, float* exp_avg,
                                     float* exp_avg_sq, uint16_t* bf16_shadow, uint16_t* dx, float* db, int32_t m, int32_t n,
                                     int64_t k, const float* adam_scalars_dev, int32_t gate_dx_by_x, int32_t moments_tiled,
                                     void* stream) {
  PV_REQUIRE(x && dy && param && exp_avg && exp_avg_sq && adam_scalars_dev, PV_EINVAL,
             "pv_linear_wgrad_dx_adam_dev_bf16: null pointer");
  PV_REQUIRE(m > 0 && m <= 32 && n > 0 && n <= 128 && n % 8 == 0, PV_ESIZE,
             "pv_linear_wgrad_dx_adam_dev_bf16: built for m <= 32 rows of x and n <= 128 (multiple of 8) outputs, got m=%d n=%d", m, n);
  PV_REQUIRE(k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_wgrad_dx_adam_dev_bf16: k must be a multiple of 8");
  PV_REQUIRE(!moments_tiled || k % FD_KT == 0, PV_ESIZE, "pv_linear_wgrad_dx_adam_dev_bf16: tiled moments need k %% %d == 0", FD_KT);
  const unsigned grid = (unsigned)((k + FD_KT - 1) / FD_KT);
  PV_LAUNCH_FC1_ONE_PASS(grid, as_stream(stream), x, dy, y_relu_mask, param, m, n,
                     (long long)k, exp_avg, exp_avg_sq, bf16_shadow, dx, db, AdamScalars{}, gate_dx_by_x, adam_scalars_dev,
                     moments_tiled ? 1 : 0);
  return check_launch("pv_linear_wgrad_dx_adam_dev_bf16");
}

int pv_linear_wgrad_bf16out(const uint16_t* x, const float* dy, const float* y_relu_mask, uint16_t* dw_bf16, int32_t m,
                            int32_t n, int64_t k, void* stream) {
  PV_REQUIRE(x && dy && dw_bf16, PV_EINVAL, "pv_linear_wgrad_bf16out: null pointer");
  PV_REQUIRE(m > 0 && n > 0 && k > 0 && k % 8 == 0, PV_ESIZE, "pv_linear_wgrad_bf16out: bad sizes (k must be a multiple of 8)");
  // (PV_WGRAD_BF16OUT_VALU: the register-tiled form for the shapes the matrix-core kernel also covers -- the tests' cross-check)
  if (m <= WB_MAXM && n <= 128 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dw_bf16 % 16) == 0 && !getenv("PV_WGRAD_BF16OUT_VALU")) {
    const long long n_tiles = (k + WB_KC - 1) / WB_KC;
    hipLaunchKernelGGL(linear_wgrad_bf16out_mfma_kernel, dim3((unsigned)std::min<long long>(n_tiles, 2 * kNumCU)), dim3(256), 0,
                       as_stream(stream), x, dy, y_relu_mask, dw_bf16, m, n, (long long)k, (int)n_tiles);
    return check_launch("pv_linear_wgrad_bf16out");
  }
  size_t lds = (size_t)m * BT * sizeof(float);
  PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_linear_wgrad_bf16out: m=%d too large", m);
  unsigned kb = (unsigned)((k / 8 + 255) / 256);
  hipLaunchKernelGGL(linear_bwd_dw_bf16_kernel<2>, dim3(xcd_grid(kb, (unsigned)((n + BT - 1) / BT))), dim3(256), lds, as_stream(stream),
                     x, dy, y_relu_mask, reinterpret_cast<float*>(dw_bf16), m, n, (long long)k, (float*)nullptr,
                     (float*)nullptr, (uint16_t*)nullptr, AdamScalars{});
  return check_launch("pv_linear_wgrad_bf16out");
}

}  // extern "C"
