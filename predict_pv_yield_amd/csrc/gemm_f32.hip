// Batched, arbitrarily strided f32 GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32
// accumulation) -- the one contraction kernel behind every dense op of the Perceiver path:
//   nn.Linear of Attention.to_q / to_kv / to_out and FeedForward, the attention products q.k^T and softmax.v
//   (perceiver_pytorch.Perceiver as instantiated by predict_pv_yield/models/perceiver/perceiver.py:70-80), and their
//   backward products (weight gradients use the split-K form).
// C[z](m,n) = sum_k A[z](m,k) * B[z](k,n)  (+ bias[n]) (ReLU optional), with
//   A[z](m,k) at a + z1*a_bs1 + z2*a_bs2 + m*a_rs + k*a_cs, B likewise, C row-major with leading dimension ldc.
// Workgroup = 4 waves = a 128 x 64 tile of C; wave w owns rows 32w..32w+31 (two 32x32 accumulators).  K advances in
// blocks of 16: the A (16 x 128) and B (16 x 64) panels are staged k-major in LDS (row strides 132 / 68 words keep the
// 32-lane operand reads and the transposing stores conflict-free); the next panels are fetched into registers while
// the current ones are multiplied.  The global->LDS mapping follows whichever stride of the operand is 1, so row-major,
// transposed and sliced (chunked k/v, per-head) operands are all read with contiguous lanes.
#include "pv_common.h"

namespace pv {

typedef float v16f_t __attribute__((ext_vector_type(16)));

constexpr int G_BM = 128, G_BN = 64, G_BK = 16;
constexpr int G_AS = G_BM + 4, G_BS = G_BN + 4;

struct GemmK {
  const float* a;
  const float* b;
  const float* bias;
  float* c;
  int m, n, k;
  long long a_rs, a_cs, b_rs, b_cs, ldc;
  int batch2, k_splits, k_chunk;
  long long a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2, c_ss;
  int relu;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmK g) {
  __shared__ float As[G_BK * G_AS];
  __shared__ float Bs[G_BK * G_BS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z / g.k_splits, split = blockIdx.z % g.k_splits;
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const float* __restrict__ A = g.a + z1 * g.a_bs1 + z2 * g.a_bs2;
  const float* __restrict__ B = g.b + z1 * g.b_bs1 + z2 * g.b_bs2;
  float* __restrict__ C = g.c + z1 * g.c_bs1 + z2 * g.c_bs2 + split * g.c_ss;
  const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;
  const int kbeg = split * g.k_chunk;
  const int kend = kbeg + g.k_chunk < g.k ? kbeg + g.k_chunk : g.k;

  // global -> register staging maps (lanes run along the unit-stride dimension of each operand)
  const bool a_k_fast = g.a_cs == 1 && g.a_rs != 1;
  const bool b_n_fast = g.b_cs == 1;
  int a_m[8], a_k[8], b_n[4], b_k[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    a_k[i] = a_k_fast ? idx % G_BK : idx / G_BM;
    a_m[i] = a_k_fast ? idx / G_BK : idx % G_BM;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    b_n[i] = b_n_fast ? idx % G_BN : idx / G_BK;
    b_k[i] = b_n_fast ? idx / G_BN : idx % G_BK;
  }
  float ar[8], br[4];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int mm = m0 + a_m[i], kk = k0 + a_k[i];
      ar[i] = (mm < g.m && kk < kend) ? A[mm * g.a_rs + kk * g.a_cs] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nn = n0 + b_n[i], kk = k0 + b_k[i];
      br[i] = (nn < g.n && kk < kend) ? B[kk * g.b_rs + nn * g.b_cs] : 0.f;
    }
  };
  v16f_t acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
  const int a_rd = (lane >> 5) * G_AS + wave * 32 + (lane & 31);
  const int b_rd = (lane >> 5) * G_BS + (lane & 31);

  if (kbeg < kend) load(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += G_BK) {
    __syncthreads();  // previous panel fully consumed
#pragma unroll
    for (int i = 0; i < 8; ++i) As[a_k[i] * G_AS + a_m[i]] = ar[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) Bs[b_k[i] * G_BS + b_n[i]] = br[i];
    __syncthreads();
    if (k0 + G_BK < kend) load(k0 + G_BK);
#pragma unroll
    for (int kk = 0; kk < G_BK; kk += 2) {
      const float av = As[kk * G_AS + a_rd];
      const float b0 = Bs[kk * G_BS + b_rd];
      const float b1 = Bs[kk * G_BS + b_rd + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
    }
  }
  // C layout of the 32x32 accumulator: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = lane & 31;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int nn = n0 + 32 * t + col;
    if (nn >= g.n) continue;
    const float bv = g.bias ? g.bias[nn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (mm < g.m) {
        float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
        if (g.relu) v = v > 0.f ? v : 0.f;
        C[(long long)mm * g.ldc + nn] = v;
      }
    }
  }
}

// out[i] = sum over s of slabs[s * stride + offset + i], i < n.  Workgroup = 32 columns x 8 slab groups: thread (c, g)
// adds slabs g, g+8, ... in index order, the 8 group sums are then added in group order -- a fixed order, 8-way parallel.
__global__ __launch_bounds__(256) void sum_slabs_f32(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                                     int n_slabs, long long stride, long long offset) {
  __shared__ float red[8][32];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x * 32 + c;
  float s = 0.f;
  if (i < n)
    for (int k = grp; k < n_slabs; k += 8) s += slabs[(size_t)k * stride + offset + i];
  red[grp][c] = s;
  __syncthreads();
  if (grp == 0 && i < n)
    out[i] = ((((((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]) + red[4][c]) + red[5][c]) + red[6][c]) + red[7][c];
}

void launch_sum_slabs(const float* slabs, float* out, long long n, int n_slabs, long long stride, long long offset,
                      hipStream_t st) {
  hipLaunchKernelGGL(sum_slabs_f32, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, slabs, out, n, n_slabs, stride, offset);
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_gemm_f32(const float* a, const float* b, const float* bias, float* c, const pv_gemm_desc* d, int relu, void* stream) {
  PV_REQUIRE(a && b && c && d, PV_EINVAL, "pv_gemm_f32: null pointer");
  PV_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0, PV_EINVAL, "pv_gemm_f32: non-positive extent (%d,%d,%d)", d->m, d->n, d->k);
  PV_REQUIRE(d->batch1 > 0 && d->batch2 > 0 && d->k_splits > 0, PV_EINVAL, "pv_gemm_f32: batch counts and k_splits must be >= 1");
  PV_REQUIRE(d->ldc >= d->n, PV_EINVAL, "pv_gemm_f32: ldc < n");
  PV_REQUIRE(!(d->k_splits > 1 && (bias || relu)), PV_EINVAL, "pv_gemm_f32: bias / ReLU cannot be applied to split-K partial products");
  const long long zs = (long long)d->batch1 * d->batch2 * d->k_splits;
  PV_REQUIRE(zs <= 65535, PV_ESIZE, "pv_gemm_f32: batch1*batch2*k_splits = %lld exceeds the grid limit", zs);
  GemmK g;
  g.a = a, g.b = b, g.bias = bias, g.c = c;
  g.m = d->m, g.n = d->n, g.k = d->k;
  g.a_rs = d->a_rs, g.a_cs = d->a_cs, g.b_rs = d->b_rs, g.b_cs = d->b_cs, g.ldc = d->ldc;
  g.batch2 = d->batch2, g.k_splits = d->k_splits;
  g.k_chunk = ((d->k + d->k_splits - 1) / d->k_splits + G_BK - 1) / G_BK * G_BK;
  g.a_bs1 = d->a_bs1, g.a_bs2 = d->a_bs2, g.b_bs1 = d->b_bs1, g.b_bs2 = d->b_bs2, g.c_bs1 = d->c_bs1, g.c_bs2 = d->c_bs2;
  g.c_ss = d->c_ss;
  g.relu = relu ? 1 : 0;
  dim3 grid((unsigned)((d->n + G_BN - 1) / G_BN), (unsigned)((d->m + G_BM - 1) / G_BM), (unsigned)zs);
  PV_REQUIRE(grid.y <= 65535, PV_ESIZE, "pv_gemm_f32: m too large for one launch");
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, as_stream(stream), g);
  return check_launch("pv_gemm_f32");
}

int pv_sum_slabs_f32(const float* slabs, float* out, int64_t n, int32_t n_slabs, void* stream) {
  PV_REQUIRE(slabs && out && n > 0 && n_slabs > 0, PV_EINVAL, "pv_sum_slabs_f32: bad arguments");
  launch_sum_slabs(slabs, out, (long long)n, n_slabs, (long long)n, 0, as_stream(stream));
  return check_launch("pv_sum_slabs_f32");
}

}  // extern "C"
