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
  const float* res;      // optional residual added in the epilogue: C = A B + bias + res (2-D products only)
  long long ldr;
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
  const int m0 = blockIdx.x * G_BM, n0 = blockIdx.y * G_BN;   // row tiles along grid.x (2^31 - 1 of them), column tiles along y
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
        if (g.res) v += g.res[(long long)mm * g.ldr + nn];
        if (g.relu) v = v > 0.f ? v : 0.f;
        C[(long long)mm * g.ldc + nn] = v;
      }
    }
  }
}

// ---- the same GEMM on the bf16 matrix cores at f32 accuracy ------------------------------------------------------------
// Every f32 operand is split by truncation into three bf16 terms x = h + m + l (8 + 8 + 8 mantissa bits, both
// subtractions exact) and a product keeps the six partial products down to 2^-16 (mm, lh, hl, mh, hm, hh: what is
// dropped is <= 2^-23 relative, one f32 rounding; farneback.hip's window blur uses the same scheme).
// v_mfma_f32_32x32x16_bf16 contracts 16 elements in 8 passes where v_mfma_f32_32x32x2_f32 needs 8 x 16 passes, so the six
// products cost 3/8 of the f32 instruction's matrix-pipe time -- and these GEMMs (19 456 rows = 152 workgroups, one
// wave per SIMD, K up to 1 024) are bound by exactly that.  The split happens ONCE per element, on the way into LDS.
// LDS image of an operand panel (32 deep), per bf16 plane, follows the operand's unit stride in memory so that the
// staging stores are packed pairs either way:
//   KC (k contiguous in memory):  [row][k], 80-byte rows   -> an MFMA operand (8 consecutive k) is one ds_read_b128
//   MC (m / n contiguous):        [k][row], 320 / 192-byte rows -> two ds_read_b64_tr_b16 (the transposing LDS read)
constexpr int X3_BK = 32;
constexpr int X3_KC_RS = 80;                      // bytes per row of a KC plane (16 rows x 16 B cover the 64 banks)
constexpr int X3_A_MC_RS = 320, X3_B_MC_RS = 192; // bytes per k-row of an MC plane (4 rows x 64 B cover the 64 banks)
constexpr int X3_A_PLANE = 10240, X3_B_PLANE = 6144;

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_g;

__device__ __forceinline__ void x3_split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  const uint32_t a = __builtin_bit_cast(uint32_t, x0), b = __builtin_bit_cast(uint32_t, x1);
  h = __builtin_amdgcn_perm(b, a, 0x07060302u);                 // (b & 0xffff0000) | (a >> 16)
  const float ra = x0 - __builtin_bit_cast(float, a & 0xffff0000u);
  const float rb = x1 - __builtin_bit_cast(float, b & 0xffff0000u);
  const uint32_t a1 = __builtin_bit_cast(uint32_t, ra), b1 = __builtin_bit_cast(uint32_t, rb);
  m = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
  const float sa = ra - __builtin_bit_cast(float, a1 & 0xffff0000u);
  const float sb = rb - __builtin_bit_cast(float, b1 & 0xffff0000u);
  l = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sb), __builtin_bit_cast(uint32_t, sa), 0x07060302u);
}

// TERMS = 3: the f32-accurate form above.  TERMS = 1: every operand rounded ONCE to bf16 (nearest even), one product, f32
// accumulation -- torch.autocast's linear / matmul (the reference trains its Perceiver under Lightning precision=16:
// experiments/003_perceiver_processes_single_sat_image_then_rnn.py:40,288-294); a sixth of the matrix work, a third of the LDS
// traffic
// A_BF16 (TERMS = 1 only): A is a bf16 matrix in memory (g.a reinterpreted, strides in elements): the gradient of a
// cross-attention's keys / values stored as bf16 by the attention backward -- the values this kernel would round A to anyway
template <bool A_KC, bool B_KC, int TERMS = 3, bool A_BF16 = false>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmK g) {
  static_assert(!A_BF16 || TERMS == 1, "a bf16 A operand is for the one-term form");
  __shared__ __attribute__((aligned(16))) unsigned char As[3 * X3_A_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * X3_B_PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z / g.k_splits, split = blockIdx.z % g.k_splits;
  const int z1 = z / g.batch2, z2 = z % g.batch2;
  const float* __restrict__ A = g.a + (A_BF16 ? 0 : z1 * g.a_bs1 + z2 * g.a_bs2);      // (A_BF16: A16 below is the operand)
  const uint16_t* __restrict__ A16 = reinterpret_cast<const uint16_t*>(g.a) + z1 * g.a_bs1 + z2 * g.a_bs2;
  const float* __restrict__ B = g.b + z1 * g.b_bs1 + z2 * g.b_bs2;
  float* __restrict__ C = g.c + z1 * g.c_bs1 + z2 * g.c_bs2 + split * g.c_ss;
  const int m0 = blockIdx.x * G_BM, n0 = blockIdx.y * G_BN;   // row tiles along grid.x (2^31 - 1 of them), column tiles along y
  const int kbeg = split * g.k_chunk;
  const int kend = kbeg + g.k_chunk < g.k ? kbeg + g.k_chunk : g.k;

  // staging: a thread moves QUADS of four elements adjacent along the operand's unit stride (4 quads of A, 2 of B per
  // 32-deep panel).  Inside the matrix (whole tile, whole panel) with a unit stride, the other stride a multiple of 4 and a
  // 16-byte aligned base a quad is ONE dwordx4 load from a precomputed 64-bit offset; the earlier form (pairs, two
  // predicated dword loads each, a 64-bit multiply per address) spent ~3 000 cycles of vector ALU per panel on addresses
  // and bounds against 768 cycles of matrix work -- the K = 19 456 weight-gradient products ran at 1.5 us per panel and
  // workgroup whatever the split.  Edge tiles / ragged panels / odd strides keep per-element predicated loads.
  int a_m[4], a_k[4], a_w[4], b_n[2], b_k[2], b_w[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    if (A_KC) { a_k[i] = 4 * (idx % 8); a_m[i] = idx / 8; a_w[i] = a_m[i] * X3_KC_RS + a_k[i] * 2; }
    else      { a_m[i] = 4 * (idx % 32); a_k[i] = idx / 32; a_w[i] = a_k[i] * X3_A_MC_RS + a_m[i] * 2; }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    if (B_KC) { b_k[i] = 4 * (idx % 8); b_n[i] = idx / 8; b_w[i] = b_n[i] * X3_KC_RS + b_k[i] * 2; }
    else      { b_n[i] = 4 * (idx % 16); b_k[i] = idx / 16; b_w[i] = b_k[i] * X3_B_MC_RS + b_n[i] * 2; }
  }
  const long long a_us = A_KC ? g.a_cs : g.a_rs, a_os = A_KC ? g.a_rs : g.a_cs;   // unit / other stride of A
  const long long b_us = B_KC ? g.b_rs : g.b_cs, b_os = B_KC ? g.b_cs : g.b_rs;
  const bool a_fast = a_us == 1 && (a_os & 3) == 0 && ((uintptr_t)(A_BF16 ? (const void*)A16 : (const void*)A) & 15) == 0 && m0 + G_BM <= g.m;
  const bool b_fast = b_us == 1 && (b_os & 3) == 0 && ((uintptr_t)B & 15) == 0 && n0 + G_BN <= g.n;
  long long a_off[4], b_off[2];   // element offset of the quad at k = 0
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = (long long)(m0 + a_m[i]) * g.a_rs + (long long)a_k[i] * g.a_cs;
#pragma unroll
  for (int i = 0; i < 2; ++i) b_off[i] = (long long)(n0 + b_n[i]) * g.b_cs + (long long)b_k[i] * g.b_rs;
  f32x4 ar[4], br[2];
  u32x2 ar16[4];            // A_BF16, whole quads: the four bf16 values as they are
  bool a_raw = false;       // (uniform) the panel in ar16 rather than in ar
  auto load = [&](int k0) {
    const bool full_k = k0 + X3_BK <= kend;
    a_raw = A_BF16 && a_fast && full_k;
    if (a_raw) {
      const uint16_t* ap = A16 + (long long)k0 * g.a_cs;
#pragma unroll
      for (int i = 0; i < 4; ++i) ar16[i] = *reinterpret_cast<const u32x2*>(ap + a_off[i]);
    } else if (a_fast && full_k) {
      const float* ap = A + (long long)k0 * g.a_cs;
#pragma unroll
      for (int i = 0; i < 4; ++i) ar[i] = *reinterpret_cast<const f32x4*>(ap + a_off[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int mm = m0 + a_m[i] + (A_KC ? 0 : e), kk = k0 + a_k[i] + (A_KC ? e : 0);
          const long long off = (long long)mm * g.a_rs + (long long)kk * g.a_cs;
          ar[i][e] = (mm < g.m && kk < kend) ? (A_BF16 ? bf16_bits_to_f32(A16[off]) : A[off]) : 0.f;
        }
    }
    if (b_fast && full_k) {
      const float* bp = B + (long long)k0 * g.b_rs;
#pragma unroll
      for (int i = 0; i < 2; ++i) br[i] = *reinterpret_cast<const f32x4*>(bp + b_off[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int nn = n0 + b_n[i] + (B_KC ? 0 : e), kk = k0 + b_k[i] + (B_KC ? e : 0);
          br[i][e] = (nn < g.n && kk < kend) ? B[(long long)kk * g.b_rs + (long long)nn * g.b_cs] : 0.f;
        }
    }
  };
  v16f_t acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;

  // operand read offsets (bytes inside a plane) for K-step ks = 0 of the panel; ks = 1 adds 32 B (KC) / 16 rows (MC)
  const int row = lane & 31, half = lane >> 5;
  const int grp = lane >> 4, qi = (lane & 15) >> 2, pi = lane & 3, cb = 16 * (grp & 1);
  int a_rd[2], b_rd[2][2];
  if (A_KC) {
    a_rd[0] = (wave * 32 + row) * X3_KC_RS + 16 * half;
    a_rd[1] = 0;
  } else {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) a_rd[s2] = (8 * half + 4 * s2 + qi) * X3_A_MC_RS + (wave * 32 + cb + 4 * pi) * 2;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (B_KC) {
      b_rd[t][0] = (32 * t + row) * X3_KC_RS + 16 * half;
      b_rd[t][1] = 0;
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) b_rd[t][s2] = (8 * half + 4 * s2 + qi) * X3_B_MC_RS + (32 * t + cb + 4 * pi) * 2;
    }
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8g;
  auto fetch = [&](const unsigned char* plane, const int (&rd)[2], bool kc, int ks, int mc_rs) -> bf16x8 {
    if (kc) return *reinterpret_cast<const bf16x8*>(plane + rd[0] + 32 * ks);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_g*)(plane + rd[0] + 16 * ks * mc_rs));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_g*)(plane + rd[1] + 16 * ks * mc_rs));
    const s16x8g v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  if (kbeg < kend) load(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += X3_BK) {
    __syncthreads();  // previous panel fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (TERMS == 1) {
        *reinterpret_cast<u32x2*>(As + a_w[i]) =
            a_raw ? ar16[i] : (u32x2){pack_bf16_pair(ar[i][0], ar[i][1]), pack_bf16_pair(ar[i][2], ar[i][3])};
      } else {
        uint32_t h0, m0_, l0, h1, m1_, l1;
        x3_split_pair(ar[i][0], ar[i][1], h0, m0_, l0);
        x3_split_pair(ar[i][2], ar[i][3], h1, m1_, l1);
        const u32x2 h = {h0, h1}, m = {m0_, m1_}, l = {l0, l1};
        *reinterpret_cast<u32x2*>(As + a_w[i]) = h;
        *reinterpret_cast<u32x2*>(As + X3_A_PLANE + a_w[i]) = m;
        *reinterpret_cast<u32x2*>(As + 2 * X3_A_PLANE + a_w[i]) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (TERMS == 1) {
        *reinterpret_cast<u32x2*>(Bs + b_w[i]) = (u32x2){pack_bf16_pair(br[i][0], br[i][1]), pack_bf16_pair(br[i][2], br[i][3])};
      } else {
        uint32_t h0, m0_, l0, h1, m1_, l1;
        x3_split_pair(br[i][0], br[i][1], h0, m0_, l0);
        x3_split_pair(br[i][2], br[i][3], h1, m1_, l1);
        const u32x2 h = {h0, h1}, m = {m0_, m1_}, l = {l0, l1};
        *reinterpret_cast<u32x2*>(Bs + b_w[i]) = h;
        *reinterpret_cast<u32x2*>(Bs + X3_B_PLANE + b_w[i]) = m;
        *reinterpret_cast<u32x2*>(Bs + 2 * X3_B_PLANE + b_w[i]) = l;
      }
    }
    __syncthreads();
    if (k0 + X3_BK < kend) load(k0 + X3_BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (TERMS == 1) {
        const bf16x8 a1 = fetch(As, a_rd, A_KC, ks, X3_A_MC_RS);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, fetch(Bs, b_rd[0], B_KC, ks, X3_B_MC_RS), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, fetch(Bs, b_rd[1], B_KC, ks, X3_B_MC_RS), acc1, 0, 0, 0);
        continue;
      }
      bf16x8 a3[3], b3[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a3[p] = fetch(As + p * X3_A_PLANE, a_rd, A_KC, ks, X3_A_MC_RS);
        b3[0][p] = fetch(Bs + p * X3_B_PLANE, b_rd[0], B_KC, ks, X3_B_MC_RS);
        b3[1][p] = fetch(Bs + p * X3_B_PLANE, b_rd[1], B_KC, ks, X3_B_MC_RS);
      }
      // smallest partial products first: mm, lh, hl, mh, hm, hh
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[0][1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[0][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[1][0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1][2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[0][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1][0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0][1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1][0], acc1, 0, 0, 0);
    }
  }
  // C layout of the 32x32 accumulator: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  One 64-bit address
  // per column half and lane, wave-uniform row offsets; only a ragged last row tile tests rows.
  const int col = lane & 31;
  const int m_lane = m0 + wave * 32 + 4 * (lane >> 5);
  const bool whole = m0 + G_BM <= g.m;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int nn = n0 + 32 * t + col;
    if (nn >= g.n) continue;
    const float bv = g.bias ? g.bias[nn] : 0.f;
    float* cp = C + (long long)m_lane * g.ldc + nn;
    const float* rp = g.res ? g.res + (long long)m_lane * g.ldr + nn : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      if (whole || m_lane + dr < g.m) {
        float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
        if (rp) v += rp[(long long)dr * g.ldr];
        if (g.relu) v = v > 0.f ? v : 0.f;
        cp[(long long)dr * g.ldc] = v;
      }
    }
  }
}

// ---- "rows" form: C[M, N] = A[M, K] B[K, N] for a tall row-major A with a short K (<= 64) ------------------------------
// The Perceiver's linears over 19 456 ... 2.5 M rows with 38 ... 128 input features.  The tiled kernel above walks such a
// product as thousands of workgroups that each load one or two panels, synchronise, multiply for a microsecond and store:
// a serial latency chain per workgroup (8-57 TFLOP/s, 2 TB/s on the memory-bound ones).  Here
//   * the weights' [K x 64] column block is split and laid into LDS ONCE per (persistent) workgroup;
//   * a wave owns blocks of 32 rows and fetches their A operand STRAIGHT INTO REGISTERS in MFMA layout (lane = (row, k half):
//     8 consecutive k of its own row, 32 bytes) -- no LDS image of A, no barrier anywhere in the loop; the next block's
//     rows are in flight while the current block is split, multiplied and stored.
// OUT_BF16: C is a bf16 matrix (round to nearest even in the store): the key / value projection of a cross-attention whose
// bf16-operand kernels would round those values anyway -- the 1.3 GB of f32 K / V of experiments/003's context (2.5 M rows x 128)
// is what bounds both this kernel's store and the attention kernels' reads
// LN_A: the rows of A pass through a LayerNorm on their way into the product -- A = the un-normalised context of a
// cross-attention, the product its key / value projection (pv_context_fwd_bf16).  The two lanes that share a row exchange their
// halves, each forms mean and rstd with the serial sums of layernorm_fwd_rows_f32 (same order: same bits) and normalises the
// values it holds with that kernel's expression; the column block 0 workgroups also store mean / rstd.  The normalised context is
// never written: LayerNorm kernel (x read, ctx written) + projection (ctx read) become one read of x.
struct GemmLnA {
  const float* w;
  const float* b;
  float* mean;
  float* rstd;
  float eps;
  // optional second source of the rows: row r = [ a[r][0 .. d1) | x2[r % period][0 .. K - d1) ] -- the images' channels followed by
  // the Fourier features of the pixel's position, which every image of the batch shares (the concatenation is never written)
  const float* x2;
  int d1;
  int period;
};
template <int KSTEPS, int VEC, bool OUT_BF16 = false, int TERMS = 3, bool LN_A = false>      // TERMS as in gemm_bf16x3_kernel
__global__ __launch_bounds__(256, 2) void gemm_rows_x3_kernel(GemmK g, int n_rowblocks, GemmLnA ln = GemmLnA{}) {
  constexpr int KP = 16 * KSTEPS;
  constexpr int BRS = 2 * KP + 16;                       // bytes per n-row of a B plane (16 rows x 16 B cover the 64 banks)
  constexpr int BPLANE = G_BN * BRS;
  __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * BPLANE];
  constexpr int CW_RS = 2 * G_BN + 16;                   // bytes per row of a wave's bf16 output tile (OUT_BF16)
  __shared__ __attribute__((aligned(16))) unsigned char Cw[OUT_BF16 ? 4 * 32 * CW_RS : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * G_BN;
  const float* __restrict__ A = g.a;
  const float* __restrict__ B = g.b;
  float* __restrict__ C = g.c;
  const int row = lane & 31, half = lane >> 5;
  const int b_rd0 = row * BRS + 16 * half, b_rd1 = (32 + row) * BRS + 16 * half;
  const float bias0 = (g.bias && n0 + row < g.n) ? g.bias[n0 + row] : 0.f;
  const float bias1 = (g.bias && n0 + 32 + row < g.n) ? g.bias[n0 + 32 + row] : 0.f;
  const int stride = gridDim.y * 4;
  float araw[2][KSTEPS][8];
  auto load_a = [&](int rb, float (&dst)[KSTEPS][8]) {
    const long long mm = (long long)rb * 32 + row;
    const bool row_ok = rb < n_rowblocks && mm < g.m;
    const float* ap = A + (row_ok ? mm : 0) * g.a_rs;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k0 = 16 * ks + 8 * half;
      if constexpr (VEC == 4) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row_ok && k0 + 4 * v < g.k) t = *reinterpret_cast<const float4*>(ap + k0 + 4 * v);   // K % 4 == 0
          dst[ks][4 * v] = t.x, dst[ks][4 * v + 1] = t.y, dst[ks][4 * v + 2] = t.z, dst[ks][4 * v + 3] = t.w;
        }
      } else if constexpr (VEC == 2) {
        if (LN_A && ln.x2) {      // two sources (d1 and K even: a pair never straddles them)
          const float* a1 = A + (row_ok ? mm : 0) * ln.d1;
          const float* a2 = ln.x2 + (size_t)((unsigned)(row_ok ? mm : 0) % (unsigned)ln.period) * (g.k - ln.d1) - ln.d1;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int k = k0 + 2 * v;
            float2 t = make_float2(0.f, 0.f);
            if (row_ok && k < g.k) t = *reinterpret_cast<const float2*>((k < ln.d1 ? a1 : a2) + k);
            dst[ks][2 * v] = t.x, dst[ks][2 * v + 1] = t.y;
          }
          continue;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float2 t = make_float2(0.f, 0.f);
          if (row_ok && k0 + 2 * v < g.k) t = *reinterpret_cast<const float2*>(ap + k0 + 2 * v);   // K % 2 == 0
          dst[ks][2 * v] = t.x, dst[ks][2 * v + 1] = t.y;
        }
      } else {
#pragma unroll
        for (int v = 0; v < 8; ++v) dst[ks][v] = (row_ok && k0 + v < g.k) ? ap[k0 + v] : 0.f;
      }
    }
  };
  float lnw_r[LN_A ? KSTEPS : 1][8], lnb_r[LN_A ? KSTEPS : 1][8];      // gamma / beta of the k slots this lane holds
  if constexpr (LN_A) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const int k = 16 * ks + 8 * half + v;
        lnw_r[ks][v] = k < g.k ? ln.w[k] : 0.f;
        lnb_r[ks][v] = k < g.k ? ln.b[k] : 0.f;
      }
  }
  auto process = [&](int rb, const float (&ar_in)[KSTEPS][8]) {
    float ar[KSTEPS][8];
    if constexpr (LN_A) {
      // element k = 16 ks + 8 h + v of the row: this lane's value (h == half) or its partner's
      float lo[KSTEPS][8], hi[KSTEPS][8];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const float other = __shfl_xor(ar_in[ks][v], 32);
          lo[ks][v] = half ? other : ar_in[ks][v];
          hi[ks][v] = half ? ar_in[ks][v] : other;
        }
      auto at = [&](int k) { return ((k >> 3) & 1) ? hi[k >> 4][k & 7] : lo[k >> 4][k & 7]; };
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8 * KSTEPS; ++j) s += at(2 * j) + at(2 * j + 1);      // (slots >= K hold zeros, as the LayerNorm kernel's do)
      const float mu = s / (float)g.k;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8 * KSTEPS; ++j)
        if (2 * j < g.k) {
          const float t0 = at(2 * j) - mu, t1 = at(2 * j + 1) - mu;
          q += t0 * t0 + t1 * t1;
        }
      const float rs = 1.0f / sqrtf(q / (float)g.k + ln.eps);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int v = 0; v < 8; ++v)
          ar[ks][v] = 16 * ks + 8 * half + v < g.k ? (ar_in[ks][v] - mu) * rs * lnw_r[ks][v] + lnb_r[ks][v] : 0.f;
      const long long mm = (long long)rb * 32 + row;
      if (blockIdx.x == 0 && half == 0 && rb < n_rowblocks && mm < g.m) ln.mean[mm] = mu, ln.rstd[mm] = rs;
    } else {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int v = 0; v < 8; ++v) ar[ks][v] = ar_in[ks][v];
    }
    v16f_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if constexpr (TERMS == 1) {
        u32x4 w1;
#pragma unroll
        for (int j = 0; j < 4; ++j) w1[j] = pack_bf16_pair(ar[ks][2 * j], ar[ks][2 * j + 1]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, w1);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, *reinterpret_cast<const bf16x8*>(Bs + b_rd0 + 32 * ks), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, *reinterpret_cast<const bf16x8*>(Bs + b_rd1 + 32 * ks), acc1, 0, 0, 0);
        continue;
      }
      u32x4 hw, mw, lw;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t h_, m_, l_;
        x3_split_pair(ar[ks][2 * j], ar[ks][2 * j + 1], h_, m_, l_);
        hw[j] = h_, mw[j] = m_, lw[j] = l_;
      }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, hw), am = __builtin_bit_cast(bf16x8, mw), al = __builtin_bit_cast(bf16x8, lw);
      const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(Bs + b_rd0 + 32 * ks);
      const bf16x8 b0m = *reinterpret_cast<const bf16x8*>(Bs + BPLANE + b_rd0 + 32 * ks);
      const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(Bs + 2 * BPLANE + b_rd0 + 32 * ks);
      const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(Bs + b_rd1 + 32 * ks);
      const bf16x8 b1m = *reinterpret_cast<const bf16x8*>(Bs + BPLANE + b_rd1 + 32 * ks);
      const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(Bs + 2 * BPLANE + b_rd1 + 32 * ks);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0m, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1m, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1h, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0m, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1m, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc1, 0, 0, 0);
    }
    // C layout of the 32x32 accumulator: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  One 64-bit
    // address per (block, column half) and lane; the row offsets are wave-uniform.  Only a ragged last block tests rows.
    const bool whole = (long long)rb * 32 + 32 <= g.m;
    if constexpr (OUT_BF16) {
      // bf16 output of a whole 64-column block with rows of whole 16-byte pieces: the wave's 32 x 64 tile goes through its own
      // LDS tile and leaves as 128 contiguous bytes per row (16 bytes per lane, four store instructions) instead of 32
      // instructions of 2-byte stores that each touch 2 x 32 cache lines -- the store side was what bounded the key / value
      // projection of a 2.5 M-row context (0.64 GB written)
      if (n0 + G_BN <= g.n && !g.res && (g.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        unsigned char* ct = Cw + wave * (32 * CW_RS);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float bv = t == 0 ? bias0 : bias1;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
            if (g.relu) v = v > 0.f ? v : 0.f;
            *reinterpret_cast<uint16_t*>(ct + ((r & 3) + 8 * (r >> 2) + 4 * half) * CW_RS + 2 * (32 * t + row)) = f32_to_bf16_bits(v);
          }
        }
        // (the same wave reads what it wrote: LDS operations of a wave complete in order)
        uint16_t* cb = reinterpret_cast<uint16_t*>(C) + (long long)rb * 32 * g.ldc + n0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rr = 8 * j + (lane >> 3), c16 = lane & 7;
          if (whole || (long long)rb * 32 + rr < g.m)
            *reinterpret_cast<u32x4*>(cb + (long long)rr * g.ldc + 8 * c16) = *reinterpret_cast<const u32x4*>(ct + rr * CW_RS + 16 * c16);
        }
        return;
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int nn = n0 + 32 * t + row;
      if (nn >= g.n) continue;
      const float bv = t == 0 ? bias0 : bias1;
      const long long m_lane = (long long)rb * 32 + 4 * half;
      float* cp = C + m_lane * g.ldc + nn;
      const float* rp = g.res ? g.res + m_lane * g.ldr + nn : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (whole || m_lane + dr < g.m) {
          float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
          if (rp) v += rp[(long long)dr * g.ldr];
          if (g.relu) v = v > 0.f ? v : 0.f;
          if constexpr (OUT_BF16) reinterpret_cast<uint16_t*>(C)[(m_lane + dr) * g.ldc + nn] = f32_to_bf16_bits(v);
          else cp[(long long)dr * g.ldc] = v;
        }
      }
    }
  };
  int rb = blockIdx.y * 4 + wave;
  load_a(rb, araw[0]);      // in flight under the weight staging below
  // ---- the weights' column block: pairs (k, k+1) of column n -> three packed bf16 words -------------------------------
  // k-contiguous weights (B = W^T of an nn.Linear: b_rs == 1) with K % 4 == 0 and a 16-byte aligned base move as quads:
  // one dwordx4 load, two splits, three 8-byte LDS stores, all of a thread's loads in flight at once -- the per-pair form
  // (two predicated dword loads behind 64-bit multiplies) made this prologue ~4 us of a workgroup that then multiplies
  // for about as long
  const bool b_quads = g.b_rs == 1 && (g.k & 3) == 0 && (g.b_cs & 3) == 0 && ((uintptr_t)B & 15) == 0 && n0 + G_BN <= g.n;
  if (b_quads) {
    const int kq = g.k >> 2;                               // quads per column
    for (int idx = tid; idx < G_BN * (KP / 4); idx += 256) {
      const int n = idx / (KP / 4), q = idx - n * (KP / 4);
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (q < kq) x = *reinterpret_cast<const f32x4*>(B + (long long)(n0 + n) * g.b_cs + 4 * q);
      uint32_t h0, m0_, l0, h1, m1_, l1;
      x3_split_pair(x[0], x[1], h0, m0_, l0);
      x3_split_pair(x[2], x[3], h1, m1_, l1);
      if constexpr (TERMS == 1) h0 = pack_bf16_pair(x[0], x[1]), h1 = pack_bf16_pair(x[2], x[3]);      // (plane 0 rounded, not truncated)
      const int off = n * BRS + q * 8;
      *reinterpret_cast<u32x2*>(Bs + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(Bs + BPLANE + off) = u32x2{m0_, m1_};
      *reinterpret_cast<u32x2*>(Bs + 2 * BPLANE + off) = u32x2{l0, l1};
    }
  } else {
    for (int idx = tid; idx < G_BN * (KP / 2); idx += 256) {
      const int n = idx / (KP / 2), kp = idx - n * (KP / 2);
      const int nn = n0 + n, k = 2 * kp;
      const float x0 = (nn < g.n && k < g.k) ? B[(long long)k * g.b_rs + (long long)nn * g.b_cs] : 0.f;
      const float x1 = (nn < g.n && k + 1 < g.k) ? B[(long long)(k + 1) * g.b_rs + (long long)nn * g.b_cs] : 0.f;
      uint32_t h, m, l;
      x3_split_pair(x0, x1, h, m, l);
      if constexpr (TERMS == 1) h = pack_bf16_pair(x0, x1);
      const int off = n * BRS + k * 2;
      *reinterpret_cast<uint32_t*>(Bs + off) = h;
      *reinterpret_cast<uint32_t*>(Bs + BPLANE + off) = m;
      *reinterpret_cast<uint32_t*>(Bs + 2 * BPLANE + off) = l;
    }
  }
  __syncthreads();
  for (; rb < n_rowblocks; rb += 2 * stride) {      // two blocks per trip: the register sets alternate at compile time
    load_a(rb + stride, araw[1]);
    process(rb, araw[0]);
    if (rb + stride >= n_rowblocks) break;
    load_a(rb + 2 * stride, araw[0]);
    process(rb + stride, araw[1]);
  }
}

// ---- norm_context + to_kv of a cross-attention in one pass, all 128 output columns per row block --------------------------
// gemm_rows_x3_kernel<.., LN_A> walks the rows once per 64-column block: for the 128 columns of K | V the rows are read, exchanged
// between partner lanes and normalised TWICE, and that vector work (~900 instructions per 32-row block), not memory, set its pace
// (395 us for 1.0 GB).  Here a wave keeps four accumulator tiles: one read of the rows, one LayerNorm, twelve matrix instructions
// in the rows-form kernel's order (same bits), the 32 x 128 bf16 tile out through the wave's LDS tile as whole 256-byte rows.
// gamma / beta come from LDS (the lanes of a half read the same words: broadcasts) instead of 48 registers.
constexpr int CF_KP = 48;                        // d <= 48: three 16-deep steps
constexpr int CF_WRS = 2 * CF_KP + 16;           // bytes per n-row of W in LDS
constexpr int CF_CRS = 2 * 128 + 16;             // bytes per row of a wave's output tile
__global__ __launch_bounds__(256, 2) void context_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                                    int d1, int period, const float* __restrict__ lnw,
                                                                    const float* __restrict__ lnb, const float* __restrict__ wkv,
                                                                    uint16_t* __restrict__ kv16, float* __restrict__ mean,
                                                                    float* __restrict__ rstd, long long rows, int d, float eps,
                                                                    int n_rowblocks) {
  __shared__ __attribute__((aligned(16))) unsigned char Ws[128 * CF_WRS];
  __shared__ __attribute__((aligned(16))) unsigned char Cw[4 * 32 * CF_CRS];
  __shared__ __attribute__((aligned(16))) float Gs[CF_KP], Bt[CF_KP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane & 31, half = lane >> 5;
  for (int i = tid; i < 128 * CF_KP; i += 256) {      // Ws[n][k] = bf16(W[n][k]) (the Linear weight is [128, d]), zero for k >= d
    const int n = i / CF_KP, k = i - n * CF_KP;
    *reinterpret_cast<uint16_t*>(Ws + n * CF_WRS + 2 * k) = f32_to_bf16_bits(k < d ? wkv[(size_t)n * d + k] : 0.f);
  }
  if (tid < CF_KP) Gs[tid] = tid < d ? lnw[tid] : 0.f, Bt[tid] = tid < d ? lnb[tid] : 0.f;
  const int stride = gridDim.x * 4;
  float araw[2][3][8];
  auto load_a = [&](int rb, float (&dst)[3][8]) {
    const long long mm = (long long)rb * 32 + row;
    const bool row_ok = rb < n_rowblocks && mm < rows;
    const long long mc = row_ok ? mm : 0;
    const float* a1 = x + mc * (x2 ? d1 : d);
    const float* a2 = x2 ? x2 + (size_t)((unsigned)mc % (unsigned)period) * (d - d1) - d1 : a1;
    const int split = x2 ? d1 : d;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int v = 0; v < 4; ++v) {      // pairs: d and d1 are even, a pair never straddles the two sources
        const int k = 16 * ks + 8 * half + 2 * v;
        float2 t = make_float2(0.f, 0.f);
        if (row_ok && k < d) t = *reinterpret_cast<const float2*>((k < split ? a1 : a2) + k);
        dst[ks][2 * v] = t.x, dst[ks][2 * v + 1] = t.y;
      }
  };
  auto process = [&](int rb, const float (&ar_in)[3][8]) {
    // the partner lane's half of the row; the serial sums of layernorm_fwd_rows_f32 (same order: same bits)
    float lo[3][8], hi[3][8];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const float other = __shfl_xor(ar_in[ks][v], 32);
        lo[ks][v] = half ? other : ar_in[ks][v];
        hi[ks][v] = half ? ar_in[ks][v] : other;
      }
    auto at = [&](int k) { return ((k >> 3) & 1) ? hi[k >> 4][k & 7] : lo[k >> 4][k & 7]; };
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CF_KP / 2; ++j) s += at(2 * j) + at(2 * j + 1);
    const float mu = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CF_KP / 2; ++j)
      if (2 * j < d) {
        const float t0 = at(2 * j) - mu, t1 = at(2 * j + 1) - mu;
        q += t0 * t0 + t1 * t1;
      }
    const float rs = 1.0f / sqrtf(q / (float)d + eps);
    const long long mm = (long long)rb * 32 + row;
    if (half == 0 && mm < rows) mean[mm] = mu, rstd[mm] = rs;
    typedef __attribute__((ext_vector_type(16))) float v16f;
    v16f acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int kb = 16 * ks + 8 * half;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gs + kb), g1 = *reinterpret_cast<const f32x4*>(Gs + kb + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bt + kb), b1 = *reinterpret_cast<const f32x4*>(Bt + kb + 4);
      float an[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const float gw = v < 4 ? g0[v] : g1[v - 4], bw = v < 4 ? b0[v] : b1[v - 4];
        an[v] = kb + v < d ? (ar_in[ks][v] - mu) * rs * gw + bw : 0.f;
      }
      u32x4 w1;
#pragma unroll
      for (int j = 0; j < 4; ++j) w1[j] = pack_bf16_pair(an[2 * j], an[2 * j + 1]);
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, w1);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, *reinterpret_cast<const bf16x8*>(Ws + (32 * t + row) * CF_WRS + 2 * kb), acc[t], 0, 0, 0);
    }
    unsigned char* ct = Cw + wave * (32 * CF_CRS);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        *reinterpret_cast<uint16_t*>(ct + ((r & 3) + 8 * (r >> 2) + 4 * half) * CF_CRS + 2 * (32 * t + row)) = f32_to_bf16_bits(acc[t][r]);
    uint16_t* cb = kv16 + (long long)rb * 32 * 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // (the same wave reads what it wrote: a wave's LDS operations complete in order)
      const int rr = 4 * j + (lane >> 4), c16 = lane & 15;
      if ((long long)rb * 32 + rr < rows)
        *reinterpret_cast<u32x4*>(cb + (long long)rr * 128 + 8 * c16) = *reinterpret_cast<const u32x4*>(ct + rr * CF_CRS + 16 * c16);
    }
  };
  int rb = blockIdx.x * 4 + wave;
  load_a(rb, araw[0]);
  __syncthreads();
  for (; rb < n_rowblocks; rb += 2 * stride) {
    load_a(rb + stride, araw[1]);
    process(rb, araw[0]);
    if (rb + stride >= n_rowblocks) break;
    load_a(rb + 2 * stride, araw[0]);
    process(rb + stride, araw[1]);
  }
}

// out[i] = sum over s of slabs[s * stride + offset + i], i < n.  Workgroup = 32 columns x 8 slab groups: thread (c, g)
// adds slabs g, g+8, ... in index order, the 8 group sums are then added in group order -- a fixed order, 8-way parallel.
// accumulate != 0: out[i] += the sum (a tied weight's gradient contributions added in arrival order, as autograd would).
__global__ __launch_bounds__(256) void sum_slabs_f32(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                                     int n_slabs, long long stride, long long offset, int accumulate) {
  __shared__ float red[8][32];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x * 32 + c;
  float s = 0.f;
  if (i < n) {
    // same order of additions as a plain loop over k = grp, grp + 8, ...; the loads of eight terms are issued together
    // (one at a time, a thread's ~10 terms cost ten memory latencies: 11 us for a 1 MB slab set)
    const float* src = slabs + offset + i;
    int k = grp;
    for (; k + 56 < n_slabs; k += 64) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + 8 * j) * stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k + 8 * j < n_slabs) ? src[(size_t)(k + 8 * j) * stride] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + 8 * j < n_slabs) s += v[j];
  }
  red[grp][c] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    const float t = ((((((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]) + red[4][c]) + red[5][c]) + red[6][c]) + red[7][c];
    out[i] = accumulate ? out[i] + t : t;
  }
}

void launch_sum_slabs(const float* slabs, float* out, long long n, int n_slabs, long long stride, long long offset,
                      hipStream_t st, int accumulate) {
  hipLaunchKernelGGL(sum_slabs_f32, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, slabs, out, n, n_slabs, stride, offset,
                     accumulate);
}

// partial column sums of x [rows][cols]: block (column block of 64, row chunk) -> part[chunk][cols]; thread = (column, row lane)
__global__ __launch_bounds__(256) void colsum_partial_f32(const float* __restrict__ x, float* __restrict__ part, long long rows,
                                                          int cols, long long rows_per_chunk) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const long long r0 = (long long)blockIdx.y * rows_per_chunk;
  const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float s0 = 0.f, s1 = 0.f;
  if (col < cols) {
    // rows r0 + rl, + 4, + 8, ... alternate between two partial sums; eight loads are in flight at a time
    long long r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x[(r + 4 * j) * cols + col];
#pragma unroll
      for (int j = 0; j < 8; j += 2) s0 += v[j], s1 += v[j + 1];
    }
    for (; r + 4 < r1; r += 8) {
      s0 += x[r * cols + col];
      s1 += x[(r + 4) * cols + col];
    }
    if (r < r1) s0 += x[r * cols + col];
  }
  red[rl][c] = s0 + s1;
  __syncthreads();
  if (rl == 0 && col < cols) part[(long long)blockIdx.y * cols + col] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_gemm_res_f32(const float* a, const float* b, const float* bias, const float* residual, int64_t ldr, float* c,
                    const pv_gemm_desc* d, int relu, void* stream);

int pv_gemm_ex_f32(const float* a, const float* b, const float* bias, const float* residual, int64_t ldr, float* c,
                   const pv_gemm_desc* d, int relu, int32_t flags, void* stream);

int pv_gemm_f32(const float* a, const float* b, const float* bias, float* c, const pv_gemm_desc* d, int relu, void* stream) {
  return pv_gemm_ex_f32(a, b, bias, nullptr, 0, c, d, relu, 0, stream);
}

int pv_gemm_res_f32(const float* a, const float* b, const float* bias, const float* residual, int64_t ldr, float* c,
                    const pv_gemm_desc* d, int relu, void* stream) {
  return pv_gemm_ex_f32(a, b, bias, residual, ldr, c, d, relu, 0, stream);
}

int pv_gemm_ex_f32(const float* a, const float* b, const float* bias, const float* residual, int64_t ldr, float* c,
                   const pv_gemm_desc* d, int relu, int32_t flags, void* stream) {
  const bool one_term = (flags & PV_GEMM_BF16_OPERANDS) != 0;
  const bool a_bf16 = (flags & PV_GEMM_A_IS_BF16) != 0;
  PV_REQUIRE(!a_bf16 || one_term, PV_EINVAL, "pv_gemm_ex_f32: PV_GEMM_A_IS_BF16 needs PV_GEMM_BF16_OPERANDS");
  PV_REQUIRE(a && b && c && d, PV_EINVAL, "pv_gemm_f32: null pointer");
  PV_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0, PV_EINVAL, "pv_gemm_f32: non-positive extent (%d,%d,%d)", d->m, d->n, d->k);
  PV_REQUIRE(d->batch1 > 0 && d->batch2 > 0 && d->k_splits > 0, PV_EINVAL, "pv_gemm_f32: batch counts and k_splits must be >= 1");
  PV_REQUIRE(d->ldc >= d->n, PV_EINVAL, "pv_gemm_f32: ldc < n");
  PV_REQUIRE(!(d->k_splits > 1 && (bias || relu || residual)), PV_EINVAL,
             "pv_gemm_f32: bias / ReLU / residual cannot be applied to split-K partial products");
  PV_REQUIRE(!residual || ((long long)d->batch1 * d->batch2 == 1 && ldr >= d->n), PV_EINVAL,
             "pv_gemm_res_f32: the residual form is for 2-D products (ldr >= n)");
  const long long zs = (long long)d->batch1 * d->batch2 * d->k_splits;
  PV_REQUIRE(zs <= 65535, PV_ESIZE, "pv_gemm_f32: batch1*batch2*k_splits = %lld exceeds the grid limit", zs);
  GemmK g;
  g.a = a, g.b = b, g.bias = bias, g.c = c;
  g.res = residual, g.ldr = ldr;
  g.m = d->m, g.n = d->n, g.k = d->k;
  g.a_rs = d->a_rs, g.a_cs = d->a_cs, g.b_rs = d->b_rs, g.b_cs = d->b_cs, g.ldc = d->ldc;
  g.batch2 = d->batch2, g.k_splits = d->k_splits;
  g.k_chunk = ((d->k + d->k_splits - 1) / d->k_splits + G_BK - 1) / G_BK * G_BK;
  g.a_bs1 = d->a_bs1, g.a_bs2 = d->a_bs2, g.b_bs1 = d->b_bs1, g.b_bs2 = d->b_bs2, g.c_bs1 = d->c_bs1, g.c_bs2 = d->c_bs2;
  g.c_ss = d->c_ss;
  g.relu = relu ? 1 : 0;
  dim3 grid((unsigned)((d->m + G_BM - 1) / G_BM), (unsigned)((d->n + G_BN - 1) / G_BN), (unsigned)zs);
  PV_REQUIRE(grid.y <= 65535, PV_ESIZE, "pv_gemm_f32: n too large for one launch");
  // default: the bf16 x 3 form (f32 accuracy at 3/8 of the matrix-pipe time); PV_GEMM_EXACT_F32=1 keeps the products on the
  // f32 matrix instruction (bit-exact f32 products)
  // (PV_EXACT_F32=1 is the one switch for every split product of the precision="fp32" path; read per call so that a test can flip it)
  const bool exact_f32 = getenv("PV_GEMM_EXACT_F32") != nullptr || getenv("PV_EXACT_F32") != nullptr;
  static const bool no_rows_form = getenv("PV_GEMM_NO_ROWS_FORM") != nullptr;
  const bool rows_form = (!exact_f32 || one_term) && !a_bf16 && !no_rows_form && zs == 1 && d->a_cs == 1 && d->k <= 64 &&
                         d->m >= 2048 && ((uintptr_t)a % 16 == 0);
  if (exact_f32 && !one_term) {
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, as_stream(stream), g);
  } else if (rows_form) {
    const int n_tiles = (d->n + G_BN - 1) / G_BN;
    const int n_rb = (d->m + 31) / 32;
    int per_col = 512 / n_tiles;                    // one round of two workgroups per CU over all column blocks
    if (per_col < 1) per_col = 1;
    if (per_col > (n_rb + 3) / 4) per_col = (n_rb + 3) / 4;
    dim3 rgrid((unsigned)n_tiles, (unsigned)per_col);
    const int vec = (d->a_rs % 4 == 0 && d->k % 4 == 0) ? 4 : ((d->a_rs % 2 == 0 && d->k % 2 == 0) ? 2 : 1);
    const int ksteps = (d->k + 15) / 16;
#define PV_ROWS(KS, T)                                                                                                           \
    if (vec == 4) hipLaunchKernelGGL((gemm_rows_x3_kernel<KS, 4, false, T>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb);       \
    else if (vec == 2) hipLaunchKernelGGL((gemm_rows_x3_kernel<KS, 2, false, T>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb);  \
    else hipLaunchKernelGGL((gemm_rows_x3_kernel<KS, 1, false, T>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb)
    if (one_term) { if (ksteps <= 3) { PV_ROWS(3, 1); } else { PV_ROWS(4, 1); } }
    else { if (ksteps <= 3) { PV_ROWS(3, 3); } else { PV_ROWS(4, 3); } }
#undef PV_ROWS
  } else {
    const bool a_kc = d->a_cs == 1 && d->a_rs != 1;
    const bool b_kc = d->b_cs != 1;
#define PV_TILED(T)                                                                                                          \
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, T>), grid, dim3(256), 0, as_stream(stream), g);            \
    else if (a_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false, T>), grid, dim3(256), 0, as_stream(stream), g);              \
    else if (b_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, T>), grid, dim3(256), 0, as_stream(stream), g);              \
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, T>), grid, dim3(256), 0, as_stream(stream), g)
    if (a_bf16) {
      if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, 1, true>), grid, dim3(256), 0, as_stream(stream), g);
      else if (a_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false, 1, true>), grid, dim3(256), 0, as_stream(stream), g);
      else if (b_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 1, true>), grid, dim3(256), 0, as_stream(stream), g);
      else hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, 1, true>), grid, dim3(256), 0, as_stream(stream), g);
    } else if (one_term) { PV_TILED(1); } else { PV_TILED(3); }
#undef PV_TILED
  }
  return check_launch("pv_gemm_f32");
}

int pv_gemm_rows_bf16out_f32(const float* a, const float* b, const float* bias, uint16_t* c_bf16, const pv_gemm_desc* d,
                             int32_t flags, void* stream) {
  PV_REQUIRE(a && b && c_bf16 && d, PV_EINVAL, "pv_gemm_rows_bf16out_f32: null pointer");
  PV_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0 && d->ldc >= d->n, PV_EINVAL, "pv_gemm_rows_bf16out_f32: bad extents");
  PV_REQUIRE((long long)d->batch1 * d->batch2 * d->k_splits == 1 && d->a_cs == 1 && d->k <= 64 && ((uintptr_t)a % 16 == 0), PV_ESIZE,
             "pv_gemm_rows_bf16out_f32: built for one tall row-major A with K <= 64 (got K=%d, batch %d x %d, k_splits %d)", d->k,
             d->batch1, d->batch2, d->k_splits);
  GemmK g;
  g.a = a, g.b = b, g.bias = bias, g.c = reinterpret_cast<float*>(c_bf16);
  g.res = nullptr, g.ldr = 0;
  g.m = d->m, g.n = d->n, g.k = d->k;
  g.a_rs = d->a_rs, g.a_cs = d->a_cs, g.b_rs = d->b_rs, g.b_cs = d->b_cs, g.ldc = d->ldc;
  g.batch2 = 1, g.k_splits = 1, g.k_chunk = (d->k + G_BK - 1) / G_BK * G_BK;
  g.a_bs1 = g.a_bs2 = g.b_bs1 = g.b_bs2 = g.c_bs1 = g.c_bs2 = 0, g.c_ss = 0;
  g.relu = 0;
  const int n_tiles = (d->n + G_BN - 1) / G_BN;
  const int n_rb = (d->m + 31) / 32;
  int per_col = 512 / n_tiles;
  if (per_col < 1) per_col = 1;
  if (per_col > (n_rb + 3) / 4) per_col = (n_rb + 3) / 4;
  dim3 rgrid((unsigned)n_tiles, (unsigned)per_col);
  const int vec = (d->a_rs % 4 == 0 && d->k % 4 == 0) ? 4 : ((d->a_rs % 2 == 0 && d->k % 2 == 0) ? 2 : 1);
  const int ksteps = (d->k + 15) / 16;
#define PV_ROWS16(KS, T)                                                                                                        \
    if (vec == 4) hipLaunchKernelGGL((gemm_rows_x3_kernel<KS, 4, true, T>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb);       \
    else if (vec == 2) hipLaunchKernelGGL((gemm_rows_x3_kernel<KS, 2, true, T>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb);  \
    else hipLaunchKernelGGL((gemm_rows_x3_kernel<KS, 1, true, T>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb)
  if (flags & PV_GEMM_BF16_OPERANDS) { if (ksteps <= 3) { PV_ROWS16(3, 1); } else { PV_ROWS16(4, 1); } }
  else { if (ksteps <= 3) { PV_ROWS16(3, 3); } else { PV_ROWS16(4, 3); } }
#undef PV_ROWS16
  return check_launch("pv_gemm_rows_bf16out_f32");
}

int pv_context_fwd_bf16(const float* x, const float* x2, int32_t d1, int64_t period, const float* ln_w, const float* ln_b,
                        const float* w_kv, uint16_t* kv16, float* mean, float* rstd, int64_t rows, int32_t d, int32_t kdim,
                        float eps, void* stream) {
  PV_REQUIRE(x && ln_w && ln_b && w_kv && kv16 && mean && rstd, PV_EINVAL, "pv_context_fwd_bf16: null pointer");
  PV_REQUIRE(!x2 || (d1 > 0 && d1 < d && d1 % 2 == 0 && period > 0 && period <= 0x7fffffffLL && ((uintptr_t)x2 & 7) == 0), PV_EINVAL,
             "pv_context_fwd_bf16: two sources need an even 0 < d1 < d, a period and an 8-byte aligned x2");
  PV_REQUIRE(rows > 0 && rows <= 0x7fffffffLL && d > 0 && d <= 48 && d % 2 == 0, PV_ESIZE,
             "pv_context_fwd_bf16: d=%d must be even and in 2..48", d);
  PV_REQUIRE(kdim > 0 && kdim % 64 == 0, PV_ESIZE, "pv_context_fwd_bf16: kdim=%d must be a multiple of 64", kdim);
  PV_REQUIRE(((uintptr_t)x & 15) == 0, PV_EINVAL, "pv_context_fwd_bf16: x must be 16-byte aligned");
  GemmK g;
  g.a = x, g.b = w_kv, g.bias = nullptr, g.c = reinterpret_cast<float*>(kv16);      // B = w_kv^T: [K = d, N = kdim], k-contiguous
  g.res = nullptr, g.ldr = 0;
  g.m = (int)rows, g.n = kdim, g.k = d;
  g.a_rs = x2 ? d1 : d, g.a_cs = 1, g.b_rs = 1, g.b_cs = d, g.ldc = kdim;
  g.batch2 = 1, g.k_splits = 1, g.k_chunk = (d + G_BK - 1) / G_BK * G_BK;
  g.a_bs1 = g.a_bs2 = g.b_bs1 = g.b_bs2 = g.c_bs1 = g.c_bs2 = 0, g.c_ss = 0;
  g.relu = 0;
  const int n_tiles = kdim / G_BN;
  const int n_rb = (int)((rows + 31) / 32);
  int per_col = 512 / n_tiles;
  if (per_col < 1) per_col = 1;
  if (per_col > (n_rb + 3) / 4) per_col = (n_rb + 3) / 4;
  dim3 rgrid((unsigned)n_tiles, (unsigned)per_col);
  if (kdim == 128 && ((uintptr_t)kv16 & 15) == 0 && !getenv("PV_CONTEXT_FWD_TWO_COLUMN_BLOCKS")) {
    // all 128 columns per row block: the rows are read and normalised once
    const int nb = std::min((n_rb + 3) / 4, 2 * kNumCU);
    hipLaunchKernelGGL(context_fwd_rows_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), x, x2, d1, (int)period, ln_w,
                       ln_b, w_kv, kv16, mean, rstd, (long long)rows, d, eps, n_rb);
    return check_launch("pv_context_fwd_bf16");
  }
  const GemmLnA ln = {ln_w, ln_b, mean, rstd, eps, x2, d1, (int)period};
  hipLaunchKernelGGL((gemm_rows_x3_kernel<3, 2, true, 1, true>), rgrid, dim3(256), 0, as_stream(stream), g, n_rb, ln);
  return check_launch("pv_context_fwd_bf16");
}

int pv_sum_slabs_f32(const float* slabs, float* out, int64_t n, int32_t n_slabs, void* stream) {
  PV_REQUIRE(slabs && out && n > 0 && n_slabs > 0, PV_EINVAL, "pv_sum_slabs_f32: bad arguments");
  launch_sum_slabs(slabs, out, (long long)n, n_slabs, (long long)n, 0, as_stream(stream), 0);
  return check_launch("pv_sum_slabs_f32");
}

int pv_sum_slabs_acc_f32(const float* slabs, float* out, int64_t n, int32_t n_slabs, int32_t accumulate, void* stream) {
  PV_REQUIRE(slabs && out && n > 0 && n_slabs > 0, PV_EINVAL, "pv_sum_slabs_acc_f32: bad arguments");
  launch_sum_slabs(slabs, out, (long long)n, n_slabs, (long long)n, 0, as_stream(stream), accumulate);
  return check_launch("pv_sum_slabs_acc_f32");
}

size_t pv_colsum_workspace_floats(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  long long chunks = (rows + 31) / 32;
  if (chunks > 512) chunks = 512;     // enough workgroups to fill the chip even for a 64-column matrix
  return (size_t)chunks * cols;
}

int pv_colsum_f32(const float* x, float* out, int64_t rows, int32_t cols, float* workspace, int32_t accumulate, void* stream) {
  PV_REQUIRE(x && out && workspace && rows > 0 && cols > 0, PV_EINVAL, "pv_colsum_f32: bad arguments");
  long long chunks = (rows + 31) / 32;
  if (chunks > 512) chunks = 512;
  const long long per = (rows + chunks - 1) / chunks;
  chunks = (rows + per - 1) / per;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(colsum_partial_f32, dim3((unsigned)((cols + 63) / 64), (unsigned)chunks), dim3(256), 0, st, x, workspace,
                     (long long)rows, cols, per);
  launch_sum_slabs(workspace, out, cols, (int)chunks, cols, 0, st, accumulate);
  return check_launch("pv_colsum_f32");
}

}  // extern "C"
