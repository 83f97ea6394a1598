// Fused attention softmax(scale * q k^T) v on the f32 matrix cores (v_mfma_f32_32x32x2_f32) with an online softmax: the
// [queries x keys] score matrix never leaves registers.
// replaces: sim = einsum('b i d, b j d -> b i j', q, k) * scale; attn = sim.softmax(dim=-1);
//           out = einsum('b i j, b j d -> b i d', attn, v)   of perceiver_pytorch's Attention.forward (third-party), as
//           instantiated by predict_pv_yield/models/perceiver/perceiver.py:70-80 (cross-attention: 128 latents x 4096
//           positions, 1 head x 64; latent self-attention: 128 x 128, 8 heads x 64).
// Workgroup = 4 waves; wave w owns 32 queries; all waves walk the keys in tiles of 32 that are staged once in LDS.
// Per tile and wave:  S^T[j][i] = K_j Q_i^T  (32 MFMAs; transposed on purpose: a lane then holds ONE query's scores, so
// the running max / sum and the rescaling of the output accumulator are lane-local),  P^T = exp(scale S^T - m),
// O^T[d][i] += V_j^T P^T  (32 MFMAs; the probabilities go from the S accumulator straight into the B operand: the MFMA
// contraction index is free to follow the accumulator's row order, so no lane exchange is needed).
// The backward recomputes the probabilities tile by tile from the saved log-sum-exp.
#include "pv_common.h"

namespace pv {

typedef float v16f_a __attribute__((ext_vector_type(16)));

constexpr int AT_D = 64;        // head dimension
constexpr int AT_LD = 68;       // LDS row stride (words) of the 32 x 64 key / value tiles
constexpr int AT_TJ = 32;       // keys per tile

struct AttnGeom {
  int n_q, n_k, heads;
  long long q_bs, q_rs;     // q / o / dq / do: element (b, h, i, d) at b*q_bs + i*q_rs + h*64 + d
  long long k_bs, k_rs;     // k, v (and dk, dv): (b, h, j, d) at b*k_bs + j*k_rs + h*64 + d (from separate base pointers)
  float scale;
};

// accumulator row held by register r of lane-half `half` (32x32 MFMA C layout)
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__global__ __launch_bounds__(256) void attn_fwd_f32(const float* __restrict__ q, const float* __restrict__ k,
                                                     const float* __restrict__ v, float* __restrict__ o,
                                                     float* __restrict__ lse, AttnGeom g) {
  __shared__ __attribute__((aligned(16))) float Ks[AT_TJ * AT_LD];
  __shared__ __attribute__((aligned(16))) float Vs[AT_TJ * AT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int i = blockIdx.x * 128 + wave * 32 + col;           // this lane's query
  const float* qb = q + b * g.q_bs + h * AT_D;
  const float* kb = k + b * g.k_bs + h * AT_D;
  const float* vb = v + b * g.k_bs + h * AT_D;
  // B operand of S^T = K Q^T: lane (query = col, d = 2 kk + half)
  float qreg[32];
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) qreg[kk] = i < g.n_q ? qb[(long long)i * g.q_rs + 2 * kk + half] : 0.f;

  v16f_a acc0, acc1;   // O^T rows d = 0..31 / 32..63, column = query
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale_log2e = g.scale * 1.44269504088896340736f;

  // staging: thread t moves 8 floats (two float4) of the 32 x 64 tile: row t / 8, columns 8 (t % 8) ..
  const int st_row = tid >> 3, st_col = (tid & 7) * 8;
  float kr[8], vr[8];
  auto load_tile = [&](int j0) {
    const int j = j0 + st_row;
    const bool ok = j < g.n_k;
    const float* kp = kb + (long long)(ok ? j : 0) * g.k_rs + st_col;
    const float* vp = vb + (long long)(ok ? j : 0) * g.k_rs + st_col;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(kp), a1 = *reinterpret_cast<const f32x4*>(kp + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(vp), b1 = *reinterpret_cast<const f32x4*>(vp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kr[e] = ok ? a0[e] : 0.f, kr[4 + e] = ok ? a1[e] : 0.f;
      vr[e] = ok ? b0[e] : 0.f, vr[4 + e] = ok ? b1[e] : 0.f;
    }
  };
  load_tile(0);
  for (int j0 = 0; j0 < g.n_k; j0 += AT_TJ) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(Ks + st_row * AT_LD + st_col) = (f32x4){kr[0], kr[1], kr[2], kr[3]};
    *reinterpret_cast<f32x4*>(Ks + st_row * AT_LD + st_col + 4) = (f32x4){kr[4], kr[5], kr[6], kr[7]};
    *reinterpret_cast<f32x4*>(Vs + st_row * AT_LD + st_col) = (f32x4){vr[0], vr[1], vr[2], vr[3]};
    *reinterpret_cast<f32x4*>(Vs + st_row * AT_LD + st_col + 4) = (f32x4){vr[4], vr[5], vr[6], vr[7]};
    __syncthreads();
    if (j0 + AT_TJ < g.n_k) load_tile(j0 + AT_TJ);
    // S^T tile: rows = keys, column = this lane's query
    v16f_a s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[col * AT_LD + 2 * kk + half], qreg[kk], s, 0, 0, 0);
    float m_tile = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = j0 + acc_row(r, half) < g.n_k;
      s[r] = ok ? s[r] * scale_log2e : -INFINITY;        // scores in log2 units: every exponential below is one v_exp_f32
      m_tile = fmaxf(m_tile, s[r]);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // first tile: exp2(-inf) = 0 on zero accumulators
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] *= alpha, acc1[r] *= alpha;
    // O^T += V^T P^T: contraction slot kk of lane-half `half` is key acc_row(kk, half) -- the row register kk already holds
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int jr = acc_row(kk, half);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[jr * AT_LD + col], s[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[jr * AT_LD + 32 + col], s[kk], acc1, 0, 0, 0);
    }
  }
  if (i < g.n_q) {
    const float inv = 1.0f / l_run;
    float* ob = o + b * g.q_bs + (long long)i * g.q_rs + h * AT_D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = acc_row(r, half);
      ob[d] = acc0[r] * inv;
      ob[32 + d] = acc1[r] * inv;
    }
    if (half == 0) lse[((long long)b * g.heads + h) * g.n_q + i] = m_run * 0.69314718055994530942f + logf(l_run);
  }
}


// delta[b, h, i] = sum_d dO[b, i, h, d] * O[b, i, h, d]  (the row term of the softmax backward)
__global__ __launch_bounds__(256) void attn_delta_f32(const float* __restrict__ o, const float* __restrict__ dout,
                                                       float* __restrict__ delta, AttnGeom g, int batch) {
  const long long idx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per (b, h, i)
  const int lane = threadIdx.x & 63;
  if (idx >= (long long)batch * g.heads * g.n_q) return;
  const int i = (int)(idx % g.n_q);
  const int h = (int)((idx / g.n_q) % g.heads);
  const long long b = idx / ((long long)g.n_q * g.heads);
  const long long off = b * g.q_bs + (long long)i * g.q_rs + h * AT_D + lane;
  float v = o[off] * dout[off];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  if (lane == 0) delta[idx] = v;
}

// Backward for n_q <= 128: one workgroup per (b, h).  Q and dO of all (<= 128) queries sit in LDS; wave w walks the key
// tiles w, w+4, ... on its own (no barriers inside the loop): it keeps that tile's K and V as MFMA B operands in
// registers, recomputes P from the saved log-sum-exp for each of the 4 query tiles, and accumulates
//   dV_j^T += dO^T P,  dK_j^T += Q^T dS  (complete within the wave: written straight out)  and
//   dQ^T  += K_j^T dS^T  (partial over this wave's keys: the four waves' partials are added through LDS at the end).
constexpr int AB_TS = 33;   // row stride of the per-wave dS transpose scratch

__global__ __launch_bounds__(256) void attn_bwd_f32(const float* __restrict__ q, const float* __restrict__ k,
                                                     const float* __restrict__ v, const float* __restrict__ dout,
                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                     float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                     AttnGeom g, int tiles_per_split, long long dq_ss) {
  __shared__ __attribute__((aligned(16))) float Qs[128 * AT_LD];
  __shared__ __attribute__((aligned(16))) float Os[128 * AT_LD];        // dO
  __shared__ __attribute__((aligned(16))) float Kw[4][AT_TJ * AT_LD];   // per-wave K tile (A operand of dQ^T = K^T dS^T)
  __shared__ float Tw[4][32 * AB_TS];                                    // per-wave dS tile, read back transposed
  __shared__ __attribute__((aligned(16))) float Ls[128];
  __shared__ __attribute__((aligned(16))) float Ds[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const float* qb = q + b * g.q_bs + h * AT_D;
  const float* ob = dout + b * g.q_bs + h * AT_D;
  const float* kb = k + b * g.k_bs + h * AT_D;
  const float* vb = v + b * g.k_bs + h * AT_D;
  // ---- stage Q, dO (zero rows beyond n_q), lse (+inf beyond n_q so that P = 0 there), delta ---------------------------
  for (int idx = tid; idx < 128 * 16; idx += 256) {
    const int i = idx >> 4, c4 = (idx & 15) * 4;
    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, oa = {0.f, 0.f, 0.f, 0.f};
    if (i < g.n_q) {
      qa = *reinterpret_cast<const f32x4*>(qb + (long long)i * g.q_rs + c4);
      oa = *reinterpret_cast<const f32x4*>(ob + (long long)i * g.q_rs + c4);
    }
    *reinterpret_cast<f32x4*>(Qs + i * AT_LD + c4) = qa;
    *reinterpret_cast<f32x4*>(Os + i * AT_LD + c4) = oa;
  }
  if (tid < 128) {
    const long long li = ((long long)b * g.heads + h) * g.n_q + tid;
    Ls[tid] = tid < g.n_q ? lse[li] * 1.44269504088896340736f : INFINITY;   // log2 units: P = exp2(scale log2e S - lse log2e)
    Ds[tid] = tid < g.n_q ? delta[li] : 0.f;
  }
  __syncthreads();

  v16f_a dqa[4][2];   // dQ^T partial of this wave: [query tile][d half], rows d, column = query
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqa[it][t][r] = 0.f;
  float* Kt = Kw[wave];
  float* Tt = Tw[wave];
  const float scale_log2e = g.scale * 1.44269504088896340736f;
  const int n_tiles = (g.n_k + AT_TJ - 1) / AT_TJ;
  const int n_qt = (g.n_q + 31) / 32;
  const int tile0 = blockIdx.x * tiles_per_split;
  const int tile1 = tile0 + tiles_per_split < n_tiles ? tile0 + tiles_per_split : n_tiles;
  dq += blockIdx.x * dq_ss;     // this key range's partial dQ (summed over the splits afterwards)
  // K, V of a tile as B operands: lane (key = col, d = 2 kk + half).  The NEXT tile's 64 loads per lane are issued before
  // the current tile's ~640 MFMAs (one wave per SIMD: nothing else would hide their issue time and latency).
  float kreg[32], vreg[32], knxt[32], vnxt[32];
  auto load_kv = [&](int tile, float (&kd)[32], float (&vd)[32]) {
    const int jj = tile * AT_TJ + col;
    const bool ok = tile < tile1 && jj < g.n_k;
    const float* kp = kb + (long long)(ok ? jj : 0) * g.k_rs + half;
    const float* vp = vb + (long long)(ok ? jj : 0) * g.k_rs + half;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      kd[kk] = ok ? kp[2 * kk] : 0.f;
      vd[kk] = ok ? vp[2 * kk] : 0.f;
    }
  };
  load_kv(tile0 + wave, knxt, vnxt);
  for (int tile = tile0 + wave; tile < tile1; tile += 4) {
    const int j0 = tile * AT_TJ;
    const int j = j0 + col;                       // this lane's key (B-operand column)
    const bool j_ok = j < g.n_k;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      kreg[kk] = knxt[kk];
      vreg[kk] = vnxt[kk];
      Kt[col * AT_LD + 2 * kk + half] = kreg[kk];   // K also into LDS for the transposed use
    }
    load_kv(tile + 4, knxt, vnxt);
    v16f_a dv0, dv1, dk0, dk1;   // dV^T / dK^T of this key tile: rows d, column = key
#pragma unroll
    for (int r = 0; r < 16; ++r) dv0[r] = 0.f, dv1[r] = 0.f, dk0[r] = 0.f, dk1[r] = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it < n_qt) {
        const float* Qt = Qs + it * 32 * AT_LD;
        const float* Ot = Os + it * 32 * AT_LD;
        // S[i][j] = Q K^T and dP[i][j] = dO V^T: A operand lane (query = col, d = 2 kk + half), rows = queries, column = key
        v16f_a sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
          sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[col * AT_LD + 2 * kk + half], kreg[kk], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ot[col * AT_LD + 2 * kk + half], vreg[kk], dp, 0, 0, 0);
        }
        // P = exp(scale S - lse_i), dS = scale * P * (dP - delta_i); masked keys give P = 0
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {     // registers 4 q4 + 0..3 hold queries 32 it + 8 q4 + 4 half + 0..3: one 16-byte read each
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + it * 32 + 8 * q4 + 4 * half);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + it * 32 + 8 * q4 + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q4 + e;
            const float pv_ = j_ok ? __builtin_amdgcn_exp2f(sc[r] * scale_log2e - l4[e]) : 0.f;
            sc[r] = pv_;                                      // sc now holds P
            dp[r] = g.scale * pv_ * (dp[r] - d4[e]);          // dp now holds dS
            Tt[acc_row(r, half) * AB_TS + col] = dp[r];
          }
        }
        // dV^T += dO^T P, dK^T += Q^T dS: contraction slot kk of lane-half `half` is query acc_row(kk, half)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const int ir = acc_row(kk, half);
          dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Ot[ir * AT_LD + col], sc[kk], dv0, 0, 0, 0);
          dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Ot[ir * AT_LD + 32 + col], sc[kk], dv1, 0, 0, 0);
          dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[ir * AT_LD + col], dp[kk], dk0, 0, 0, 0);
          dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[ir * AT_LD + 32 + col], dp[kk], dk1, 0, 0, 0);
        }
        // dQ^T[d][i] += K^T[d][j] dS^T[j][i]: B operand lane (query = col, key = 2 kk + half) read back transposed
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const int jr = 2 * kk + half;
          const float dst = Tt[col * AB_TS + jr];
          dqa[it][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[jr * AT_LD + col], dst, dqa[it][0], 0, 0, 0);
          dqa[it][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[jr * AT_LD + 32 + col], dst, dqa[it][1], 0, 0, 0);
        }
      }
    }
    if (j_ok) {
      float* dkp = dk + b * g.k_bs + (long long)j * g.k_rs + h * AT_D;
      float* dvp = dv + b * g.k_bs + (long long)j * g.k_rs + h * AT_D;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = acc_row(r, half);
        dkp[d] = dk0[r], dkp[32 + d] = dk1[r];
        dvp[d] = dv0[r], dvp[32 + d] = dv1[r];
      }
    }
  }
  // ---- add the four waves' dQ^T partials (wave order) and write dq; two query tiles per round through Qs | Os ----------
  for (int round = 0; round < 2; ++round) {
    __syncthreads();   // everyone is done with Qs / Os (first round) or with the previous round's sums
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = 2 * round + u;
      float* dst = (u == 0 ? Qs : Os);   // reused as [wave][query 32][d 64 (+1)] = 4 * 2080 floats <= 128 * 68: a lane writes
                                         // its query's column (pitch 65: 32 lanes on 32 banks), the sums read along d
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(wave * 32 + col) * 65 + 32 * t + acc_row(r, half)] = dqa[it][t][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * 32 * 64; idx += 256) {
      const int u = idx / (32 * 64), rem = idx % (32 * 64);
      const int il = rem / 64, d = rem % 64;
      const int i = (2 * round + u) * 32 + il;
      if (i < g.n_q) {
        const float* src = (u == 0 ? Qs : Os);
        const float sum = ((src[(0 * 32 + il) * 65 + d] + src[(1 * 32 + il) * 65 + d]) + src[(2 * 32 + il) * 65 + d]) +
                          src[(3 * 32 + il) * 65 + d];
        dq[b * g.q_bs + (long long)i * g.q_rs + h * AT_D + d] = sum;
      }
    }
  }
}

void launch_sum_slabs(const float* slabs, float* out, long long n, int n_slabs, long long stride, long long offset,
                      hipStream_t st, int accumulate = 0);   // gemm_f32.hip

// delta = rowsum(dO * O), shared with the bf16-operand backward (attention_bf16.hip)
void launch_attn_delta(const float* o, const float* dout, float* delta, const pv_attention_desc* d, hipStream_t st) {
  AttnGeom g;
  g.n_q = d->n_q, g.n_k = d->n_k, g.heads = d->heads;
  g.q_bs = d->q_batch_stride, g.q_rs = d->q_row_stride, g.k_bs = d->k_batch_stride, g.k_rs = d->k_row_stride;
  g.scale = d->scale;
  const long long rows = (long long)d->batch * d->heads * d->n_q;
  hipLaunchKernelGGL(attn_delta_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, o, dout, delta, g, d->batch);
}

}  // namespace pv

using namespace pv;

extern "C" {

static int attn_geom(const pv_attention_desc* d, AttnGeom* g, const char* who) {
  PV_REQUIRE(d, PV_EINVAL, "%s: null descriptor", who);
  PV_REQUIRE(d->batch > 0 && d->heads > 0 && d->n_q > 0 && d->n_k > 0, PV_EINVAL, "%s: non-positive extent", who);
  PV_REQUIRE(d->head_dim == AT_D, PV_ESIZE, "%s: head_dim must be %d", who, AT_D);
  PV_REQUIRE(d->q_row_stride % 4 == 0 && d->k_row_stride % 4 == 0 && d->q_batch_stride % 4 == 0 && d->k_batch_stride % 4 == 0,
             PV_EINVAL, "%s: strides must be multiples of 4 elements (16-byte rows)", who);
  PV_REQUIRE(d->batch <= 65535 && d->heads <= 65535, PV_ESIZE, "%s: batch / heads exceed the grid limit", who);
  g->n_q = d->n_q, g->n_k = d->n_k, g->heads = d->heads;
  g->q_bs = d->q_batch_stride, g->q_rs = d->q_row_stride, g->k_bs = d->k_batch_stride, g->k_rs = d->k_row_stride;
  g->scale = d->scale;
  return PV_OK;
}

int pv_attention_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, const pv_attention_desc* d,
                         void* stream) {
  AttnGeom g;
  int rc = attn_geom(d, &g, "pv_attention_fwd_f32");
  if (rc) return rc;
  PV_REQUIRE(q && k && v && o && lse, PV_EINVAL, "pv_attention_fwd_f32: null pointer");
  PV_REQUIRE(((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0), PV_EINVAL, "pv_attention_fwd_f32: k / v must be 16-byte aligned");
  dim3 grid((unsigned)((d->n_q + 127) / 128), (unsigned)d->heads, (unsigned)d->batch);
  hipLaunchKernelGGL(attn_fwd_f32, grid, dim3(256), 0, as_stream(stream), q, k, v, o, lse, g);
  return check_launch("pv_attention_fwd_f32");
}

// key splits of the backward: enough workgroups to fill the chip a few times over (one resident workgroup per CU)
static int attn_bwd_splits(const pv_attention_desc* d) {
  const long long groups = (long long)d->batch * d->heads;
  const int n_tiles = (d->n_k + AT_TJ - 1) / AT_TJ;
  int s = (int)((1024 + groups - 1) / groups);
  if (s > n_tiles / 8) s = n_tiles / 8;      // at least 8 key tiles (2 per wave) per workgroup
  return s < 1 ? 1 : s;
}

size_t pv_attention_bwd_workspace_floats(const pv_attention_desc* d) {
  if (!d || d->batch <= 0 || d->heads <= 0 || d->n_q <= 0) return 0;
  const size_t rows = (size_t)d->batch * d->heads * d->n_q;
  const int s = attn_bwd_splits(d);
  return rows + (s > 1 ? (size_t)s * d->batch * d->q_batch_stride : 0);
}

int pv_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse,
                         float* delta_ws, float* dq, float* dk, float* dv, const pv_attention_desc* d, void* stream) {
  AttnGeom g;
  int rc = attn_geom(d, &g, "pv_attention_bwd_f32");
  if (rc) return rc;
  PV_REQUIRE(q && k && v && o && dout && lse && delta_ws && dq && dk && dv, PV_EINVAL, "pv_attention_bwd_f32: null pointer");
  PV_REQUIRE(d->n_q <= 128, PV_ESIZE, "pv_attention_bwd_f32: n_q=%d > 128 queries per (batch, head) is not built", d->n_q);
  PV_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)dout % 16 == 0), PV_EINVAL, "pv_attention_bwd_f32: q / dout must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const long long rows = (long long)d->batch * d->heads * d->n_q;
  hipLaunchKernelGGL(attn_delta_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, o, dout, delta_ws, g, d->batch);
  const int splits = attn_bwd_splits(d);
  const int n_tiles = (d->n_k + AT_TJ - 1) / AT_TJ;
  const int per = ((n_tiles + splits - 1) / splits + 3) / 4 * 4;
  const int nsp = (n_tiles + per - 1) / per;
  if (nsp > 1) {
    // dQ partials go to the workspace behind delta, laid out like dq per split, then are added in split order.
    // (this path assumes q is densely packed per batch: q_batch_stride = n_q * q_row_stride, checked by the caller)
    float* part = delta_ws + rows;
    const long long n = (long long)d->batch * d->q_batch_stride;
    hipLaunchKernelGGL(attn_bwd_f32, dim3((unsigned)nsp, (unsigned)d->heads, (unsigned)d->batch), dim3(256), 0, st, q, k, v, dout,
                       lse, (const float*)delta_ws, part, dk, dv, g, per, n);
    launch_sum_slabs(part, dq, n, nsp, n, 0, st);
  } else {
    hipLaunchKernelGGL(attn_bwd_f32, dim3(1, (unsigned)d->heads, (unsigned)d->batch), dim3(256), 0, st, q, k, v, dout, lse,
                       (const float*)delta_ws, dq, dk, dv, g, n_tiles, 0ll);
  }
  return check_launch("pv_attention_bwd_f32");
}

}  // extern "C"
