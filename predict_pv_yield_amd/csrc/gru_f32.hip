// GRU recurrence (nn.GRU, batch_first, gate order r, z, n) for the encoder / decoder heads of the Perceiver models
// (predict_pv_yield/models/perceiver/perceiver.py:94-109,193-196): hidden size 16, at most a few dozen time steps.
// The input projection gi = x W_ih^T + b_ih of ALL time steps is a GEMM (pv_gemm_f32); these kernels run the sequential
// part: one wave per batch row, lane j = hidden unit j, h broadcast across lanes with shuffles, W_hh from L1/L2.
//   r = sigmoid(gi_r + W_hr h + b_hr);  z = sigmoid(gi_z + W_hz h + b_hz)
//   n = tanh(gi_n + r * (W_hn h + b_hn));  h' = (1 - z) * n + z * h
// The forward saves (r, z, n, W_hn h + b_hn) per step; the backward walks the sequence in reverse and produces the
// gradient of gi (hence, through the GEMM's backward, of x / W_ih / b_ih), of h0, and per-row partial sums of
// dW_hh / db_hh that are added over the batch in index order (deterministic).
#include "pv_common.h"

namespace pv {

constexpr int GRU_MAXH = 64;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(64) void gru_seq_fwd_f32(const float* __restrict__ gi, const float* __restrict__ h0,
                                                       const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                       float* __restrict__ out, float* __restrict__ saved, int t_len, int hs) {
  const int row = blockIdx.x, j = threadIdx.x;
  const bool act = j < hs;
  float h = (act && h0) ? h0[(size_t)row * hs + j] : 0.f;
  const float br = act ? b_hh[j] : 0.f, bz = act ? b_hh[hs + j] : 0.f, bn = act ? b_hh[2 * hs + j] : 0.f;
  for (int t = 0; t < t_len; ++t) {
    float ar = br, az = bz, an = bn;
    for (int k = 0; k < hs; ++k) {
      const float hk = __shfl(h, k, 64);
      if (act) {
        ar = fmaf(w_hh[(size_t)j * hs + k], hk, ar);
        az = fmaf(w_hh[(size_t)(hs + j) * hs + k], hk, az);
        an = fmaf(w_hh[(size_t)(2 * hs + j) * hs + k], hk, an);
      }
    }
    if (act) {
      const float* g = gi + ((size_t)row * t_len + t) * 3 * hs;
      const float r = sigmoidf_(g[j] + ar);
      const float z = sigmoidf_(g[hs + j] + az);
      const float n = tanhf(g[2 * hs + j] + r * an);
      h = (1.0f - z) * n + z * h;
      out[((size_t)row * t_len + t) * hs + j] = h;
      float* s = saved + ((size_t)row * t_len + t) * 4 * hs;
      s[j] = r, s[hs + j] = z, s[2 * hs + j] = n, s[3 * hs + j] = an;
    }
  }
}

__global__ __launch_bounds__(64) void gru_seq_bwd_f32(const float* __restrict__ dout, const float* __restrict__ dh_last,
                                                       const float* __restrict__ h0, const float* __restrict__ out,
                                                       const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                       float* __restrict__ dgi, float* __restrict__ dh0,
                                                       float* __restrict__ dw_part /* [rows][3hs][hs] */,
                                                       float* __restrict__ db_part /* [rows][3hs] */, int t_len, int hs) {
  const int row = blockIdx.x, j = threadIdx.x;
  const bool act = j < hs;
  float dh = (act && dh_last) ? dh_last[(size_t)row * hs + j] : 0.f;
  float dbr = 0.f, dbz = 0.f, dbn = 0.f;
  float* dwp = dw_part + (size_t)row * 3 * hs * hs;
  for (int i = j; i < 3 * hs * hs; i += 64) dwp[i] = 0.f;
  for (int t = t_len - 1; t >= 0; --t) {
    float gr = 0.f, gz = 0.f, gn = 0.f, hprev = 0.f, dh_keep = 0.f;
    if (act) {
      if (dout) dh += dout[((size_t)row * t_len + t) * hs + j];
      const float* s = saved + ((size_t)row * t_len + t) * 4 * hs;
      const float r = s[j], z = s[hs + j], n = s[2 * hs + j], an = s[3 * hs + j];
      hprev = t > 0 ? out[((size_t)row * t_len + t - 1) * hs + j] : (h0 ? h0[(size_t)row * hs + j] : 0.f);
      const float dn_pre = dh * (1.0f - z) * (1.0f - n * n);
      const float dz_pre = dh * (hprev - n) * z * (1.0f - z);
      const float dr_pre = dn_pre * an * r * (1.0f - r);
      float* d = dgi + ((size_t)row * t_len + t) * 3 * hs;
      d[j] = dr_pre, d[hs + j] = dz_pre, d[2 * hs + j] = dn_pre;
      gr = dr_pre, gz = dz_pre, gn = dn_pre * r;          // gradients of the three W_hh h + b_hh pre-activations
      dbr += gr, dbz += gz, dbn += gn;
      dh_keep = dh * z;
    }
    // dW_hh[g*hs + j][k] += g_j * hprev_k ;  dh_prev[k] = dh * z + sum_j W_hh[g*hs + j][k] * g_j
    float dh_prev = dh_keep;
    for (int k = 0; k < hs; ++k) {
      const float hk = __shfl(hprev, k, 64);
      if (act) {
        dwp[(size_t)j * hs + k] += gr * hk;
        dwp[(size_t)(hs + j) * hs + k] += gz * hk;
        dwp[(size_t)(2 * hs + j) * hs + k] += gn * hk;
      }
      // column k of W_hh^T g: every lane contributes its row j, summed across lanes, kept by lane k
      float c = act ? (w_hh[(size_t)j * hs + k] * gr + w_hh[(size_t)(hs + j) * hs + k] * gz + w_hh[(size_t)(2 * hs + j) * hs + k] * gn) : 0.f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
      if (j == k) dh_prev += c;
    }
    dh = dh_prev;
  }
  if (act) {
    if (dh0) dh0[(size_t)row * hs + j] = dh;
    float* dbp = db_part + (size_t)row * 3 * hs;
    dbp[j] = dbr, dbp[hs + j] = dbz, dbp[2 * hs + j] = dbn;
  }
}

// ---- the same two kernels for a hidden size known at compile time (16: every GRU of the reference's models) ---------------
// The recurrence is a latency chain of a few dozen steps on ONE wave per batch row: what a step costs is its longest dependent
// sequence.  The general kernels above fetch W_hh from memory inside the k loop (48 loads per step) and, backward, add into
// dW_hh in memory (48 load + add + store per step) and reduce every column of W_hh^T g across the wave (6 shuffles x hs).
// Here lane j keeps its three rows of W_hh (forward) / its three rows AND its three columns (backward) and its rows of dW_hh in
// registers; W_hh^T g is a loop over the lanes' g values (broadcast shuffles, as the forward does with h).  Forward: identical
// bits.  Backward: dW_hh / db_hh / d gi identical, dh_prev summed over j in index order instead of a shuffle tree.
template <int HS>
__global__ __launch_bounds__(64) void gru_seq_fwd_hs_f32(const float* __restrict__ gi, const float* __restrict__ h0,
                                                          const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                          float* __restrict__ out, float* __restrict__ saved, int t_len) {
  const int row = blockIdx.x, j = threadIdx.x;
  const bool act = j < HS;
  const int jc = act ? j : 0;
  float wr[HS], wz[HS], wn[HS];
#pragma unroll
  for (int k = 0; k < HS; ++k)
    wr[k] = w_hh[(size_t)jc * HS + k], wz[k] = w_hh[(size_t)(HS + jc) * HS + k], wn[k] = w_hh[(size_t)(2 * HS + jc) * HS + k];
  float h = (act && h0) ? h0[(size_t)row * HS + j] : 0.f;
  const float br = b_hh[jc], bz = b_hh[HS + jc], bn = b_hh[2 * HS + jc];
  const float* g = gi + (size_t)row * t_len * 3 * HS;
  float g0 = act ? g[j] : 0.f, g1 = act ? g[HS + j] : 0.f, g2 = act ? g[2 * HS + j] : 0.f;
  for (int t = 0; t < t_len; ++t) {
    const float c0 = g0, c1 = g1, c2 = g2;
    if (t + 1 < t_len && act) {      // the next step's input projection is in flight under this step's chain
      const float* gn_ = g + (size_t)(t + 1) * 3 * HS;
      g0 = gn_[j], g1 = gn_[HS + j], g2 = gn_[2 * HS + j];
    }
    float ar = br, az = bz, an = bn;
#pragma unroll
    for (int k = 0; k < HS; ++k) {
      const float hk = __shfl(h, k, 64);
      ar = fmaf(wr[k], hk, ar);
      az = fmaf(wz[k], hk, az);
      an = fmaf(wn[k], hk, an);
    }
    if (act) {
      const float r = sigmoidf_(c0 + ar);
      const float z = sigmoidf_(c1 + az);
      const float n = tanhf(c2 + r * an);
      h = (1.0f - z) * n + z * h;
      out[((size_t)row * t_len + t) * HS + j] = h;
      float* sv = saved + ((size_t)row * t_len + t) * 4 * HS;
      sv[j] = r, sv[HS + j] = z, sv[2 * HS + j] = n, sv[3 * HS + j] = an;
    }
  }
}

template <int HS>
__global__ __launch_bounds__(64) void gru_seq_bwd_hs_f32(const float* __restrict__ dout, const float* __restrict__ dh_last,
                                                          const float* __restrict__ h0, const float* __restrict__ out,
                                                          const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                          float* __restrict__ dgi, float* __restrict__ dh0,
                                                          float* __restrict__ dw_part, float* __restrict__ db_part, int t_len) {
  const int row = blockIdx.x, j = threadIdx.x;
  const bool act = j < HS;
  const int jc = act ? j : 0;
  // columns jc of the three gate blocks of W_hh (for dh_prev[jc] = sum_i W[g HS + i][jc] g_i) and this lane's rows of dW_hh
  float cr[HS], cz[HS], cn[HS], dwr[HS], dwz[HS], dwn[HS];
#pragma unroll
  for (int i = 0; i < HS; ++i) {
    cr[i] = w_hh[(size_t)i * HS + jc], cz[i] = w_hh[(size_t)(HS + i) * HS + jc], cn[i] = w_hh[(size_t)(2 * HS + i) * HS + jc];
    dwr[i] = 0.f, dwz[i] = 0.f, dwn[i] = 0.f;
  }
  float dh = (act && dh_last) ? dh_last[(size_t)row * HS + j] : 0.f;
  float dbr = 0.f, dbz = 0.f, dbn = 0.f;
  for (int t = t_len - 1; t >= 0; --t) {
    float gr = 0.f, gz = 0.f, gn = 0.f, hprev = 0.f, dh_keep = 0.f;
    if (act) {
      if (dout) dh += dout[((size_t)row * t_len + t) * HS + j];
      const float* sv = saved + ((size_t)row * t_len + t) * 4 * HS;
      const float r = sv[j], z = sv[HS + j], n = sv[2 * HS + j], an = sv[3 * HS + j];
      hprev = t > 0 ? out[((size_t)row * t_len + t - 1) * HS + j] : (h0 ? h0[(size_t)row * HS + j] : 0.f);
      const float dn_pre = dh * (1.0f - z) * (1.0f - n * n);
      const float dz_pre = dh * (hprev - n) * z * (1.0f - z);
      const float dr_pre = dn_pre * an * r * (1.0f - r);
      float* d = dgi + ((size_t)row * t_len + t) * 3 * HS;
      d[j] = dr_pre, d[HS + j] = dz_pre, d[2 * HS + j] = dn_pre;
      gr = dr_pre, gz = dz_pre, gn = dn_pre * r;
      dbr += gr, dbz += gz, dbn += gn;
      dh_keep = dh * z;
    }
    float dh_prev = dh_keep;
#pragma unroll
    for (int k = 0; k < HS; ++k) {
      const float hk = __shfl(hprev, k, 64);
      dwr[k] += gr * hk, dwz[k] += gz * hk, dwn[k] += gn * hk;
      const float grk = __shfl(gr, k, 64), gzk = __shfl(gz, k, 64), gnk = __shfl(gn, k, 64);
      dh_prev += cr[k] * grk + cz[k] * gzk + cn[k] * gnk;
    }
    dh = act ? dh_prev : 0.f;
  }
  if (act) {
    if (dh0) dh0[(size_t)row * HS + j] = dh;
    float* dbp = db_part + (size_t)row * 3 * HS;
    dbp[j] = dbr, dbp[HS + j] = dbz, dbp[2 * HS + j] = dbn;
    float* dwp = dw_part + (size_t)row * 3 * HS * HS;
#pragma unroll
    for (int k = 0; k < HS; ++k)
      dwp[(size_t)j * HS + k] = dwr[k], dwp[(size_t)(HS + j) * HS + k] = dwz[k], dwp[(size_t)(2 * HS + j) * HS + k] = dwn[k];
  }
}

__global__ __launch_bounds__(256) void gru_sum_rows_f32(const float* __restrict__ part, float* __restrict__ out, int n, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += part[(size_t)r * n + i];
  out[i] = s;
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_gru_seq_fwd_f32(const float* gi, const float* h0, const float* w_hh, const float* b_hh, float* out, float* saved,
                       int32_t batch, int32_t t_len, int32_t hidden, void* stream) {
  PV_REQUIRE(gi && w_hh && b_hh && out && saved, PV_EINVAL, "pv_gru_seq_fwd_f32: null pointer");
  PV_REQUIRE(batch > 0 && t_len > 0 && hidden > 0 && hidden <= GRU_MAXH, PV_ESIZE,
             "pv_gru_seq_fwd_f32: bad sizes (hidden must be <= %d)", GRU_MAXH);
  if (hidden == 16 && !getenv("PV_GRU_GENERAL"))
    hipLaunchKernelGGL(gru_seq_fwd_hs_f32<16>, dim3((unsigned)batch), dim3(64), 0, as_stream(stream), gi, h0, w_hh, b_hh, out,
                       saved, t_len);
  else
  hipLaunchKernelGGL(gru_seq_fwd_f32, dim3((unsigned)batch), dim3(64), 0, as_stream(stream), gi, h0, w_hh, b_hh, out, saved,
                     t_len, hidden);
  return check_launch("pv_gru_seq_fwd_f32");
}

int pv_gru_seq_bwd_f32(const float* dout, const float* dh_last, const float* h0, const float* out, const float* saved,
                       const float* w_hh, float* dgi, float* dh0, float* dw_hh, float* db_hh, int32_t batch, int32_t t_len,
                       int32_t hidden, void* ws, size_t ws_bytes, void* stream) {
  PV_REQUIRE(out && saved && w_hh && dgi && dw_hh && db_hh, PV_EINVAL, "pv_gru_seq_bwd_f32: null pointer");
  PV_REQUIRE(batch > 0 && t_len > 0 && hidden > 0 && hidden <= GRU_MAXH, PV_ESIZE,
             "pv_gru_seq_bwd_f32: bad sizes (hidden must be <= %d)", GRU_MAXH);
  const size_t nw = (size_t)3 * hidden * hidden, nb = (size_t)3 * hidden;
  PV_REQUIRE(ws && ws_bytes >= (size_t)batch * (nw + nb) * sizeof(float), PV_EINVAL, "pv_gru_seq_bwd_f32: workspace too small");
  float* dw_part = (float*)ws;
  float* db_part = dw_part + (size_t)batch * nw;
  hipStream_t st = as_stream(stream);
  if (hidden == 16 && !getenv("PV_GRU_GENERAL"))
    hipLaunchKernelGGL(gru_seq_bwd_hs_f32<16>, dim3((unsigned)batch), dim3(64), 0, st, dout, dh_last, h0, out, saved, w_hh, dgi,
                       dh0, dw_part, db_part, t_len);
  else
  hipLaunchKernelGGL(gru_seq_bwd_f32, dim3((unsigned)batch), dim3(64), 0, st, dout, dh_last, h0, out, saved, w_hh, dgi, dh0,
                     dw_part, db_part, t_len, hidden);
  hipLaunchKernelGGL(gru_sum_rows_f32, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, dw_part, dw_hh, (int)nw, batch);
  hipLaunchKernelGGL(gru_sum_rows_f32, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, db_part, db_hh, (int)nb, batch);
  return check_launch("pv_gru_seq_bwd_f32");
}

}  // extern "C"
