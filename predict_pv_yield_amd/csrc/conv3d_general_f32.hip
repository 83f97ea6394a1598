// General Conv3D (any kernel extent up to 3x3x3, any stride, any symmetric padding) in exact fp32 on the reference
// layout NCDHW, plus MaxPool3d and the mean-squared-error loss.
// replaces: the nn.Conv3d stack of the optical-flow notebook model
//   notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:969-1027 (LitAutoEncoder: kernel (2,3,3), padding (0,1,1),
//   last layer stride (1,2,2); F.mse_loss; Adam lr 1e-4)
// and nn.MaxPool3d(3, stride=(1,2,2), padding=1) of Conv3dMaxPool
//   predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:42-57.
// Direct convolution on the f32 VALU, one FMA chain per output in (ci, kt, kh, kw) order; weights broadcast from LDS.
#include "pv_common.h"

namespace pv {

constexpr int GCO = 16;  // channels per thread (fwd: output channels, dgrad: input channels)

struct Geom {
  int c_in, c_out, t_in, h_in, w_in, t_out, h_out, w_out;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
};

// One thread = one output voxel x GCO output channels.  KT/KH/KW compile-time (0 = runtime extents from g).
template <int KT, int KH, int KW>
__global__ __launch_bounds__(256) void conv3d_general_fwd_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              Geom g, int relu) {
  extern __shared__ float wl[];  // [c_in][taps][GCO]
  const int kt_n = KT ? KT : g.kt, kh_n = KH ? KH : g.kh, kw_n = KW ? KW : g.kw;
  const int taps = kt_n * kh_n * kw_n;
  const int co0 = blockIdx.y * GCO;
  const int b = blockIdx.z;
  for (int i = threadIdx.x; i < g.c_in * taps * GCO; i += blockDim.x) {
    int j = i % GCO;
    int tap = (i / GCO) % taps;
    int ci = i / (GCO * taps);
    int co = co0 + j;
    wl[i] = co < g.c_out ? w[((size_t)co * g.c_in + ci) * taps + tap] : 0.f;
  }
  __syncthreads();
  const int plane_out = g.h_out * g.w_out;
  const int vox_out = g.t_out * plane_out;
  const size_t plane_in = (size_t)g.h_in * g.w_in;
  const size_t vox_in = (size_t)g.t_in * plane_in;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vox_out; v += gridDim.x * blockDim.x) {
    int to = v / plane_out;
    int r = v - to * plane_out;
    int ho = r / g.w_out;
    int wo = r - ho * g.w_out;
    float acc[GCO];
#pragma unroll
    for (int j = 0; j < GCO; ++j) acc[j] = (bias && co0 + j < g.c_out) ? bias[co0 + j] : 0.f;
    const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
    for (int ci = 0; ci < g.c_in; ++ci) {
      const float* xc = x + ((size_t)b * g.c_in + ci) * vox_in;
      const float* wc = wl + (size_t)ci * taps * GCO;
#pragma unroll
      for (int kt = 0; kt < kt_n; ++kt) {
        int ti = t0 + kt;
        bool t_ok = (unsigned)ti < (unsigned)g.t_in;
#pragma unroll
        for (int kh = 0; kh < kh_n; ++kh) {
          int hi = h0 + kh;
          bool h_ok = t_ok && (unsigned)hi < (unsigned)g.h_in;
#pragma unroll
          for (int kw = 0; kw < kw_n; ++kw) {
            int wi = w0 + kw;
            // unconditional load from a clamped address + select: a load inside an `if` is waited for at the join of
            // its branch, one memory latency per tap instead of one per batch of taps
            const bool ok = h_ok && (unsigned)wi < (unsigned)g.w_in;
            float xv = xc[ok ? (size_t)ti * plane_in + (size_t)hi * g.w_in + wi : 0];
            xv = ok ? xv : 0.f;
            const float* wt = wc + ((kt * kh_n + kh) * kw_n + kw) * GCO;
#pragma unroll
            for (int j = 0; j < GCO; ++j) acc[j] = fmaf(xv, wt[j], acc[j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < GCO; ++j) {
      if (co0 + j < g.c_out) {
        float o = acc[j];
        if (relu) o = o > 0.f ? o : 0.f;
        y[((size_t)b * g.c_out + co0 + j) * vox_out + v] = o;
      }
    }
  }
}

// Unit-stride fast path (forward, and dgrad as a correlation with mirrored, channel-swapped weights): one thread =
// TW consecutive output columns x GCO channels, so each LDS weight read feeds TW FMAs and each input value KW of them.
// Per-output summation order is still (channel, kt, kh, kw) ascending.
constexpr int TW = 8;

template <int KT, int KH, int KW, bool GATE>
__global__ __launch_bounds__(256) void conv3d_tiled_s1_f32(const float* __restrict__ x, const float* __restrict__ gate,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, Geom g, int relu, int flip,
                                                           int w_ci_dim, const float* __restrict__ out_gate) {
  extern __shared__ float wl[];  // [c_in][taps][GCO]
  constexpr int TAPS = KT * KH * KW;
  const int co0 = blockIdx.y * GCO;
  const int b = blockIdx.z;
  for (int i = threadIdx.x; i < g.c_in * TAPS * GCO; i += blockDim.x) {
    int j = i % GCO;
    int tap = (i / GCO) % TAPS;
    int ci = i / (GCO * TAPS);
    int co = co0 + j;
    float v = 0.f;
    if (co < g.c_out)
      v = flip ? w[((size_t)ci * w_ci_dim + co) * TAPS + (TAPS - 1 - tap)] : w[((size_t)co * w_ci_dim + ci) * TAPS + tap];
    wl[i] = v;
  }
  __syncthreads();
  const int n_wt = (g.w_out + TW - 1) / TW;
  const int tiles = g.t_out * g.h_out * n_wt;
  const int plane_out = g.h_out * g.w_out;
  const int vox_out = g.t_out * plane_out;
  const int plane_in = g.h_in * g.w_in;
  const size_t vox_in = (size_t)g.t_in * plane_in;
  for (int tile = blockIdx.x * blockDim.x + threadIdx.x; tile < tiles; tile += gridDim.x * blockDim.x) {
    const int wt = tile % n_wt;
    const int r = tile / n_wt;
    const int ho = r % g.h_out;
    const int to = r / g.h_out;
    const int wo0 = wt * TW;
    float acc[TW][GCO];
#pragma unroll
    for (int j = 0; j < GCO; ++j) {
      const float bv = (bias && co0 + j < g.c_out) ? bias[co0 + j] : 0.f;
#pragma unroll
      for (int v = 0; v < TW; ++v) acc[v][j] = bv;
    }
    const int wi0 = wo0 - g.pw;
    for (int ci = 0; ci < g.c_in; ++ci) {
      const float* xc = x + ((size_t)b * g.c_in + ci) * vox_in;
      const float* gc = GATE ? gate + ((size_t)b * g.c_in + ci) * vox_in : nullptr;
      const float* wc = wl + (size_t)ci * TAPS * GCO;
      // one (kt, kh) input row at a time: keeps only KW x GCO weights live (a full unroll hoists every tap's weights)
#pragma unroll 1
      for (int row = 0; row < KT * KH; ++row) {
        const int kt = row / KH, kh = row - kt * KH;
        const int ti = to + kt - g.pt;
        const int hi = ho + kh - g.ph;
        const bool ok = (unsigned)ti < (unsigned)g.t_in && (unsigned)hi < (unsigned)g.h_in;
        const int base = ok ? ti * plane_in + hi * g.w_in : 0;
        float xr[TW + KW - 1];
#pragma unroll
        for (int i = 0; i < TW + KW - 1; ++i) {
          const int wi = wi0 + i;
          float xv = 0.f;
          if (ok && (unsigned)wi < (unsigned)g.w_in) {
            xv = xc[base + wi];
            if (GATE && !(gc[base + wi] > 0.f)) xv = 0.f;
          }
          xr[i] = xv;
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const float4* wt4 = reinterpret_cast<const float4*>(wc + (row * KW + kw) * GCO);
          float wv[GCO];
#pragma unroll
          for (int q = 0; q < GCO / 4; ++q) {
            const float4 t4 = wt4[q];
            wv[4 * q] = t4.x, wv[4 * q + 1] = t4.y, wv[4 * q + 2] = t4.z, wv[4 * q + 3] = t4.w;
          }
#pragma unroll
          for (int v = 0; v < TW; ++v)
#pragma unroll
            for (int j = 0; j < GCO; ++j) acc[v][j] = fmaf(xr[v + kw], wv[j], acc[v][j]);
        }
      }
    }
    const size_t out0 = (size_t)to * plane_out + (size_t)ho * g.w_out + wo0;
    const bool full = wo0 + TW <= g.w_out && (g.w_out & 3) == 0;
#pragma unroll
    for (int j = 0; j < GCO; ++j) {
      if (co0 + j >= g.c_out) continue;
      float* yo = y + ((size_t)b * g.c_out + co0 + j) * vox_out + out0;
      float o[TW];
#pragma unroll
      for (int v = 0; v < TW; ++v) o[v] = relu ? fmaxf(acc[v][j], 0.f) : acc[v][j];
      if (out_gate) {  // dgrad into a ReLU output: zero where that output was not positive
        const float* og = out_gate + ((size_t)b * g.c_out + co0 + j) * vox_out + out0;
#pragma unroll
        for (int v = 0; v < TW; ++v)
          if (wo0 + v < g.w_out && !(og[v] > 0.f)) o[v] = 0.f;
      }
      if (full) {
#pragma unroll
        for (int q = 0; q < TW / 4; ++q)
          reinterpret_cast<float4*>(yo)[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
      } else {
#pragma unroll
        for (int v = 0; v < TW; ++v)
          if (wo0 + v < g.w_out) yo[v] = o[v];
      }
    }
  }
}

typedef float v16f __attribute__((ext_vector_type(16)));

// Unit-stride forward / dgrad on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation)
// for 16 <= c_in <= 32, c_out <= 32 and the notebook model's (2,3,3) layers:
//   D[co][pos] = sum_{tap, ci} W[co][ci][tap] * X[ci][pos + tap],   one 32 x 32 accumulator per wave.
// Workgroup = 4 waves (one per SIMD, so a wave may use the whole register file) = 2 position tiles (64 output columns
// of one output row) x 2 halves of the input channels; a wave keeps the weights of its channel half as MFMA A operands
// in registers (TAPS x 8 VGPRs = 144 for (2,3,3)) for the whole launch, the B operand (2 channels x 32 positions) is a
// ds_read_b32 from the staged input rows, fetched one (kt, kh) row ahead of its MFMAs.  The workgroup marches down a
// segment of output rows: the KT x (KH + 1) input rows live in an LDS ring (channel stride = 32 mod 64 words: the two
// channels of an operand read hit disjoint banks) filled global -> LDS directly, each step fetching only the KT new
// rows while the MFMAs of the current row run.  The two channel halves are added through LDS (fixed order), then
// bias / ReLU / out_gate and 128-byte row stores.  Same flip / out_gate conventions as conv3d_tiled_s1_f32.
template <int KT, int KH, int KW, bool M16>
struct FwdMfmaCfg {
  static constexpr int TAPS = KT * KH * KW;
  static constexpr int SLOTS = KH + 1;                            // ring slots per kt: KH rows in use + the one in flight
  static constexpr int WT = 64;
  static constexpr int R = WT + KW - 1;
  // channel stride in words: = 32 (mod 64) for the 32x32x2 operand (2 channels x 32 positions per read),
  //                          = 16 (mod 64) for the 16x16x4 operand (4 channels x 16 positions)
  static constexpr int CS = M16 ? ((KT * SLOTS * R + 47) / 64) * 64 + 16 : ((KT * SLOTS * R + 31) / 64) * 64 + 32;
};

// SPLIT_CI: the two wave pairs split the contraction by input-channel half (c_in > 16); otherwise (c_in <= 16, all
// channel pairs fit one half) by tap plane kt, so that neither pair multiplies zero padding.
// M16 (c_out <= 16, SPLIT_CI): v_mfma_f32_16x16x4_f32 instead -- 16 output channels x 16 positions, 4 channels per
// step; a wave's 32 positions are two such tiles, so no half of the matrix tile multiplies zero rows.
template <int KT, int KH, int KW, bool SPLIT_CI, bool M16 = false>
__global__ __launch_bounds__(256, 1) void conv3d_fwd_mfma_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, Geom g,
                                                           int relu, int flip, int w_ci_dim,
                                                           const float* __restrict__ out_gate, int hseg, int n_hseg) {
  using C = FwdMfmaCfg<KT, KH, KW, M16>;
  static_assert(!M16 || SPLIT_CI, "the 16-row variant splits the contraction by channel half");
  constexpr int TAPS = C::TAPS, SLOTS = C::SLOTS, WT = C::WT, R = C::R, CS = C::CS;
  __shared__ __attribute__((aligned(16))) float xT[32 * CS];
  __shared__ float red[2 * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pt = wave & 1, kk = wave >> 1, h = lane >> 5, l32 = lane & 31;
  const int kq = lane >> 4, l16 = lane & 15;   // M16 roles: channel of a 4-channel step, row / column of a 16 x 16 tile
  const int n_wt = (g.w_out + WT - 1) / WT;
  int item = blockIdx.x;
  const int hs = item % n_hseg;
  item /= n_hseg;
  const int wt = item % n_wt;
  item /= n_wt;
  const int to = item % g.t_out;
  const int b = item / g.t_out;
  const int ho0 = hs * hseg, ho1 = min(ho0 + hseg, g.h_out);
  const int wo0 = wt * WT;
  const int plane_in = g.h_in * g.w_in, plane_out = g.h_out * g.w_out;
  const size_t vox_in = (size_t)g.t_in * plane_in, vox_out = (size_t)g.t_out * plane_out;

  // ---- weights of this wave's channel half: A operand of tap t, channel pair p = W[co = l32][ci = 16kk + 2p + h][t] -----
  constexpr int NT = SPLIT_CI ? TAPS : KH * KW;   // taps this wave multiplies (tap-plane split: only plane kt = kk)
  constexpr int NP = M16 ? 4 : 8;                 // contraction steps per tap: 4-channel steps / channel pairs
  float areg[NT][NP];
#pragma unroll
  for (int tl = 0; tl < NT; ++tl)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int t = SPLIT_CI ? tl : kk * KH * KW + tl;
      const int ci = M16 ? 16 * kk + 4 * p + kq : (SPLIT_CI ? 16 * kk : 0) + 2 * p + h;
      const int co = M16 ? l16 : l32;
      float v = 0.f;
      if (co < g.c_out && ci < g.c_in)
        v = flip ? w[((size_t)ci * w_ci_dim + co) * TAPS + (TAPS - 1 - t)] : w[((size_t)co * w_ci_dim + ci) * TAPS + t];
      areg[tl][p] = v;
    }
  static_assert(SPLIT_CI || KT == 2, "the tap-plane split deals one kt plane to each wave pair");

  // ---- staging, global -> LDS direct (buffer_load_dword ... lds): one wave instruction = 64 consecutive columns of one
  // (channel, input row); a row of R = WT + KW - 1 columns = one full instruction + one with KW - 1 lanes.  Wave w stages
  // channels 8w .. 8w+7.  Out-of-range positions (padding, image edge, channels >= c_in) read as zeros. ----------------
  constexpr uint32_t INVALID = 0x40000000u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * g.c_in * vox_in), 0,
                                                                        (int)(g.c_in * vox_in * 4), 0x00020000);
  const bool gated = out_gate != nullptr;
  const __amdgpu_buffer_rsrc_t grs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(gated ? out_gate + (size_t)b * g.c_out * vox_out : y), 0, gated ? (int)(g.c_out * vox_out * 4) : 0, 0x00020000);
  uint32_t col_off[2];   // byte offset of this lane's column inside an input row, per 64-column segment
#pragma unroll
  for (int sgm = 0; sgm < 2; ++sgm) {
    const int wi = wo0 - g.pw + 64 * sgm + lane;
    col_off[sgm] = (unsigned)wi < (unsigned)g.w_in ? (uint32_t)wi * 4u : INVALID;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto stage_chan = [&](int kt, int c8, int hi) {   // input row (to + kt - pt, hi) of channel 8w + c8 -> slot hi mod SLOTS
    const int ti = to + kt - g.pt;
    const bool ok = (unsigned)ti < (unsigned)g.t_in && (unsigned)hi < (unsigned)g.h_in;
    const uint32_t ro = ok ? (uint32_t)((ti * plane_in + hi * g.w_in) * 4) : INVALID;
    const int slot = ((hi % SLOTS) + SLOTS) % SLOTS;
    const int ci = 8 * wave + c8;
    const uint32_t co = ci < g.c_in ? (uint32_t)((size_t)ci * vox_in * 4) + ro : INVALID;
    float* dst = xT + ci * CS + (kt * SLOTS + slot) * R;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)dst, 4, col_off[0] + co, 0, 0, 0);
    if (lane < R - 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + 64), 4, col_off[1] + co, 0, 0, 0);
  };
  auto stage_row = [&](int kt, int hi) {
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) stage_chan(kt, c8, hi);
  };

  // ---- prologue: the KH input rows of the segment's first output row -----------------------------------------------
#pragma unroll 1
  for (int kh = 0; kh < KH; ++kh)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) stage_row(kt, ho0 - g.ph + kh);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();

  const int b_lane = M16 ? (16 * kk + kq) * CS + 32 * pt + l16 : ((SPLIT_CI ? 16 * kk : 0) + h) * CS + 32 * pt + l32;
  // epilogue roles: (M16) this wave finishes N tile kk: channel 4 kq + r, position 32 pt + 16 kk + l16, r < 4;
  //                 (else) registers 8 kk .. 8 kk + 7 of the 32 x 32 tile: channel acc_row, position 32 pt + l32
  constexpr int NE = M16 ? 4 : 8;
  const int pos = wo0 + 32 * pt + (M16 ? 16 * kk + l16 : l32);
  int e_co[NE];
  float bias_r[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int r = 8 * kk + e;
    e_co[e] = M16 ? 4 * kq + e : (r & 3) + 8 * (r >> 2) + 4 * h;
    bias_r[e] = (bias && e_co[e] < g.c_out) ? bias[e_co[e]] : 0.f;
  }
  for (int ho = ho0; ho < ho1; ++ho) {
    const bool more = ho + 1 < ho1;
    // out_gate values of this wave's outputs, fetched now so that their latency hides under the MFMAs.  Branch-free raw
    // buffer loads (no gate: a zero-sized resource, every load returns 0 and `gated` is false): a conditional load would
    // be waited for at the join of its branch, eight serialized memory latencies per step (measured +6.5k cycles).
    float og[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const bool ok = e_co[e] < g.c_out && pos < g.w_out;
      const uint32_t off = ok ? (uint32_t)(((size_t)e_co[e] * vox_out + (size_t)to * plane_out + (size_t)ho * g.w_out + pos) * 4)
                              : INVALID;
      og[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs_out, off, 0, 0));
    }
    constexpr int NR = SPLIT_CI ? KT * KH : KH;           // (kt, kh) rows this wave multiplies
    const int r0 = SPLIT_CI ? 0 : kk * KH;                // first of them
    // B operands of one (kt, kh) input row (24 values) are read while the previous row's MFMAs run; the staging of the
    // next step's new input rows (16 LDS-direct instructions per kt and wave, ~50 issue cycles each) is spread over the
    // row loop as well.  Two accumulation chains per tile: a dependent MFMA cannot start before its predecessor has left
    // the pipe, and with one wave per SIMD nothing else would fill that gap.
    auto row_base = [&](int rr) {
      const int kt = rr / KH, kh = rr - kt * KH;
      const int hi = ho - g.ph + kh;
      const int slot = ((hi % SLOTS) + SLOTS) % SLOTS;
      return xT + b_lane + (kt * SLOTS + slot) * R;
    };
    auto stage_share = [&](int i) {
      if (more) {
        constexpr int PER = (KT * 8 + NR - 1) / NR;
#pragma unroll
        for (int q = i * PER; q < (i + 1) * PER && q < KT * 8; ++q) stage_chan(q / 8, q % 8, ho + KH - g.ph);
      }
    };
    float outv[NE];   // this wave's finished values
    if constexpr (M16) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      v4f a0[2], a1[2];   // [N tile], two chains
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) a0[nt][r] = 0.f, a1[nt][r] = 0.f;
      float bv[2][4][KW][2];
      auto read_row = [&](int rr, float (&dstv)[4][KW][2]) {
        const float* rowp = row_base(rr);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int kw = 0; kw < KW; ++kw) dstv[p][kw][0] = rowp[4 * p * CS + kw], dstv[p][kw][1] = rowp[4 * p * CS + kw + 16];
      };
      read_row(r0, bv[0]);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if (i + 1 < NR) read_row(r0 + i + 1, bv[(i + 1) & 1]);
        stage_share(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
          for (int p = 0; p < 4; p += 2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              a0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i * KW + kw][p], bv[i & 1][p][kw][nt], a0[nt], 0, 0, 0);
              a1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i * KW + kw][p + 1], bv[i & 1][p + 1][kw][nt], a1[nt], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      // each wave pair parks the N tile the OTHER pair finishes, then adds the partner's copy of its own tile
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(pt * 8 + 4 * (1 - kk) + r) * 64 + lane] = a0[1 - kk][r] + a1[1 - kk][r];
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's part of the next row has landed
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float mine = a0[kk][r] + a1[kk][r], other = red[(pt * 8 + 4 * kk + r) * 64 + lane];
        outv[r] = kk == 0 ? mine + other : other + mine;   // always (lower pair's partial) + (upper pair's partial)
      }
    } else {
      v16f acc, acc_b;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc_b[r] = 0.f;
      float bv[2][8][KW];
      auto read_row = [&](int rr, float (&dstv)[8][KW]) {
        const float* rowp = row_base(rr);
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
          for (int kw = 0; kw < KW; ++kw) dstv[p][kw] = rowp[2 * p * CS + kw];
      };
      read_row(r0, bv[0]);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if (i + 1 < NR) read_row(r0 + i + 1, bv[(i + 1) & 1]);
        stage_share(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
          for (int p = 0; p < 8; p += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[i * KW + kw][p], bv[i & 1][p][kw], acc, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[i * KW + kw][p + 1], bv[i & 1][p + 1][kw], acc_b, 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += acc_b[r];
      // each wave pair parks the half of its tile that the OTHER pair finishes (registers 8kk' .. 8kk'+7, kk' = 1 - kk)
#pragma unroll
      for (int r = 0; r < 8; ++r) red[(pt * 16 + 8 * (1 - kk) + r) * 64 + lane] = kk == 0 ? acc[8 + r] : acc[r];
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's part of the next row has landed
      __syncthreads();
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const float mine = kk == 0 ? acc[r8] : acc[8 + r8];
        const float other = red[(pt * 16 + 8 * kk + r8) * 64 + lane];
        outv[r8] = kk == 0 ? mine + other : other + mine;   // always (lower pair's partial) + (upper pair's partial)
      }
    }
    // ---- epilogue: bias, ReLU, out_gate, 64 / 128-byte row stores -------------------------------------------------------
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (e_co[e] < g.c_out && pos < g.w_out) {
        float v = outv[e] + bias_r[e];
        if (relu) v = fmaxf(v, 0.f);
        if (gated && !(og[e] > 0.f)) v = 0.f;
        y[((size_t)b * g.c_out + e_co[e]) * vox_out + (size_t)to * plane_out + (size_t)ho * g.w_out + pos] = v;
      }
    }
    __syncthreads();   // red is free again before the next step's upper half writes it
  }
}

// dgrad, gather form: dx[b,ci,ti,hi,wi] = sum_{co,taps : to*st - pt + kt == ti ...} dy'[b,co,to,ho,wo] * w[co,ci,tap],
// dy' = dy gated by (gate > 0) when the forward was followed by a ReLU.  One thread = one dx voxel x GCO input channels.
template <int KT, int KH, int KW>
__global__ __launch_bounds__(256) void conv3d_general_dgrad_f32(const float* __restrict__ dy,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ w, float* __restrict__ dx,
                                                                Geom g, const float* __restrict__ out_gate) {
  extern __shared__ float wl[];  // [c_out][taps][GCO]
  const int kt_n = KT ? KT : g.kt, kh_n = KH ? KH : g.kh, kw_n = KW ? KW : g.kw;
  const int taps = kt_n * kh_n * kw_n;
  const int ci0 = blockIdx.y * GCO;
  const int b = blockIdx.z;
  for (int i = threadIdx.x; i < g.c_out * taps * GCO; i += blockDim.x) {
    int j = i % GCO;
    int tap = (i / GCO) % taps;
    int co = i / (GCO * taps);
    int ci = ci0 + j;
    wl[i] = ci < g.c_in ? w[((size_t)co * g.c_in + ci) * taps + tap] : 0.f;
  }
  __syncthreads();
  const int plane_in = g.h_in * g.w_in;
  const int vox_in = g.t_in * plane_in;
  const size_t plane_out = (size_t)g.h_out * g.w_out;
  const size_t vox_out = (size_t)g.t_out * plane_out;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vox_in; v += gridDim.x * blockDim.x) {
    int ti = v / plane_in;
    int r = v - ti * plane_in;
    int hi = r / g.w_in;
    int wi = r - hi * g.w_in;
    float acc[GCO];
#pragma unroll
    for (int j = 0; j < GCO; ++j) acc[j] = 0.f;
    for (int co = 0; co < g.c_out; ++co) {
      const float* dc = dy + ((size_t)b * g.c_out + co) * vox_out;
      const bool gated = gate != nullptr;
      const float* gc = gated ? gate + ((size_t)b * g.c_out + co) * vox_out : dc;
      const float* wc = wl + (size_t)co * taps * GCO;
#pragma unroll
      for (int kt = 0; kt < kt_n; ++kt) {
        int nt = ti + g.pt - kt;
        int to = nt / g.st;
        bool t_ok = nt >= 0 && to * g.st == nt && to < g.t_out;
#pragma unroll
        for (int kh = 0; kh < kh_n; ++kh) {
          int nh = hi + g.ph - kh;
          int ho = nh / g.sh;
          bool h_ok = t_ok && nh >= 0 && ho * g.sh == nh && ho < g.h_out;
#pragma unroll
          for (int kw = 0; kw < kw_n; ++kw) {
            int nw = wi + g.pw - kw;
            int wo = nw / g.sw;
            // unconditional loads from a clamped address + selects (see conv3d_general_fwd_f32); without a gate the
            // "gate" pointer is dy itself and its value is ignored
            const bool ok = h_ok && nw >= 0 && wo * g.sw == nw && wo < g.w_out;
            const size_t off = ok ? (size_t)to * plane_out + (size_t)ho * g.w_out + wo : 0;
            float d = dc[off];
            const float gv = gc[off];
            d = (ok && !(gated && !(gv > 0.f))) ? d : 0.f;
            const float* wt = wc + ((kt * kh_n + kh) * kw_n + kw) * GCO;
#pragma unroll
            for (int j = 0; j < GCO; ++j) acc[j] = fmaf(d, wt[j], acc[j]);
          }
        }
      }
    }
    // the out_gate values of the GCO channels are fetched back to back (no gate: dx's own, ignored, contents)
    const bool og_on = out_gate != nullptr;
    const float* ogp = og_on ? out_gate : dx;
    float ogv[GCO];
#pragma unroll
    for (int j = 0; j < GCO; ++j) {
      const int cj = min(ci0 + j, g.c_in - 1);
      ogv[j] = ogp[((size_t)b * g.c_in + cj) * vox_in + v];
    }
#pragma unroll
    for (int j = 0; j < GCO; ++j)
      if (ci0 + j < g.c_in) {
        const size_t off = ((size_t)b * g.c_in + ci0 + j) * vox_in + v;
        dx[off] = (og_on && !(ogv[j] > 0.f)) ? 0.f : acc[j];
      }
  }
}

// wgrad: block = (ci, group of WCO output channels, slab of positions); thread-private taps x WCO partial sums over
// PW consecutive output columns per iteration (PW = 2 needs unit stride: neighbouring outputs share input columns), then
// a deterministic block reduction into slabs[slab][co][ci][tap]; slab_reduce adds the slabs in index order.  The ci == 0
// blocks also reduce dy itself into bias_slabs[slab][co] (dbias costs no extra pass over dy).
template <int KT, int KH, int KW, int WCO, int PW>
__global__ __launch_bounds__(256) void conv3d_general_wgrad_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ gate, float* __restrict__ slabs,
                                                                float* __restrict__ bias_slabs, Geom g, int batch,
                                                                int n_slabs) {
  constexpr int TAPS = KT * KH * KW;
  const int ci = blockIdx.x;
  const int co0 = blockIdx.y * WCO;
  const int slab = blockIdx.z;
  const int n_wp = (g.w_out + PW - 1) / PW;
  const int plane_out = g.h_out * g.w_out;
  const int vox_out = g.t_out * plane_out;
  const int plane_in = g.h_in * g.w_in;
  const size_t vox_in = (size_t)g.t_in * plane_in;
  float acc[WCO][TAPS];
  float bsum[WCO];
#pragma unroll
  for (int j = 0; j < WCO; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[j][k] = 0.f;
  }
  const long long total = (long long)batch * g.t_out * g.h_out * n_wp;
  const long long per = (total + n_slabs - 1) / n_slabs;
  const long long i0 = (long long)slab * per;
  const long long i1 = i0 + per < total ? i0 + per : total;
  for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const int wp = (int)(i % n_wp);
    long long r = i / n_wp;
    const int ho = (int)(r % g.h_out);
    r /= g.h_out;
    const int to = (int)(r % g.t_out);
    const int b = (int)(r / g.t_out);
    const int wo0 = wp * PW;
    const int v0 = to * plane_out + ho * g.w_out + wo0;
    float d[WCO][PW];
#pragma unroll
    for (int j = 0; j < WCO; ++j)
#pragma unroll
      for (int u = 0; u < PW; ++u) {
        float dv = 0.f;
        if (co0 + j < g.c_out && wo0 + u < g.w_out) {
          const size_t off = ((size_t)b * g.c_out + co0 + j) * vox_out + v0 + u;
          dv = dy[off];
          if (gate && !(gate[off] > 0.f)) dv = 0.f;
        }
        d[j][u] = dv;
        bsum[j] += dv;
      }
    const float* xc = x + ((size_t)b * g.c_in + ci) * vox_in;
    const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph, w0 = wo0 * g.sw - g.pw;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int ti = t0 + kt;
      const bool t_ok = (unsigned)ti < (unsigned)g.t_in;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const int hi = h0 + kh;
        const bool ok = t_ok && (unsigned)hi < (unsigned)g.h_in;
        const int base = ok ? ti * plane_in + hi * g.w_in : 0;
        float xr[KW + PW - 1];
#pragma unroll
        for (int q = 0; q < KW + PW - 1; ++q) {
          const int wi = w0 + q;
          xr[q] = (ok && (unsigned)wi < (unsigned)g.w_in) ? xc[base + wi] : 0.f;
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
          for (int u = 0; u < PW; ++u)
#pragma unroll
            for (int j = 0; j < WCO; ++j)
              acc[j][(kt * KH + kh) * KW + kw] = fmaf(d[j][u], xr[kw + u], acc[j][(kt * KH + kh) * KW + kw]);
      }
    }
  }
  __shared__ float red[4][WCO * (TAPS + 1)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < WCO; ++j) {
#pragma unroll
    for (int k = 0; k <= TAPS; ++k) {
      float v = k < TAPS ? acc[j][k < TAPS ? k : 0] : bsum[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][j * (TAPS + 1) + k] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < WCO * (TAPS + 1)) {
    const int j = threadIdx.x / (TAPS + 1), k = threadIdx.x % (TAPS + 1);
    if (co0 + j < g.c_out) {
      const float sum = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
      if (k < TAPS) slabs[(((size_t)slab * g.c_out + co0 + j) * g.c_in + ci) * TAPS + k] = sum;
      else if (ci == 0 && bias_slabs) bias_slabs[(size_t)slab * g.c_out + co0 + j] = sum;
    }
  }
}

// wgrad on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation) for unit stride and
// 16 < channels <= 32: D[co][ci] += sum_p dy[co][p] * x[ci][p + tap offset], one 32x32 accumulator tile per tap.
// Block = 8 waves; wave (w & 3) owns taps w&3, (w&3)+4, ...; waves 0-3 take the first half of the row tile, 4-7 the
// second half (combined through LDS in fixed order at the end).  An extra "ones" tap yields dbias = sum_p dy[co][p].
// A step = one (b, to, ho, w-tile): dy row [32][WT] and the KT*KH input rows [32][ROWS][WT+KW-1] staged in LDS with
// channel strides = 2 (mod 64) words, so the 32-channel x 2-position operand reads are bank-conflict free.
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KT, int KH, int KW>
struct WgradMfmaCfg {
  static constexpr int TAPS = KT * KH * KW;
  static constexpr int ROWS = KT * KH;
  static constexpr int WT = ROWS > 6 ? 64 : 128;          // output columns per step
  static constexpr int R = WT + 2;                         // staged row length (>= WT + KW - 1)
  static constexpr int S = WT + 2;                         // dy channel stride
  static constexpr int CS = ((ROWS * R + 63) / 64) * 64 + 2;  // x channel stride
  static constexpr int NSLOT = (TAPS + 1 + 3) / 4;         // taps (+ ones tap) per wave
};

template <int KT, int KH, int KW>
__global__ __launch_bounds__(512) void conv3d_wgrad_mfma_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ gate, float* __restrict__ slabs,
                                                             float* __restrict__ bias_slabs, Geom g, int batch,
                                                             int n_steps, int steps_per_block) {
  using C = WgradMfmaCfg<KT, KH, KW>;
  constexpr int TAPS = C::TAPS, ROWS = C::ROWS, WT = C::WT, R = C::R, S = C::S, CS = C::CS, NSLOT = C::NSLOT;
  __shared__ float dyT[32 * S];
  // xT is reused at the end to combine the two half-row wave groups (4 waves x NSLOT tiles x 16 x 64 floats)
  constexpr int XT_ELEMS = 32 * CS > 4 * NSLOT * 16 * 64 ? 32 * CS : 4 * NSLOT * 16 * 64;
  __shared__ float xT[XT_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w4 = wave & 3, half = wave >> 2;
  const int n_wt = (g.w_out + WT - 1) / WT;
  const int plane_out = g.h_out * g.w_out;
  const size_t vox_out = (size_t)g.t_out * plane_out;
  const int plane_in = g.h_in * g.w_in;
  const size_t vox_in = (size_t)g.t_in * plane_in;

  v16f acc[NSLOT];
  int off[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tap = w4 + 4 * i;
    off[i] = tap < TAPS ? (tap / KW) * R + (tap % KW) : -1;   // -1: ones tap (tap == TAPS) or unused (tap > TAPS)
  }
  const int a_base = (lane & 31) * S + (lane >> 5);
  const int b_base = (lane & 31) * CS + (lane >> 5);

  // staging registers: every wave moves 4 dy rows and 4*ROWS input rows per step (lanes along w); the global loads of
  // step s+1 are issued before the MFMA loop of step s and land in LDS after it
  constexpr int DPL = WT / 64;               // dy values per lane per row
  constexpr int XPL = R / 64;                // full-lane x values per row (columns 0 .. MAIN-1)
  constexpr int MAIN = XPL * 64;
  constexpr int HW = R - MAIN;               // halo columns per row (KW - 1 = 2), gathered lane-wise across rows
  constexpr int XROWS = 32 * ROWS / 8;       // input rows per wave
  constexpr int HPL = (XROWS * HW + 63) / 64;
  float dreg[4][DPL];
  float xreg[XROWS][XPL];
  float hreg[HPL];

  // raw buffer loads: per-sample resources, 32-bit byte offsets, out-of-range offset -> 0.0 (no address VGPR pairs,
  // no selects); the launcher checks that one sample of x / dy is at most 2^30 bytes
  constexpr uint32_t INVALID = 0x40000000u;
  const uint32_t x_sample_b = (uint32_t)(g.c_in * vox_in * 4), dy_sample_b = (uint32_t)(g.c_out * vox_out * 4);

  auto load_step = [&](int s) {
    const int wt = s % n_wt;
    int r = s / n_wt;
    const int ho = r % g.h_out;
    r /= g.h_out;
    const int to = r % g.t_out;
    const int b = r / g.t_out;
    const int wo0 = wt * WT;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * g.c_in * vox_in), 0, (int)x_sample_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (size_t)b * g.c_out * vox_out), 0, (int)dy_sample_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((gate ? gate : dy) + (size_t)b * g.c_out * vox_out), 0, (int)dy_sample_b, 0x00020000);
    const uint32_t d_row = (uint32_t)((to * plane_out + ho * g.w_out + wo0) * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = wave * 4 + k;
      const uint32_t srow = (uint32_t)co * (uint32_t)(vox_out * 4) + d_row + (co < g.c_out ? 0u : INVALID);
#pragma unroll
      for (int e = 0; e < DPL; ++e) {
        const int p = lane + 64 * e;
        const uint32_t voff = srow + (wo0 + p < g.w_out ? (uint32_t)p * 4u : INVALID);
        float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, voff, 0, 0));
        if (gate) {
          const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, voff, 0, 0));
          if (!(gv > 0.f)) v = 0.f;
        }
        dreg[k][e] = v;
      }
    }
    const int wi0 = wo0 - g.pw;
    auto row_off = [&](int k) -> uint32_t {   // byte offset of this wave's k-th input row (ci, kt, kh), or INVALID
      const int ci = wave * 4 + k / ROWS, row = k % ROWS;
      const int kt = row / KH, kh = row - kt * KH;
      const int ti = to + kt - g.pt, hi = ho + kh - g.ph;
      const bool ok = ci < g.c_in && (unsigned)ti < (unsigned)g.t_in && (unsigned)hi < (unsigned)g.h_in;
      return ok ? (uint32_t)(((ci * g.t_in + ti) * g.h_in + hi) * g.w_in * 4) : INVALID;
    };
    uint32_t lane_voff[XPL];
#pragma unroll
    for (int e = 0; e < XPL; ++e) {
      const int wi = wi0 + lane + 64 * e;
      lane_voff[e] = (unsigned)wi < (unsigned)g.w_in ? (uint32_t)wi * 4u : INVALID;
    }
#pragma unroll
    for (int k = 0; k < XROWS; ++k) {
      const uint32_t srow = row_off(k);
#pragma unroll
      for (int e = 0; e < XPL; ++e)
        xreg[k][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, srow + lane_voff[e], 0, 0));
    }
#pragma unroll
    for (int e = 0; e < HPL; ++e) {
      const int hidx = lane + 64 * e;
      const int k = hidx / HW, c = MAIN + hidx - k * HW;
      const int wi = wi0 + c;
      const uint32_t voff = (k < XROWS ? row_off(k) : INVALID) + ((unsigned)wi < (unsigned)g.w_in ? (uint32_t)wi * 4u : INVALID);
      hreg[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, voff, 0, 0));
    }
  };
  // LDS store addresses = one per-lane base + compile-time offsets (wave w owns channels 4w .. 4w+3)
  float* const dy_st = dyT + wave * 4 * S + lane;
  float* const x_st = xT + wave * 4 * CS + lane;
  auto store_step = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < DPL; ++e) dy_st[k * S + 64 * e] = dreg[k][e];
#pragma unroll
    for (int k = 0; k < XROWS; ++k)
#pragma unroll
      for (int e = 0; e < XPL; ++e) x_st[(k / ROWS) * CS + (k % ROWS) * R + 64 * e] = xreg[k][e];
#pragma unroll
    for (int e = 0; e < HPL; ++e) {
      const int hidx = lane + 64 * e;
      const int k = hidx / HW, c = MAIN + hidx - k * HW;
      if (k < XROWS) xT[(wave * 4 + k / ROWS) * CS + (k % ROWS) * R + c] = hreg[e];
    }
  };

  const int s0 = blockIdx.x * steps_per_block;
  const int s1 = s0 + steps_per_block < n_steps ? s0 + steps_per_block : n_steps;
  if (s0 < s1) load_step(s0);
  for (int s = s0; s < s1; ++s) {
    store_step();
    __syncthreads();
    if (s + 1 < s1) load_step(s + 1);
    // ---- MFMA over this wave's half of the row: LDS operands fetched QB position pairs ahead of their MFMAs ---------
    constexpr int QB = 4;
    const int q0 = half * (WT / 4), q1 = q0 + WT / 4;
    for (int q = q0; q < q1; q += QB) {
      float av[QB];
#pragma unroll
      for (int u = 0; u < QB; ++u) av[u] = dyT[a_base + 2 * (q + u)];
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        const int tap = w4 + 4 * i;                            // wave-uniform
        if (4 * i + 3 < TAPS || tap <= TAPS) {                 // compile-time true except in the last slot
          const bool ones = 4 * i + 3 >= TAPS && tap == TAPS;
          const int o = ones ? 0 : off[i];
          float bv[QB];
#pragma unroll
          for (int u = 0; u < QB; ++u) {
            const float t = xT[b_base + o + 2 * (q + u)];
            bv[u] = ones ? 1.0f : t;
          }
#pragma unroll
          for (int u = 0; u < QB; ++u) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- combine the two half-row groups (fixed order: first half + second half), write this block's slab -----------------
  if (half == 1) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) xT[((w4 * NSLOT + i) * 16 + r) * 64 + lane] = acc[i][r];
  }
  __syncthreads();
  if (half == 0) {
    const int ci = lane & 31;
    const size_t slab = blockIdx.x;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int tap = w4 + 4 * i;
      if (tap > TAPS) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[i][r] + xT[((w4 * NSLOT + i) * 16 + r) * 64 + lane];
        if (co < g.c_out) {
          if (tap < TAPS) {
            if (ci < g.c_in) slabs[((slab * g.c_out + co) * g.c_in + ci) * TAPS + tap] = v;
          } else if (ci == 0 && bias_slabs) {
            bias_slabs[slab * g.c_out + co] = v;
          }
        }
      }
    }
  }
}

static bool wgrad_mfma_ok(const Geom& g) {
  const bool k_ok = (g.kt == 2 || g.kt == 3 || g.kt == 1) && g.kh == 3 && g.kw == 3;
  return k_ok && g.st == 1 && g.sh == 1 && g.sw == 1 && g.c_in >= 8 && g.c_in <= 32 && g.c_out >= 16 && g.c_out <= 32 &&
         g.w_out >= 32 && (long long)g.c_in * g.t_in * g.h_in * g.w_in * 4 <= (1ll << 30) &&
         (long long)g.c_out * g.t_out * g.h_out * g.w_out * 4 <= (1ll << 30);
}

static int wgrad_mfma_wt(const Geom& g) { return g.kt * g.kh > 6 ? 64 : 128; }

// out[i] = sum over slabs (fixed order: deterministic).  Block = 64 elements x 8 slab groups (one wave each, four
// independent partial sums in flight), LDS combine of the 8 partials in a fixed order -- the one-thread-per-element loop
// over 512 slabs it replaces was latency-bound at 0.4 TB/s.
__global__ __launch_bounds__(512) void slab_reduce_f32(const float* __restrict__ slabs, float* __restrict__ out, int n,
                                                       int n_slabs) {
  __shared__ float part[8][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int k = grp;
    for (; k + 24 < n_slabs; k += 32) {
      s0 += slabs[(size_t)k * n + i];
      s1 += slabs[(size_t)(k + 8) * n + i];
      s2 += slabs[(size_t)(k + 16) * n + i];
      s3 += slabs[(size_t)(k + 24) * n + i];
    }
    for (; k < n_slabs; k += 8) s0 += slabs[(size_t)k * n + i];
  }
  part[grp][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && i < n)
    out[i] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
             ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
}

// dbias[co] = sum over (b, voxels) of dy (gated); one block per channel, fixed order
__global__ __launch_bounds__(256) void conv3d_general_dbias_f32(const float* __restrict__ dy, const float* __restrict__ gate,
                                                                float* __restrict__ db, int batch, int c_out, int vox_out) {
  const int co = blockIdx.x;
  float s = 0.f;
  const long long total = (long long)batch * vox_out;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    int b = (int)(i / vox_out);
    int v = (int)(i - (long long)b * vox_out);
    size_t off = ((size_t)b * c_out + co) * vox_out + v;
    float d = dy[off];
    if (gate && !(gate[off] > 0.f)) d = 0.f;
    s += d;
  }
  __shared__ float red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) db[co] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- MaxPool3d -------------------------------------------------------------------------------------------------
// Window scan order (kt, kh, kw) ascending, update on "val > max || isnan(val)": the FIRST maximum wins and NaN
// propagates, as in torch's CPU max_pool3d; idx = flat (t*H + h)*W + w of the winner inside its input plane stack.
__global__ __launch_bounds__(256) void maxpool3d_fwd_f32(const float* __restrict__ x, float* __restrict__ y,
                                                         int32_t* __restrict__ idx, long long n_out, Geom g) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int plane_out = g.h_out * g.w_out;
  const int vox_out = g.t_out * plane_out;
  const long long p = i / vox_out;  // (b, c) plane stack
  int v = (int)(i - p * vox_out);
  int to = v / plane_out;
  int r = v - to * plane_out;
  int ho = r / g.w_out;
  int wo = r - ho * g.w_out;
  const float* xp = x + (size_t)p * g.t_in * g.h_in * g.w_in;
  float best = -INFINITY;
  int best_i = -1;
  for (int kt = 0; kt < g.kt; ++kt) {
    int ti = to * g.st - g.pt + kt;
    if ((unsigned)ti >= (unsigned)g.t_in) continue;
    for (int kh = 0; kh < g.kh; ++kh) {
      int hi = ho * g.sh - g.ph + kh;
      if ((unsigned)hi >= (unsigned)g.h_in) continue;
      for (int kw = 0; kw < g.kw; ++kw) {
        int wi = wo * g.sw - g.pw + kw;
        if ((unsigned)wi >= (unsigned)g.w_in) continue;
        int off = (ti * g.h_in + hi) * g.w_in + wi;
        float val = xp[off];
        if (best_i < 0 || val > best || val != val) {
          best = val;
          best_i = off;
        }
      }
    }
  }
  y[i] = best;
  if (idx) idx[i] = best_i;
}

// dx[p, off] = sum of dy over the output windows whose winner is off (gather: deterministic, no atomics)
__global__ __launch_bounds__(256) void maxpool3d_bwd_f32(const float* __restrict__ dy, const int32_t* __restrict__ idx,
                                                         float* __restrict__ dx, long long n_in, Geom g) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  const int plane_in = g.h_in * g.w_in;
  const int vox_in = g.t_in * plane_in;
  const long long p = i / vox_in;
  int off = (int)(i - p * vox_in);
  int ti = off / plane_in;
  int r = off - ti * plane_in;
  int hi = r / g.w_in;
  int wi = r - hi * g.w_in;
  const size_t base = (size_t)p * g.t_out * g.h_out * g.w_out;
  float s = 0.f;
  for (int kt = 0; kt < g.kt; ++kt) {
    int nt = ti + g.pt - kt;
    int to = nt / g.st;
    if (nt < 0 || to * g.st != nt || to >= g.t_out) continue;
    for (int kh = 0; kh < g.kh; ++kh) {
      int nh = hi + g.ph - kh;
      int ho = nh / g.sh;
      if (nh < 0 || ho * g.sh != nh || ho >= g.h_out) continue;
      for (int kw = 0; kw < g.kw; ++kw) {
        int nw = wi + g.pw - kw;
        int wo = nw / g.sw;
        if (nw < 0 || wo * g.sw != nw || wo >= g.w_out) continue;
        size_t o = base + ((size_t)to * g.h_out + ho) * g.w_out + wo;
        if (idx[o] == off) s += dy[o];
      }
    }
  }
  dx[i] = s;
}

// ---- mean squared error ------------------------------------------------------------------------------------------
// out[0] = mean((y_hat - y)^2) with a fixed-order single-block reduction (f64 partials); grad = 2 (y_hat - y) / n * scale
// nn.MSELoss (mean) + its gradient: MSE_BLOCKS blocks leave one double partial sum each (thread-strided double
// accumulation, fixed-order combine), a one-block second pass adds the partials in index order.
constexpr int MSE_BLOCKS = 128;
__global__ __launch_bounds__(256) void mse_partial_f32(const float* __restrict__ y_hat, const float* __restrict__ y,
                                                       long long n, float grad_scale, double* __restrict__ partial,
                                                       float* __restrict__ grad) {
  double s = 0.0;
  const float gs = 2.0f * grad_scale / (float)n;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = y_hat[i] - y[i];
    s += (double)d * (double)d;
    if (grad) grad[i] = d * gs;
  }
  __shared__ double red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(64) void mse_final_f32(const double* __restrict__ partial, int n_partial, long long n,
                                                    float* __restrict__ out) {
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < n_partial; ++k) t += partial[k];
    out[0] = (float)(t / (double)n);
  }
}

static int make_geom(const pv_conv3d_geom* d, Geom* g, const char* who) {
  PV_REQUIRE(d, PV_EINVAL, "%s: null geometry", who);
  PV_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_out > 0 && d->t_in > 0 && d->h_in > 0 && d->w_in > 0, PV_EINVAL,
             "%s: non-positive dimension", who);
  PV_REQUIRE(d->k_t >= 1 && d->k_t <= 3 && d->k_h >= 1 && d->k_h <= 3 && d->k_w >= 1 && d->k_w <= 3, PV_ESIZE,
             "%s: kernel extents must be 1..3", who);
  PV_REQUIRE(d->stride_t >= 1 && d->stride_h >= 1 && d->stride_w >= 1, PV_EINVAL, "%s: stride must be >= 1", who);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_h >= 0 && d->pad_w >= 0 && d->pad_t <= 2 && d->pad_h <= 2 && d->pad_w <= 2, PV_EINVAL,
             "%s: padding must be 0..2", who);
  PV_REQUIRE(d->t_in + 2 * d->pad_t >= d->k_t && d->h_in + 2 * d->pad_h >= d->k_h && d->w_in + 2 * d->pad_w >= d->k_w,
             PV_ESIZE, "%s: input smaller than the kernel", who);
  g->c_in = d->c_in, g->c_out = d->c_out, g->t_in = d->t_in, g->h_in = d->h_in, g->w_in = d->w_in;
  g->kt = d->k_t, g->kh = d->k_h, g->kw = d->k_w, g->st = d->stride_t, g->sh = d->stride_h, g->sw = d->stride_w;
  g->pt = d->pad_t, g->ph = d->pad_h, g->pw = d->pad_w;
  g->t_out = (d->t_in + 2 * d->pad_t - d->k_t) / d->stride_t + 1;
  g->h_out = (d->h_in + 2 * d->pad_h - d->k_h) / d->stride_h + 1;
  g->w_out = (d->w_in + 2 * d->pad_w - d->k_w) / d->stride_w + 1;
  return PV_OK;
}

static int wgrad_wco(const Geom& g) { return g.c_out >= 8 ? 8 : 4; }

static int wgrad_mfma_steps(const Geom& g, int batch) {
  return batch * g.t_out * g.h_out * ((g.w_out + wgrad_mfma_wt(g) - 1) / wgrad_mfma_wt(g));
}

static int wgrad_slabs(const Geom& g, int batch) {
  if (wgrad_mfma_ok(g)) {  // one slab per block; one resident block per CU (LDS), two rounds
    const int n_steps = wgrad_mfma_steps(g, batch);
    const int per = (n_steps + 511) / 512;
    return (n_steps + per - 1) / per;
  }
  const long long total = (long long)batch * g.t_out * g.h_out * g.w_out;
  const int wco = wgrad_wco(g);
  const long long blocks = (long long)g.c_in * ((g.c_out + wco - 1) / wco);
  long long n = (4096 + blocks - 1) / blocks;                 // aim at >= 4096 blocks ...
  const long long cap = (total + 4095) / 4096;                // ... of at least 4096 positions each
  if (n > cap) n = cap;
  if (n > 256) n = 256;
  return n < 1 ? 1 : (int)n;
}

}  // namespace pv

using namespace pv;

#define PV_DISPATCH_K(KERNEL, ...)                                                                         \
  do {                                                                                                     \
    if (g.kt == 2 && g.kh == 3 && g.kw == 3) KERNEL<2, 3, 3> __VA_ARGS__;                                  \
    else if (g.kt == 3 && g.kh == 3 && g.kw == 3) KERNEL<3, 3, 3> __VA_ARGS__;                             \
    else if (g.kt == 1 && g.kh == 3 && g.kw == 3) KERNEL<1, 3, 3> __VA_ARGS__;                             \
    else if (g.kt == 1 && g.kh == 1 && g.kw == 1) KERNEL<1, 1, 1> __VA_ARGS__;                             \
    else KERNEL<0, 0, 0> __VA_ARGS__;                                                                      \
  } while (0)

extern "C" {

int pv_conv3d_general_out_extent(const pv_conv3d_geom* d, int32_t* t_out, int32_t* h_out, int32_t* w_out) {
  Geom g;
  int rc = make_geom(d, &g, "pv_conv3d_general_out_extent");
  if (rc) return rc;
  if (t_out) *t_out = g.t_out;
  if (h_out) *h_out = g.h_out;
  if (w_out) *w_out = g.w_out;
  return PV_OK;
}

// unit-stride launch of the tiled kernel; `g` is the geometry AS SEEN BY THE PASS (dgrad: channels and extents swapped)
static bool launch_tiled_s1(const float* x, const float* gate, const float* w, const float* bias, float* y, const Geom& g,
                            int batch, int relu, int flip, int w_ci_dim, const float* out_gate, hipStream_t stream) {
  const size_t lds = (size_t)g.c_in * g.kt * g.kh * g.kw * GCO * sizeof(float);
  const int tiles = g.t_out * g.h_out * ((g.w_out + TW - 1) / TW);
  dim3 grid((unsigned)std::min((tiles + 255) / 256, 4096), (unsigned)((g.c_out + GCO - 1) / GCO), (unsigned)batch);
#define PV_TILED(KT_, KH_, KW_)                                                                                         \
  do {                                                                                                                  \
    if (gate) conv3d_tiled_s1_f32<KT_, KH_, KW_, true><<<grid, dim3(256), lds, stream>>>(x, gate, w, bias, y, g, relu,    \
                                                                                        flip, w_ci_dim, out_gate);      \
    else conv3d_tiled_s1_f32<KT_, KH_, KW_, false><<<grid, dim3(256), lds, stream>>>(x, gate, w, bias, y, g, relu, flip,  \
                                                                                    w_ci_dim, out_gate);                \
    return true;                                                                                                        \
  } while (0)
  if (g.kt == 2 && g.kh == 3 && g.kw == 3) PV_TILED(2, 3, 3);
  if (g.kt == 3 && g.kh == 3 && g.kw == 3) PV_TILED(3, 3, 3);
  if (g.kt == 1 && g.kh == 3 && g.kw == 3) PV_TILED(1, 3, 3);
#undef PV_TILED
  return false;
}

// unit-stride launch of the matrix-core kernel (same argument conventions as launch_tiled_s1); false = shape not covered
static bool launch_mfma_s1(const float* x, const float* gate, const float* w, const float* bias, float* y, const Geom& g,
                           int batch, int relu, int flip, int w_ci_dim, const float* out_gate, hipStream_t stream) {
  if (!((g.kt == 2 || g.kt == 3) && g.kh == 3 && g.kw == 3)) return false;
  // (3,3,3): the 32 -> 32 layers of the PV-yield model in exact f32 (precision="fp32"); its 11-channel first layer would
  // leave one wave pair multiplying zero channels (the tap-plane split needs two planes) and takes the tiled kernel
  if (g.c_in < 16 || g.c_in > 32 || g.c_out > 32 || g.w_out < 48) return false;
  if (g.kt == 3 && g.c_in <= 16) return false;
  if ((size_t)g.c_in * g.t_in * g.h_in * g.w_in * 4 > 0x40000000ull) return false;   // 32-bit raw-buffer offsets
  if ((size_t)g.c_out * g.t_out * g.h_out * g.w_out * 4 > 0x40000000ull) return false;
  const int n_wt = (g.w_out + 63) / 64;
  const long long cols = (long long)batch * g.t_out * n_wt;
  // output rows are marched in segments; enough segments for ~2 work items per CU, none shorter than 8 rows
  int n_hseg = (int)std::min<long long>(std::max<long long>((512 + cols - 1) / cols, 1), std::max(g.h_out / 8, 1));
  const int hseg = (g.h_out + n_hseg - 1) / n_hseg;
  n_hseg = (g.h_out + hseg - 1) / hseg;
  const long long items = cols * n_hseg;
  if (items > 0x7fffffffLL) return false;
  if (gate) return false;   // an input gate cannot be applied on the global -> LDS path: the tiled kernel serves it
#define PV_MFMA_S1(KT_, ...)                                                                                     \
  conv3d_fwd_mfma_f32<KT_, 3, 3, __VA_ARGS__><<<dim3((unsigned)items), dim3(256), 0, stream>>>(                  \
      x, w, bias, y, g, relu, flip, w_ci_dim, out_gate, hseg, n_hseg)
  if (g.kt == 3) {
    if (g.c_out <= 16) PV_MFMA_S1(3, true, true);
    else PV_MFMA_S1(3, true);
  } else if (g.c_in > 16 && g.c_out <= 16) PV_MFMA_S1(2, true, true);
  else if (g.c_in > 16) PV_MFMA_S1(2, true);
  else PV_MFMA_S1(2, false);
#undef PV_MFMA_S1
  return true;
}

static bool unit_stride(const Geom& g) { return g.st == 1 && g.sh == 1 && g.sw == 1; }

int pv_conv3d_general_fwd_f32(const float* x, const float* w, const float* bias, float* y, const pv_conv3d_geom* d,
                              int relu, void* stream) {
  Geom g;
  int rc = make_geom(d, &g, "pv_conv3d_general_fwd_f32");
  if (rc) return rc;
  PV_REQUIRE(x && w && y, PV_EINVAL, "pv_conv3d_general_fwd_f32: null pointer");
  const size_t lds = (size_t)g.c_in * g.kt * g.kh * g.kw * GCO * sizeof(float);
  PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_conv3d_general_fwd_f32: c_in=%d too large for the LDS weight tile", g.c_in);
  if (unit_stride(g) && launch_mfma_s1(x, nullptr, w, bias, y, g, d->batch, relu ? 1 : 0, 0, g.c_in, nullptr, as_stream(stream)))
    return check_launch("pv_conv3d_general_fwd_f32");
  if (unit_stride(g) && g.w_out >= TW &&
      launch_tiled_s1(x, nullptr, w, bias, y, g, d->batch, relu ? 1 : 0, 0, g.c_in, nullptr, as_stream(stream)))
    return check_launch("pv_conv3d_general_fwd_f32");
  const int vox = g.t_out * g.h_out * g.w_out;
  dim3 grid((unsigned)std::min((vox + 255) / 256, 4096), (unsigned)((g.c_out + GCO - 1) / GCO), (unsigned)d->batch);
  PV_DISPATCH_K(conv3d_general_fwd_f32, <<<grid, dim3(256), lds, as_stream(stream)>>>(x, w, bias, y, g, relu ? 1 : 0));
  return check_launch("pv_conv3d_general_fwd_f32");
}

int pv_conv3d_general_bwd_data_f32(const float* dy, const float* y_relu_mask, const float* w, float* dx,
                                   const float* x_relu_mask, const pv_conv3d_geom* d, void* stream) {
  Geom g;
  int rc = make_geom(d, &g, "pv_conv3d_general_bwd_data_f32");
  if (rc) return rc;
  PV_REQUIRE(dy && w && dx, PV_EINVAL, "pv_conv3d_general_bwd_data_f32: null pointer");
  const size_t lds = (size_t)g.c_out * g.kt * g.kh * g.kw * GCO * sizeof(float);
  PV_REQUIRE(lds <= 64 * 1024, PV_ESIZE, "pv_conv3d_general_bwd_data_f32: c_out=%d too large for the LDS weight tile",
             g.c_out);
  if (unit_stride(g) && g.w_in >= TW && g.pt <= g.kt - 1 && g.ph <= g.kh - 1 && g.pw <= g.kw - 1) {
    // dgrad = correlation of dy (padding k-1-p) with mirrored, channel-swapped weights
    Geom t = g;
    t.c_in = g.c_out, t.c_out = g.c_in;
    t.t_in = g.t_out, t.h_in = g.h_out, t.w_in = g.w_out;
    t.t_out = g.t_in, t.h_out = g.h_in, t.w_out = g.w_in;
    t.pt = g.kt - 1 - g.pt, t.ph = g.kh - 1 - g.ph, t.pw = g.kw - 1 - g.pw;
    if (launch_mfma_s1(dy, y_relu_mask, w, nullptr, dx, t, d->batch, 0, 1, g.c_in, x_relu_mask, as_stream(stream)))
      return check_launch("pv_conv3d_general_bwd_data_f32");
    if (launch_tiled_s1(dy, y_relu_mask, w, nullptr, dx, t, d->batch, 0, 1, g.c_in, x_relu_mask, as_stream(stream)))
      return check_launch("pv_conv3d_general_bwd_data_f32");
  }
  const int vox = g.t_in * g.h_in * g.w_in;
  dim3 grid((unsigned)std::min((vox + 255) / 256, 4096), (unsigned)((g.c_in + GCO - 1) / GCO), (unsigned)d->batch);
  PV_DISPATCH_K(conv3d_general_dgrad_f32, <<<grid, dim3(256), lds, as_stream(stream)>>>(dy, y_relu_mask, w, dx, g, x_relu_mask));
  return check_launch("pv_conv3d_general_bwd_data_f32");
}

int pv_conv3d_general_bwd_weight_workspace_bytes(const pv_conv3d_geom* d, size_t* bytes) {
  Geom g;
  int rc = make_geom(d, &g, "pv_conv3d_general_bwd_weight_workspace_bytes");
  if (rc) return rc;
  PV_REQUIRE(bytes, PV_EINVAL, "pv_conv3d_general_bwd_weight_workspace_bytes: null pointer");
  *bytes = (size_t)wgrad_slabs(g, d->batch) * ((size_t)g.c_out * g.c_in * g.kt * g.kh * g.kw + g.c_out) * sizeof(float);
  return PV_OK;
}

int pv_conv3d_general_bwd_weight_f32(const float* x, const float* dy, const float* y_relu_mask, float* dw, float* dbias,
                                     const pv_conv3d_geom* d, void* ws, size_t ws_bytes, void* stream) {
  Geom g;
  int rc = make_geom(d, &g, "pv_conv3d_general_bwd_weight_f32");
  if (rc) return rc;
  PV_REQUIRE(x && dy, PV_EINVAL, "pv_conv3d_general_bwd_weight_f32: null pointer");
  hipStream_t st = as_stream(stream);
  if (dw) {
    const int n_slabs = wgrad_slabs(g, d->batch);
    const int n = g.c_out * g.c_in * g.kt * g.kh * g.kw;
    PV_REQUIRE(ws && ws_bytes >= (size_t)n_slabs * ((size_t)n + g.c_out) * sizeof(float), PV_EINVAL,
               "pv_conv3d_general_bwd_weight_f32: workspace too small");
    float* slabs = (float*)ws;
    float* bias_slabs = dbias ? slabs + (size_t)n_slabs * n : nullptr;
    if (wgrad_mfma_ok(g)) {
      const int n_steps = wgrad_mfma_steps(g, d->batch);
      const int per = (n_steps + n_slabs - 1) / n_slabs;
      if (g.kt == 2) conv3d_wgrad_mfma_f32<2, 3, 3><<<dim3((unsigned)n_slabs), dim3(512), 0, st>>>(x, dy, y_relu_mask, slabs, bias_slabs, g, d->batch, n_steps, per);
      else if (g.kt == 3) conv3d_wgrad_mfma_f32<3, 3, 3><<<dim3((unsigned)n_slabs), dim3(512), 0, st>>>(x, dy, y_relu_mask, slabs, bias_slabs, g, d->batch, n_steps, per);
      else conv3d_wgrad_mfma_f32<1, 3, 3><<<dim3((unsigned)n_slabs), dim3(512), 0, st>>>(x, dy, y_relu_mask, slabs, bias_slabs, g, d->batch, n_steps, per);
      rc = check_launch("pv_conv3d_general_bwd_weight_f32(mfma)");
      if (rc) return rc;
      slab_reduce_f32<<<dim3((unsigned)((n + 63) / 64)), dim3(512), 0, st>>>(slabs, dw, n, n_slabs);
      if (dbias) slab_reduce_f32<<<dim3((unsigned)((g.c_out + 63) / 64)), dim3(512), 0, st>>>(bias_slabs, dbias, g.c_out, n_slabs);
      return check_launch("pv_conv3d_general_bwd_weight_f32(reduce)");
    }
    const int wco = wgrad_wco(g);
    dim3 grid((unsigned)g.c_in, (unsigned)((g.c_out + wco - 1) / wco), (unsigned)n_slabs);
    const bool pairs = unit_stride(g);
#define PV_WG(KT_, KH_, KW_)                                                                                              \
  do {                                                                                                                    \
    if (wco == 8 && pairs)                                                                                                \
      conv3d_general_wgrad_f32<KT_, KH_, KW_, 8, 2><<<grid, dim3(256), 0, st>>>(x, dy, y_relu_mask, slabs, bias_slabs, g,   \
                                                                               d->batch, n_slabs);                        \
    else if (wco == 8)                                                                                                    \
      conv3d_general_wgrad_f32<KT_, KH_, KW_, 8, 1><<<grid, dim3(256), 0, st>>>(x, dy, y_relu_mask, slabs, bias_slabs, g,   \
                                                                               d->batch, n_slabs);                        \
    else                                                                                                                  \
      conv3d_general_wgrad_f32<KT_, KH_, KW_, 4, 1><<<grid, dim3(256), 0, st>>>(x, dy, y_relu_mask, slabs, bias_slabs, g,   \
                                                                               d->batch, n_slabs);                        \
  } while (0)
    if (g.kt == 2 && g.kh == 3 && g.kw == 3) PV_WG(2, 3, 3);
    else if (g.kt == 3 && g.kh == 3 && g.kw == 3) PV_WG(3, 3, 3);
    else if (g.kt == 1 && g.kh == 3 && g.kw == 3) PV_WG(1, 3, 3);
    else if (g.kt == 1 && g.kh == 1 && g.kw == 1) PV_WG(1, 1, 1);
    else PV_REQUIRE(false, PV_ESIZE, "pv_conv3d_general_bwd_weight_f32: kernel extent (%d,%d,%d) not instantiated", g.kt,
                    g.kh, g.kw);
#undef PV_WG
    rc = check_launch("pv_conv3d_general_bwd_weight_f32");
    if (rc) return rc;
    slab_reduce_f32<<<dim3((unsigned)((n + 63) / 64)), dim3(512), 0, st>>>(slabs, dw, n, n_slabs);
    if (dbias) slab_reduce_f32<<<dim3((unsigned)((g.c_out + 63) / 64)), dim3(512), 0, st>>>(bias_slabs, dbias, g.c_out, n_slabs);
    return check_launch("pv_conv3d_general_bwd_weight_f32(reduce)");
  }
  if (dbias) {
    conv3d_general_dbias_f32<<<dim3((unsigned)g.c_out), dim3(256), 0, st>>>(dy, y_relu_mask, dbias, d->batch, g.c_out,
                                                                            g.t_out * g.h_out * g.w_out);
    rc = check_launch("pv_conv3d_general_bwd_weight_f32(dbias)");
  }
  return rc;
}

int pv_maxpool3d_fwd_f32(const float* x, float* y, int32_t* argmax, const pv_conv3d_geom* d, void* stream) {
  Geom g;
  int rc = make_geom(d, &g, "pv_maxpool3d_fwd_f32");
  if (rc) return rc;
  PV_REQUIRE(x && y, PV_EINVAL, "pv_maxpool3d_fwd_f32: null pointer");
  PV_REQUIRE(g.pt * 2 <= g.kt && g.ph * 2 <= g.kh && g.pw * 2 <= g.kw, PV_EINVAL,
             "pv_maxpool3d_fwd_f32: padding must be at most half the window");
  const long long n_out = (long long)d->batch * g.c_in * g.t_out * g.h_out * g.w_out;
  maxpool3d_fwd_f32<<<dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(x, y, argmax, n_out, g);
  return check_launch("pv_maxpool3d_fwd_f32");
}

int pv_maxpool3d_bwd_f32(const float* dy, const int32_t* argmax, float* dx, const pv_conv3d_geom* d, void* stream) {
  Geom g;
  int rc = make_geom(d, &g, "pv_maxpool3d_bwd_f32");
  if (rc) return rc;
  PV_REQUIRE(dy && argmax && dx, PV_EINVAL, "pv_maxpool3d_bwd_f32: null pointer");
  const long long n_in = (long long)d->batch * g.c_in * g.t_in * g.h_in * g.w_in;
  maxpool3d_bwd_f32<<<dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(dy, argmax, dx, n_in, g);
  return check_launch("pv_maxpool3d_bwd_f32");
}

int pv_mse_loss_f32(const float* y_hat, const float* y, int64_t n, float grad_scale, float* out, float* grad,
                    void* stream) {
  PV_REQUIRE(y_hat && y && out, PV_EINVAL, "pv_mse_loss_f32: null pointer");
  PV_REQUIRE(n > 0, PV_EINVAL, "pv_mse_loss_f32: n must be positive");
  // per-device scratch for the block partials (1 KB, allocated on first use, never freed)
  static double* partial_ws[64] = {nullptr};
  int dev = 0;
  PV_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, PV_ELAUNCH, "pv_mse_loss_f32: hipGetDevice failed");
  if (!partial_ws[dev])
    PV_REQUIRE(hipMalloc((void**)&partial_ws[dev], MSE_BLOCKS * sizeof(double)) == hipSuccess, PV_ELAUNCH,
               "pv_mse_loss_f32: scratch allocation failed");
  const int blocks = (int)std::min<long long>(MSE_BLOCKS, (n + 255) / 256);
  mse_partial_f32<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(y_hat, y, (long long)n, grad_scale,
                                                                               partial_ws[dev], grad);
  mse_final_f32<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(partial_ws[dev], blocks, (long long)n, out);
  return check_launch("pv_mse_loss_f32");
}

}  // extern "C"
