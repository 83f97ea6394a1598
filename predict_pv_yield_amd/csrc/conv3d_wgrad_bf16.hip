// bf16 MFMA weight gradient of the 3x3x3 Conv3D (autograd wgrad of F.conv3d,
// predict_pv_yield/models/conv3d/model.py:117-120 under loss.backward()).
//
//   dW[co][tap][ci] = sum_voxels dYeff[v][co] * X[v + tap][ci],   dYeff = dY ⊙ (Y > 0)
//
// GEMM view: M = co (32), N = ci (CPAD), K = voxels.  The contraction index is the VOXEL, while both
// operands are stored voxel-major / channel-minor (NDHWC), i.e. transposed with respect to what an
// MFMA lane wants (8 consecutive k per lane).  gfx950's ds_read_b64_tr_b16 does that transpose on the
// LDS read: a 16-lane group fetches a 4-voxel x 16-channel block and every lane receives one channel
// of the 4 voxels.  Two such reads = one 32x32x16 operand fragment.
//
// Structure: same time-marching tile as the forward kernel (8 output rows x 64 columns, X ring of
// three slices in LDS, dY tile of the current slice beside it); the 27 taps are dealt round-robin to
// the 4 waves (7,7,7,6 accumulators of 32x32 f32); wave 3's spare accumulator multiplies dYeff by a
// fragment of ones, which yields dbias for free.  With 16 (padded) input channels the 32 B-operand columns hold TWO
// taps (columns 0..15: tap 2p, columns 16..31: tap 2p+1, the ones-tap being tap 27), so 14 accumulators (4,4,3,3)
// cover all taps instead of 28 half-empty ones.  Each workgroup writes one f32 partial slab; a
// second kernel sums the slabs in a fixed order (deterministic, no atomics).
#include "pv_common.h"

namespace pv {

constexpr int WTR = 8;
constexpr int WTRI = WTR + 2;
constexpr int WTW = 64;
constexpr int WTW_VALID = WTW - 2;
constexpr int SLAB_ELEMS = 28 * 32 * 32;  // 27 taps + ones-tap, [co][ci] each

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifdef PV_DIAG_STAMPS
__device__ unsigned long long wgrad_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif

__device__ __forceinline__ uint32_t wg_gate_word(uint32_t x, uint32_t g) {
  uint32_t lo = ((g & 0x7fffu) != 0u && (g & 0x8000u) == 0u) ? 0x0000ffffu : 0u;
  uint32_t hi = ((g & 0x7fff0000u) != 0u && (g & 0x80000000u) == 0u) ? 0xffff0000u : 0u;
  return x & (lo | hi);
}

// scheduling shape of one k-step: MFMA q, then the q-th share of the 2 + 2 * NACC LDS reads of the next k-step
template <int NACC, int Q>
__device__ __forceinline__ void wg_interleave() {
  if constexpr (Q < NACC) {
    constexpr int R = 2 + 2 * NACC;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, R / NACC + (Q < R % NACC ? 1 : 0), 0);
    wg_interleave<NACC, Q + 1>();
  }
}

template <int CPAD, bool HAS_GATE>
__global__ __launch_bounds__(256, 1) void conv3d_wgrad_bf16_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, const uint16_t* __restrict__ ymask,
    float* __restrict__ slabs, int t_in, int h_in, int w_in, int t_out, int h_out, int w_out, int pad_t,
    int pad_h, int pad_w, int n_colblk, int t_chunk) {
  constexpr int NCH = CPAD / 8;
  constexpr int VPR = 16 / NCH;
  constexpr int VOXB = CPAD * 2;
  constexpr int ROWB = WTW * VOXB;
  constexpr int SLOTB = WTRI * ROWB;
  constexpr int NLOAD_X = WTRI * WTW * NCH / 256;
  // dY tile: always 32 channels
  constexpr int DROWB = WTW * 64;
  constexpr int DTILEB = WTR * DROWB;
  constexpr int NLOAD_D = WTR * WTW * 4 / 256;
  // [X ring: 3 slots][512 B of zeros: reads of the last row run past the ring][128 B of bf16 ones + 128 B of zeros: the
  // B operand of the ones-tap (dbias) and of the empty tap slots, fetched like any other fragment][dY tile]
  __shared__ __attribute__((aligned(256))) unsigned char lds[3 * SLOTB + 512 + 256 + DTILEB];
  unsigned char* lds_const = lds + 3 * SLOTB + 512;
  unsigned char* lds_dy = lds_const + 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // transposed-read roles of this lane
  const int grp = lane >> 4;        // 16-lane group
  const int qi = (lane & 15) >> 2;  // block row (voxel) this lane addresses
  const int pi = lane & 3;          // 4-channel piece this lane addresses
  const int hh = grp >> 1;          // k half (voxels 8*hh ..)
  const int cb = 16 * (grp & 1);    // channel base of the group

  const int rowblk = blockIdx.x / n_colblk;
  const int colblk = blockIdx.x - rowblk * n_colblk;
  const int h0 = rowblk * WTR;
  const int w0 = colblk * WTW_VALID;
  const int b = blockIdx.z;
  const int tc0 = blockIdx.y * t_chunk;
  const int tc1 = min(tc0 + t_chunk, t_out);
  const int wg_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* slab = slabs + (size_t)wg_id * SLAB_ELEMS;

  constexpr bool PAIRED = CPAD == 16;      // two taps per accumulator
  constexpr int NACC = PAIRED ? 4 : 7;
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  if (tc0 < tc1) {
    if (tid < 128) reinterpret_cast<uint32_t*>(lds + 3 * SLOTB)[tid] = 0u;
    if (tid < 64) reinterpret_cast<uint32_t*>(lds_const)[tid] = tid < 32 ? 0x3f803f80u : 0u;

    // ---- per-lane byte offsets of the transposed reads --------------------------------------
    // dY (A operand, 32 channels): voxel column 8*hh + 4*s + qi, channels cb + 4*pi .. +3
    int aoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int v = 8 * hh + 4 * s + qi;
      int chunk = (cb + 4 * pi) >> 3;
      aoff[s] = v * 64 + ((chunk ^ ((v >> 2) & 3)) << 4) + (pi & 1) * 8;
    }
    // X (B operand) per tap of this wave: tap = wave + 4*i; PAIRED: this LANE's tap = 2*(wave + 4*i) + (grp & 1)
    int boff[NACC][2];
    int bkt[NACC];
    int my_tap[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      int tap = PAIRED ? 2 * (wave + 4 * i) + (grp & 1) : wave + 4 * i;
      my_tap[i] = tap;
      if (tap > 26) tap = 26;
      int kt = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      bkt[i] = kt;
      const int cbx = PAIRED ? 0 : cb;  // PAIRED: both column halves read channels 0..15, each at its own tap
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        int v = kw + 8 * hh + 4 * s + qi;
        int chunk = (cbx + 4 * pi) >> 3;
        boff[i][s] = kh * ROWB + v * VOXB + ((chunk ^ ((v / VPR) % NCH)) << 4) + (pi & 1) * 8;
      }
    }

    // ---- staging -----------------------------------------------------------------------------

    // staging geometry (see conv3d_bf16.hip): rows are wave-uniform, loads are branch-free (clamped address +
    // select), so the NLOAD loads of a slice issue back to back under the MFMAs of the previous one
    constexpr int CPR = WTW * NCH;
    constexpr int RPI = 256 / CPR;
    const int srow0 = RPI == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid / CPR);
    const int srem = tid - srow0 * CPR;
    const int scol = srem / NCH, sc = srem - scol * NCH;
    const int swi = w0 - pad_w + scol;
    const bool scol_ok = (unsigned)swi < (unsigned)w_in;
    const int x_lds_lane = scol * VOXB + ((sc ^ ((scol / VPR) % NCH)) << 4);
    // dY tile: 256 chunks per row, one row per i
    const int dcol = tid >> 2, dc = tid & 3;
    const int dwo = w0 + dcol;
    const bool dcol_ok = dcol < WTW_VALID && dwo < w_out;
    const int d_lds_lane = dcol * 64 + ((dc ^ ((dcol >> 2) & 3)) << 4);
    // raw buffer loads, hardware zero fill for everything outside the tensor (see conv3d_bf16.hip)
    constexpr uint32_t INVALID = 0x40000000u;
    const uint32_t x_plane_b = (uint32_t)h_in * w_in * CPAD * 2u, x_row_b = (uint32_t)w_in * CPAD * 2u;
    const uint32_t d_plane_b = (uint32_t)h_out * w_out * 64u, d_row_b = (uint32_t)w_out * 64u;
    const size_t x_sample = (size_t)t_in * h_in * w_in * CPAD, d_sample = (size_t)t_out * h_out * w_out * 32;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * x_sample), 0, (int)(x_sample * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (size_t)b * d_sample), 0, (int)(d_sample * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((HAS_GATE ? ymask : dy) + (size_t)b * d_sample), 0, (int)(d_sample * 2), 0x00020000);
    const uint32_t x_lane_voff = scol_ok ? (uint32_t)(swi * CPAD + sc * 8) * 2u : INVALID;
    const uint32_t d_lane_voff = dcol_ok ? (uint32_t)(dwo * 32 + dc * 8) * 2u : INVALID;

    u32x4 stage_x[NLOAD_X];
    u32x4 stage_d[NLOAD_D];
    u32x4 stage_m[HAS_GATE ? NLOAD_D : 1];
    auto load_x = [&](int s) {
      const int ti = s - pad_t;
      const bool t_ok = (unsigned)ti < (unsigned)t_in;
      const uint32_t toff = (uint32_t)min(max(ti, 0), t_in - 1) * x_plane_b;
#pragma unroll
      for (int i = 0; i < NLOAD_X; ++i) {
        const int hi = h0 - pad_h + i * RPI + srow0;
        const bool row_ok = t_ok && (unsigned)hi < (unsigned)h_in;
        const uint32_t srow = toff + (uint32_t)min(max(hi, 0), h_in - 1) * x_row_b + (row_ok ? 0u : INVALID);
        stage_x[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, x_lane_voff + srow, 0, 0);
      }
    };
    auto store_x = [&](int s) {
      unsigned char* slot = lds + (s % 3) * SLOTB + srow0 * ROWB + x_lds_lane;
#pragma unroll
      for (int i = 0; i < NLOAD_X; ++i) *reinterpret_cast<u32x4*>(slot + i * RPI * ROWB) = stage_x[i];
    };
    auto load_d = [&](int t) {
      const uint32_t toff = (uint32_t)t * d_plane_b;
#pragma unroll
      for (int i = 0; i < NLOAD_D; ++i) {
        const int ho = h0 + i;
        const bool row_ok = ho < h_out;
        const uint32_t srow = toff + (uint32_t)min(ho, h_out - 1) * d_row_b + (row_ok ? 0u : INVALID);
        stage_d[i] = __builtin_amdgcn_raw_buffer_load_b128(drsrc, d_lane_voff + srow, 0, 0);
        if constexpr (HAS_GATE) stage_m[i] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, d_lane_voff + srow, 0, 0);
      }
    };
    auto store_d = [&]() {
#pragma unroll
      for (int i = 0; i < NLOAD_D; ++i) {
        u32x4 v = stage_d[i];
        if constexpr (HAS_GATE) {
          const u32x4 g = stage_m[i];
          v[0] = wg_gate_word(v[0], g[0]); v[1] = wg_gate_word(v[1], g[1]);
          v[2] = wg_gate_word(v[2], g[2]); v[3] = wg_gate_word(v[3], g[3]);
        }
        *reinterpret_cast<u32x4*>(lds_dy + i * DROWB + d_lds_lane) = v;
      }
    };


    load_x(tc0);
    store_x(tc0);
    load_x(tc0 + 1);
    store_x(tc0 + 1);
    load_x(tc0 + 2);
    load_d(tc0);

#ifdef PV_DIAG_STAMPS
    unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, s0, s1, s2, s3, s4, s5;
#endif
    for (int t = tc0; t < tc1; ++t) {
      PV_STAMP(s0);
      store_x(t + 2);
      store_d();
      PV_STAMP(s1);
      __syncthreads();
      PV_STAMP(s2);
      if (t + 1 < tc1) {
        load_x(t + 3);
        load_d(t + 1);
      }
      PV_STAMP(s3);
      int slot_of_kt[3];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) slot_of_kt[kt] = ((t + kt) % 3) * SLOTB;

      // 32 k-steps (8 rows x 4 column groups of 16 voxels); each = 16 transposed reads + 7 MFMAs.  Software-pipelined
      // by one k-step: all reads of step s+1 are issued before the MFMAs of step s (one wave per SIMD: the LDS
      // latency has nothing else to hide behind).
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      // this lane's constant fragment for the last slot: ones for tap 27, zeros past it, none (nullptr) for a real tap
      const int last_tap = PAIRED ? my_tap[NACC - 1] : (wave == 3 ? 27 : 0);
      const unsigned char* const_frag =
          last_tap < 27 ? nullptr : lds_const + (last_tap == 27 ? 0 : 128) + (lane & 15) * 8;
      int bslot[NACC];
#pragma unroll
      for (int i = 0; i < NACC; ++i) bslot[i] = bkt[i] == 0 ? slot_of_kt[0] : (bkt[i] == 1 ? slot_of_kt[1] : slot_of_kt[2]);
      auto read_step = [&](int st, s16x4 (&ra)[2], s16x4 (&rb)[NACC][2]) {
        const int rho = st >> 2, cg = st & 3;
        const unsigned char* ap = lds_dy + rho * DROWB + cg * 16 * 64;
        ra[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + aoff[0]));
        ra[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + aoff[1]));
        const int xo = rho * ROWB + cg * 16 * VOXB;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          const unsigned char* bp = lds + bslot[i] + xo;
          const unsigned char* p0 = bp + boff[i][0];
          const unsigned char* p1 = bp + boff[i][1];
          if (i == NACC - 1) {  // only the last slot can hold the ones-tap (27) or run past it: the select is on the
            p0 = const_frag ? const_frag : p0;  // ADDRESS, so nothing waits for the data (a select on the fragment put an
            p1 = const_frag ? const_frag : p1;  // lgkmcnt(0) behind every k-step's reads)
          }
          rb[i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
          rb[i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
        }
      };
      auto mfma_step = [&](const s16x4 (&ra)[2], const s16x4 (&rb)[NACC][2]) {
        const s16x8 a8 = {ra[0][0], ra[0][1], ra[0][2], ra[0][3], ra[1][0], ra[1][1], ra[1][2], ra[1][3]};
        const bf16x8 afr = __builtin_bit_cast(bf16x8, a8);
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          s16x8 t8 = {rb[i][0][0], rb[i][0][1], rb[i][0][2], rb[i][0][3], rb[i][1][0], rb[i][1][1], rb[i][1][2], rb[i][1][3]};
          bf16x8 bfr = __builtin_bit_cast(bf16x8, t8);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[i], 0, 0, 0);
        }
      };
      s16x4 ra0[2], ra1[2], rb0[NACC][2], rb1[NACC][2];
      read_step(0, ra0, rb0);
      // The 2 + 2 * NACC transposed reads of k-step s+1 are dealt out between the NACC MFMAs of k-step s: a read issued in
      // an MFMA gap costs about a cycle, a block of 16 reads in front of 7 back-to-back MFMAs left the matrix pipe idle
      // for the whole block (PMC, round 2: pipe busy 50 % of the wave's residency).
      for (int st = 0; st < WTR * 4; st += 2) {
        __builtin_amdgcn_sched_barrier(0);
        read_step(st + 1, ra1, rb1);
        mfma_step(ra0, rb0);
        wg_interleave<NACC, 0>();
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < WTR * 4) {
          read_step(st + 2, ra0, rb0);
          mfma_step(ra1, rb1);
          wg_interleave<NACC, 0>();
        } else {
          mfma_step(ra1, rb1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      PV_STAMP(s4);
      __syncthreads();
#ifdef PV_DIAG_STAMPS
      PV_STAMP(s5);
      dg[0] += s1 - s0;   // wait for the staged loads + LDS writes of the next slice
      dg[1] += s2 - s1;   // barrier
      dg[2] += s3 - s2;   // issue of the next slice's global loads
      dg[3] += s4 - s3;   // 32 k-steps
      dg[4] += s5 - s4;   // barrier
      dg[5] += 1;
#endif
    }
#ifdef PV_DIAG_STAMPS
    if (lane == 0 && wg_id * 4 + wave < PV_DIAG_WAVES)
      for (int i = 0; i < PV_DIAG_SLOTS; ++i) wgrad_diag[(size_t)(wg_id * 4 + wave) * PV_DIAG_SLOTS + i] = dg[i];
#endif
  }

  // ---- write the partial slab: [tapslot][co][ci], C layout: col = ci = lane&31, row = co ----------
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    // PAIRED: column c of the accumulator belongs to tap 2p + (c >> 4), input channel c & 15
    const int tapslot = PAIRED ? 2 * (wave + 4 * i) + ((lane & 31) >> 4) : wave + 4 * i;  // 27 = ones-tap
    const int ci = PAIRED ? (lane & 15) : (lane & 31);
    if (tapslot <= 27) {
      float* dst = slab + (size_t)tapslot * 1024 + ci;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = (j & 3) + 8 * (j >> 2) + 4 * half;
        dst[co * 32] = acc[i][j];
      }
    }
  }
}

// dw[co][ci][tap] = sum over slabs (fixed order: deterministic); dbias[co] from the ones-tap column 0.
// block = 64 elements x 16 slab groups (one wave each, coalesced 256-byte reads): a thread adds slabs g, g + 16, ... with
// all of its (<= 16 for 256 slabs) loads in flight, then the 16 group sums are combined in a fixed tree.  (4 groups of up to
// 64 terms each, four loads at a time, made the launch a chain of ~16 memory latencies: 8.6-10 us for 29 MB.)
constexpr int WR_GROUPS = 16;
__global__ __launch_bounds__(64 * WR_GROUPS) void conv3d_wgrad_reduce_kernel(const float* __restrict__ slabs, int n_slabs,
                                                                              float* __restrict__ dw, float* __restrict__ dbias,
                                                                              int c_out, int c_in) {
  __shared__ float part[WR_GROUPS][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;  // index into [28][32][32]; SLAB_ELEMS % 64 == 0
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k = grp; k < n_slabs; k += 16 * WR_GROUPS) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (k + WR_GROUPS * j < n_slabs) ? slabs[(size_t)(k + WR_GROUPS * j) * SLAB_ELEMS + i] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; j += 4) s0 += v[j], s1 += v[j + 1], s2 += v[j + 2], s3 += v[j + 3];
  }
  part[grp][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0) {
    float t[WR_GROUPS];
#pragma unroll
    for (int q = 0; q < WR_GROUPS; ++q) t[q] = part[q][lane];
#pragma unroll
    for (int w = WR_GROUPS / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int q = 0; q < w; ++q) t[q] = t[2 * q] + t[2 * q + 1];
    const float s = t[0];
    const int tap = i >> 10, co = (i >> 5) & 31, ci = i & 31;
    if (tap < 27) {
      if (dw && co < c_out && ci < c_in) dw[((size_t)co * c_in + ci) * 27 + tap] = s;
    } else if (dbias && ci == 0 && co < c_out) {
      dbias[co] = s;
    }
  }
}

static void wgrad_grid(const pv_conv3d_dims* d, int* n_rowblk, int* n_colblk, int* n_tchunk, int* t_chunk) {
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  *n_rowblk = (ho + WTR - 1) / WTR;
  *n_colblk = (wo + WTW_VALID - 1) / WTW_VALID;
  long long tiles = (long long)d->batch * *n_rowblk * *n_colblk;
  // One workgroup per CU and launch round.  Cutting the time march gives more workgroups but every chunk re-stages two
  // halo slices: pick the cut that minimises rounds x (slices per chunk + 2).  (224 tiles -- 56 output rows at B = 32 --
  // were cut in two: 2 rounds x 7 slice loads instead of 1 round x 12.)
  int max_chunks = (to + 1) / 2;
  if (max_chunks < 1) max_chunks = 1;
  int ntc = 1;
  long long best = -1;
  for (int c = 1; c <= max_chunks && tiles * c <= 8 * 256; ++c) {
    const int tch = (to + c - 1) / c, nch = (to + tch - 1) / tch;
    const long long rounds = (tiles * nch + 255) / 256;
    const long long cost = rounds * (tch + 2);
    if (best < 0 || cost < best) best = cost, ntc = nch;
  }
  *t_chunk = (to + ntc - 1) / ntc;
  *n_tchunk = (to + *t_chunk - 1) / *t_chunk;
}

// LDS-direct form (conv3d_wgrad_bf16_v2.hip): serves every launch whose dY needs no gate while it is staged
size_t wgrad_v2_workspace_bytes(const pv_conv3d_dims* d);
int launch_conv3d_wgrad_bf16_v2(const uint16_t* x, const uint16_t* dy, float* slabs, const pv_conv3d_dims* d, int to, int ho,
                                int wo, hipStream_t st, int* n_slabs, bool f16);

}  // namespace pv

using namespace pv;

extern "C" {

int pv_conv3d_bwd_weight_bf16_workspace_bytes(const pv_conv3d_dims* d, size_t* bytes) {
  PV_REQUIRE(d && bytes, PV_EINVAL, "pv_conv3d_bwd_weight_bf16_workspace_bytes: null pointer");
  PV_REQUIRE(d->batch > 0 && d->t_in > 0 && d->h_in > 0 && d->w_in > 0, PV_EINVAL,
             "pv_conv3d_bwd_weight_bf16_workspace_bytes: bad dims");
  int nrb, ncb, ntc, tch;
  wgrad_grid(d, &nrb, &ncb, &ntc, &tch);
  *bytes = std::max((size_t)d->batch * nrb * ncb * ntc * SLAB_ELEMS * sizeof(float), wgrad_v2_workspace_bytes(d));
  return PV_OK;
}

int pv_conv3d_bwd_weight_bf16(const uint16_t* x, const uint16_t* dy, const uint16_t* y_relu_mask, float* dw,
                              float* dbias, const pv_conv3d_dims* d, void* workspace, size_t workspace_bytes,
                              void* stream) {
  PV_REQUIRE(d && x && dy && workspace, PV_EINVAL, "pv_conv3d_bwd_weight_bf16: null pointer");
  PV_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_in <= 32 && d->c_out > 0 && d->c_out <= 32, PV_ESIZE,
             "pv_conv3d_bwd_weight_bf16: channels (%d -> %d) must be in 1..32", d->c_in, d->c_out);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_t <= 2 && d->pad_h >= 0 && d->pad_h <= 2 && d->pad_w >= 0 && d->pad_w <= 2,
             PV_EINVAL, "pv_conv3d_bwd_weight_bf16: padding must be 0..2");
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  PV_REQUIRE(to > 0 && ho > 0 && wo > 0, PV_ESIZE, "pv_conv3d_bwd_weight_bf16: input smaller than the kernel");
  int nrb, ncb, ntc, tch;
  wgrad_grid(d, &nrb, &ncb, &ntc, &tch);
  const int n_slabs = d->batch * nrb * ncb * ntc;
  PV_REQUIRE(workspace_bytes >= (size_t)n_slabs * SLAB_ELEMS * sizeof(float), PV_ESIZE,
             "pv_conv3d_bwd_weight_bf16: workspace too small");
  PV_REQUIRE(d->batch <= 65535, PV_ESIZE, "pv_conv3d_bwd_weight_bf16: batch too large for grid.z");
  PV_REQUIRE((size_t)d->t_in * d->h_in * d->w_in * 64 <= 0x40000000ull, PV_ESIZE,
             "pv_conv3d_bwd_weight_bf16: one sample exceeds 1 GiB (buffer-addressing limit of this kernel)");
  hipStream_t st = as_stream(stream);
  if (!y_relu_mask && workspace_bytes >= wgrad_v2_workspace_bytes(d)) {
    int n2 = 0;
    if (launch_conv3d_wgrad_bf16_v2(x, dy, (float*)workspace, d, to, ho, wo, st, &n2, false) == 0) {
      hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(SLAB_ELEMS / 64), dim3(64 * WR_GROUPS), 0, st,
                         (const float*)workspace, n2, dw, dbias, d->c_out, d->c_in);
      return check_launch("pv_conv3d_bwd_weight_bf16(v2)");
    }
  }
  dim3 grid((unsigned)(nrb * ncb), (unsigned)ntc, (unsigned)d->batch);
  const int cpad = pv_bf16_cpad(d->c_in);
#define PV_LAUNCH_WGRAD(CP, HG)                                                                                   \
  hipLaunchKernelGGL((conv3d_wgrad_bf16_kernel<CP, HG>), grid, dim3(256), 0, st, x, dy, y_relu_mask, (float*)workspace, \
                     d->t_in, d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, ncb, tch)
  if (cpad == 16) {
    if (y_relu_mask) PV_LAUNCH_WGRAD(16, true); else PV_LAUNCH_WGRAD(16, false);
  } else {
    if (y_relu_mask) PV_LAUNCH_WGRAD(32, true); else PV_LAUNCH_WGRAD(32, false);
  }
#undef PV_LAUNCH_WGRAD
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(SLAB_ELEMS / 64), dim3(64 * WR_GROUPS), 0, st,
                     (const float*)workspace, n_slabs, dw, dbias, d->c_out, d->c_in);
  return check_launch("pv_conv3d_bwd_weight_bf16");
}

int pv_conv3d_bwd_weight_f16(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, const pv_conv3d_dims* d,
                             void* workspace, size_t workspace_bytes, void* stream) {
  PV_REQUIRE(d && x && dy && workspace, PV_EINVAL, "pv_conv3d_bwd_weight_f16: null pointer");
  PV_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_in <= 32 && d->c_out > 0 && d->c_out <= 32, PV_ESIZE,
             "pv_conv3d_bwd_weight_f16: channels (%d -> %d) must be in 1..32", d->c_in, d->c_out);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_t <= 2 && d->pad_h >= 0 && d->pad_h <= 2 && d->pad_w >= 0 && d->pad_w <= 2,
             PV_EINVAL, "pv_conv3d_bwd_weight_f16: padding must be 0..2");
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  PV_REQUIRE(to > 0 && ho > 0 && wo > 0, PV_ESIZE, "pv_conv3d_bwd_weight_f16: input smaller than the kernel");
  PV_REQUIRE(d->batch <= 65535, PV_ESIZE, "pv_conv3d_bwd_weight_f16: batch too large for grid.z");
  PV_REQUIRE((size_t)d->t_in * d->h_in * d->w_in * 64 <= 0x40000000ull, PV_ESIZE,
             "pv_conv3d_bwd_weight_f16: one sample exceeds 1 GiB (buffer-addressing limit of this kernel)");
  PV_REQUIRE(workspace_bytes >= wgrad_v2_workspace_bytes(d), PV_ESIZE, "pv_conv3d_bwd_weight_f16: workspace too small");
  hipStream_t st = as_stream(stream);
  int n2 = 0;
  PV_REQUIRE(launch_conv3d_wgrad_bf16_v2(x, dy, (float*)workspace, d, to, ho, wo, st, &n2, true) == 0, PV_EINVAL,
             "pv_conv3d_bwd_weight_f16: x and dy must be 16-byte aligned");
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(SLAB_ELEMS / 64), dim3(64 * WR_GROUPS), 0, st, (const float*)workspace, n2,
                     dw, dbias, d->c_out, d->c_in);
  return check_launch("pv_conv3d_bwd_weight_f16");
}

#ifdef PV_DIAG_STAMPS
int pv_diag_read_wgrad(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(wgrad_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
