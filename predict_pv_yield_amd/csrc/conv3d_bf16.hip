// bf16 MFMA Conv3D (3x3x3, stride 1) for gfx950 — the throughput path.
// replaces: F.conv3d + F.relu forward and dgrad as called by
//   predict_pv_yield/models/conv3d/model.py:80-90,117-120 (and model_sat_nwp.py:102-115 via pad_t = 1).
//
// Layout: activations NDHWC bf16 with the channel count padded to CPAD (16 or 32): one voxel is a
// 32/64-byte vector, which is exactly the k-slice an MFMA operand lane wants.
//
// Kernel structure (conv3d_fwd_bf16_kernel):
//   * one workgroup (4 waves, one per SIMD, 1 WG per CU) owns an output tile of 8 rows x 64 columns
//     of one sample and MARCHES OVER TIME: three input time-slices (10 rows x 64 voxels each) live
//     in an LDS ring, so every input voxel is fetched once per tile and reused by all 27 taps;
//   * the next slice is prefetched global->registers while the MFMAs of the current one run
//     (issue-early / write-late), 2 barriers per output slice;
//   * out[cout, voxel] = sum_k W[cout, k] * X[k, voxel]: weights are the MFMA A operand and stay in
//     registers for the whole kernel (27 x CPAD/16 fragments = 216 VGPR/AGPR at CPAD = 32, which is
//     why the kernel runs one wave per SIMD with the full 512-register file); activations are the B
//     operand, one ds_read_b128 per 32 voxels x 8 channels, reused by the 3 kh taps that share an
//     input row (a wave owns 4 output rows x 32 columns);
//   * the LDS image is XOR-swizzled on 16-byte chunks so that the 16 lanes of a ds_read_b128 group
//     hit 16 distinct chunks of the 256-byte bank row for every tap shift;
//   * bias is the initial accumulator, ReLU + bf16 rounding are fused into the epilogue; the last
//     layer can write NCDHW (the flatten order fc1 expects, model.py:122).
// dgrad is the same kernel on dy (zero padding 2-p, ReLU gate applied while staging) with
// channel-swapped, mirrored weights (pv_conv3d_pack_weight_bf16(..., transpose_flip = 1)).
#include "pv_common.h"

namespace pv {

constexpr int TR = 8;          // output rows per workgroup tile
constexpr int TRI = TR + 2;    // input rows per slice
constexpr int TW = 64;         // columns per LDS row (62 valid output columns per tile)
constexpr int TW_VALID = TW - 2;

#ifdef PV_DIAG_STAMPS
__device__ unsigned long long v1_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif

__device__ __forceinline__ uint32_t gate_word(uint32_t x, uint32_t g) {
  // keep each bf16 half of x only where the matching half of g is > 0
  uint32_t lo = ((g & 0x7fffu) != 0u && (g & 0x8000u) == 0u) ? 0x0000ffffu : 0u;
  uint32_t hi = ((g & 0x7fff0000u) != 0u && (g & 0x80000000u) == 0u) ? 0xffff0000u : 0u;
  return x & (lo | hi);
}

template <int CPAD>
struct SliceGeom {
  static constexpr int NCH = CPAD / 8;            // 16-byte chunks per voxel
  static constexpr int VPR = 16 / NCH;            // voxels per 256-byte bank row
  static constexpr int VOX_BYTES = CPAD * 2;
  static constexpr int ROW_BYTES = TW * VOX_BYTES;
  static constexpr int SLOT_BYTES = TRI * ROW_BYTES;
  static constexpr int CHUNKS = TRI * TW * NCH;   // 16-byte chunks per slice
  static constexpr int NLOAD = CHUNKS / 256;      // chunks per thread
  __device__ static __forceinline__ int swz(int v) { return (v / VPR) % NCH; }
};

// X_F32 (first layer only, pad_w == 0): x is the reference's f32 NCDHW tensor [B, c_in_real, T, H, W]; the staging
// fetches 4 consecutive voxels of each of a chunk's 8 channel planes with one dwordx4 load, rounds to bf16 on the way
// into LDS and -- for the voxels this workgroup owns -- also writes the NDHWC bf16 image the weight-gradient kernel
// will read (xp_out).  That replaces the separate pack pass over the input (read 104 MB + write 75 MB + read 75 MB at B = 32).
template <int CPAD, bool HAS_GATE, bool Y_NCDHW, bool OUT_GATE, bool X_F32 = false>
__global__ __launch_bounds__(256, 1) void conv3d_fwd_bf16_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ gate, const uint16_t* __restrict__ wp,
    const float* __restrict__ bias, uint16_t* __restrict__ y, const uint16_t* __restrict__ out_gate, int t_in, int h_in, int w_in, int t_out,
    int h_out, int w_out, int pad_t, int pad_h, int pad_w, int relu, int n_colblk,
    int t_chunk, int c_out, int c_in_real = 0, uint16_t* __restrict__ xp_out = nullptr,
    uint32_t* __restrict__ mask_out = nullptr) {
  using G = SliceGeom<CPAD>;
  constexpr int KS = CPAD / 16;
  // ring of 3 slices | 256 B of zeros (tap reads of masked columns run 2 voxels past a slot) | 32 bias floats
  // + (NDHWC epilogue) 8 KB per wave to transpose the accumulators into whole 1-KB output lines
  constexpr int EPI_BYTES = Y_NCDHW ? 0 : 4 * 8192;
  __shared__ __attribute__((aligned(256))) unsigned char lds[3 * G::SLOT_BYTES + 512 + EPI_BYTES];
  float* lds_bias = reinterpret_cast<float*>(lds + 3 * G::SLOT_BYTES + 256);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int ws = wave & 1;   // column segment (32 columns)
  const int wr = wave >> 1;  // row half (4 rows)

  const int rowblk = blockIdx.x / n_colblk;
  const int colblk = blockIdx.x - rowblk * n_colblk;
  const int h0 = rowblk * TR;          // first output row of the tile
  const int w0 = colblk * TW_VALID;    // first output column of the tile
  const int b = blockIdx.z;
  const int tc0 = blockIdx.y * t_chunk;
  const int tc1 = min(tc0 + t_chunk, t_out);
  if (tc0 >= tc1) return;

  if (tid < 64) reinterpret_cast<uint32_t*>(lds + 3 * G::SLOT_BYTES)[tid] = 0u;
  if (tid < 32) lds_bias[tid] = (bias && tid < c_out) ? bias[tid] : 0.f;

  // ---- weights: A fragments, resident in registers ------------------------------------------
  bf16x8 wfrag[27][KS];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wfrag[tap][ks] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)(tap * KS + ks) * 64 + lane) * 8);

  // ---- per-lane LDS read offsets (bytes inside a slot row 0) --------------------------------
  int voff[3][KS];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    int v = 32 * ws + r + kw;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) voff[kw][ks] = v * G::VOX_BYTES + (((ks * 2 + hh) ^ G::swz(v)) << 4);
  }

  // ---- staging geometry: chunk id = i*256 + tid; one i covers RPI rows, so the row (and the whole
  // validity test except the column) is wave-uniform and the loads need NO branch: out-of-range taps are
  // loaded from a clamped address and zeroed by a select, so all NLOAD loads issue back to back. ---------
  constexpr int CPR = TW * G::NCH;   // chunks per tile row: 256 (CPAD 32) / 128 (CPAD 16)
  constexpr int RPI = 256 / CPR;     // tile rows per i: 1 / 2
  const int srow0 = RPI == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid / CPR);  // wave-uniform -> SGPR
  const int srem = tid - srow0 * CPR;
  const int scol = srem / G::NCH, sc = srem - scol * G::NCH;
  const int swi = w0 - pad_w + scol;
  const bool scol_ok = (unsigned)swi < (unsigned)w_in;
  const int lds_lane = scol * G::VOX_BYTES + ((sc ^ G::swz(scol)) << 4);        // swz depends on the column only
  // Raw buffer loads with hardware range checking: a tap outside the image (padding, halo beyond the edge)
  // gets an offset >= num_records and comes back as zeros -- no select, no branch, no VALU on the loaded
  // data, so the NLOAD loads of a slice stay in flight under the MFMAs until store_slice needs them.
  constexpr uint32_t INVALID = 0x40000000u;                                      // launcher checks sample bytes <= 2^30
  const uint32_t x_plane_b = (uint32_t)h_in * w_in * CPAD * 2u;                  // bytes per (b, t) slice
  const uint32_t x_row_b = (uint32_t)w_in * CPAD * 2u;
  const size_t sample_elems = (size_t)t_in * h_in * w_in * CPAD;
  const size_t f32_sample_b = (size_t)c_in_real * t_in * h_in * w_in * 4;       // X_F32: bytes of one NCDHW sample
  const __amdgpu_buffer_rsrc_t xrsrc =
      X_F32 ? __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(x) + (size_t)b * f32_sample_b),
                                                0, (int)f32_sample_b, 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * sample_elems), 0, (int)(sample_elems * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((HAS_GATE ? gate : x) + (size_t)b * sample_elems), 0, (int)(sample_elems * 2), 0x00020000);
  const uint32_t lane_voff = scol_ok ? (uint32_t)(swi * CPAD + sc * 8) * 2u : INVALID;
  // X_F32 staging tasks: (tile row 0..9) x (quad of 4 consecutive voxels) = 160 tasks of ALL channels, 40 per wave (lanes
  // 0..39 of every wave), so the four waves stage the same amount.  (The first version dealt 320 half-channel tasks over
  // 256 threads: wave 0 took two of them per lane and everyone waited for it at the barrier.)  A task = one dwordx4 load
  // per real channel (4 voxels of its plane, pad_w == 0) -> 4 voxels x 2 chunks of 8 channels.
  const uint32_t f_plane_b = (uint32_t)t_in * h_in * w_in * 4u, f_slice_b = (uint32_t)h_in * w_in * 4u, f_row_b = (uint32_t)w_in * 4u;
  constexpr int F_CH = 16;                      // channel slots of a task (c_in_real <= 16 are loaded)
  const int f_lane = tid & 63;
  const int f_task = (tid >> 6) * 40 + f_lane;  // valid for f_lane < 40
  const bool f_has = X_F32 && f_lane < 40;
  const int f_row = f_task >> 4;                // 0..9
  const int f_col = 4 * (f_task & 15);          // first tile column of the quad
  const bool own_last_row = (h0 + TR >= h_out), own_last_col = (w0 + TW_VALID >= w_out), own_last_t = (tc1 >= t_out);

  u32x4 stage[X_F32 ? 1 : G::NLOAD];
  f32x4 stage_f[X_F32 ? F_CH : 1];
  u32x4 stage_g[HAS_GATE ? G::NLOAD : 1];
  auto load_slice = [&](int s) {
    // slice index s = input time + pad_t  (s in [tc0, tc1 + 2))
    const int ti = s - pad_t;
    const bool t_ok = (unsigned)ti < (unsigned)t_in;
    if constexpr (X_F32) {
      const uint32_t toff = t_ok ? (uint32_t)ti * f_slice_b : INVALID;
      if (f_has) {
        const int hi = h0 - pad_h + f_row;
        const bool ok = (unsigned)hi < (unsigned)h_in;
        const uint32_t base = ok ? toff + (uint32_t)hi * f_row_b + (uint32_t)(w0 + f_col) * 4u : INVALID;
#pragma unroll
        for (int ch = 0; ch < F_CH; ++ch) {
          if (ch < c_in_real) {        // uniform: absent channels are never fetched
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, base + (uint32_t)ch * f_plane_b, 0, 0);
            stage_f[ch] = __builtin_bit_cast(f32x4, v);
          } else {
            stage_f[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    } else {
      const uint32_t toff = (uint32_t)min(max(ti, 0), t_in - 1) * x_plane_b;
#pragma unroll
      for (int i = 0; i < G::NLOAD; ++i) {
        const int hi = h0 - pad_h + i * RPI + srow0;
        const bool row_ok = t_ok && (unsigned)hi < (unsigned)h_in;                 // scalar
        const uint32_t srow = toff + (uint32_t)min(max(hi, 0), h_in - 1) * x_row_b + (row_ok ? 0u : INVALID);
        const uint32_t voff = lane_voff + srow;                                    // < 2^32, >= 2^30 if anything is invalid
        stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0);
        if constexpr (HAS_GATE) stage_g[i] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, voff, 0, 0);
      }
    }
  };
  auto store_slice = [&](int s) {
    if constexpr (X_F32) {
      unsigned char* slot = lds + (s % 3) * G::SLOT_BYTES;
      if (f_has) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int scol = f_col + q, swi = w0 + scol;
          const bool col_ok = swi < w_in;   // a quad may straddle the right image edge: those voxels are zero
#pragma unroll
          for (int sc = 0; sc < 2; ++sc) {
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pack_bf16_pair(stage_f[8 * sc + 2 * e][q], stage_f[8 * sc + 2 * e + 1][q]);
            if (!col_ok) v = (u32x4){0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(slot + f_row * G::ROW_BYTES + scol * G::VOX_BYTES + ((sc ^ G::swz(scol)) << 4)) = v;
          }
        }
      }
    } else {
      unsigned char* slot = lds + (s % 3) * G::SLOT_BYTES + srow0 * G::ROW_BYTES + lds_lane;
#pragma unroll
      for (int i = 0; i < G::NLOAD; ++i) {
        u32x4 v = stage[i];
        if constexpr (HAS_GATE) {  // ReLU gate of dgrad, applied only now (after the loads have landed)
          const u32x4 g = stage_g[i];
          v[0] = gate_word(v[0], g[0]); v[1] = gate_word(v[1], g[1]);
          v[2] = gate_word(v[2], g[2]); v[3] = gate_word(v[3], g[3]);
        }
        *reinterpret_cast<u32x4*>(slot + i * RPI * G::ROW_BYTES) = v;
      }
    }
  };

  // X_F32: the NDHWC bf16 image for the weight gradient leaves from the staged LDS slot, lane-contiguous (a tile row of
  // 64 voxels x 32 B is one 2 KB run of xp).  Every input voxel is written by exactly one workgroup: the tile it belongs
  // to as an OUTPUT position; the last row / column block and time chunk also own the 2-voxel rim.
  auto write_xp = [&](int s) {
    if constexpr (X_F32) {
      const int ti = s - pad_t;
      if (!xp_out || (unsigned)ti >= (unsigned)t_in || !(s < tc1 || own_last_t)) return;
      const unsigned char* slot = lds + (s % 3) * G::SLOT_BYTES;
#pragma unroll
      for (int i = 0; i < G::NLOAD; ++i) {
        const int rowi = i * RPI + srow0, hi = h0 - pad_h + rowi;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slot + rowi * G::ROW_BYTES + lds_lane);
        if ((unsigned)hi < (unsigned)h_in && scol_ok && (rowi < TR || own_last_row) && (scol < TW_VALID || own_last_col))
          *reinterpret_cast<u32x4*>(xp_out + ((((size_t)b * t_in + ti) * h_in + hi) * w_in + swi) * CPAD + sc * 8) = v;
      }
    }
  };

  // ---- prologue: two slices into the ring, third in flight ------------------------------------
  load_slice(tc0);
  store_slice(tc0);
  load_slice(tc0 + 1);
  store_slice(tc0 + 1);
  load_slice(tc0 + 2);

  const int plane_out = h_out * w_out;
  // per-lane, t-independent store offsets (bytes) and validity of the 4 output rows of this wave
  uint32_t st_off[4];
  bool st_ok[4];
  {
    const int col_t = 32 * ws + r;  // column inside the tile
    const int wo = w0 + col_t;
    const bool col_ok = col_t < TW_VALID && wo < w_out;
#pragma unroll
    for (int orow = 0; orow < 4; ++orow) {
      const int ho = h0 + 4 * wr + orow;
      st_ok[orow] = col_ok && ho < h_out;
      const uint32_t vox = (uint32_t)(ho * w_out + wo);
      if constexpr (!Y_NCDHW) st_off[orow] = (vox * 32u + 4u * hh) * 2u;
      else st_off[orow] = (vox + (uint32_t)(4 * hh) * (uint32_t)(t_out * plane_out)) * 2u;
    }
  }
  const __amdgpu_buffer_rsrc_t ogrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((OUT_GATE ? out_gate : y) + (size_t)b * t_out * plane_out * 32), 0, (int)((size_t)t_out * plane_out * 64), 0x00020000);
  // NDHWC write-out geometry: lane -> (voxel 16*half + lane/4, chunk lane%4) of the wave's 32-column segment
  uint32_t wr_off[4];
  bool wr_ok[4][2];
#pragma unroll
  for (int orow = 0; orow < 4; ++orow) {
    const int ho = h0 + 4 * wr + orow;
    const int wo0 = w0 + 32 * ws + (lane >> 2);
    wr_off[orow] = ((uint32_t)(ho * w_out + wo0) * 32u + 8u * (lane & 3)) * 2u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col_t = 32 * ws + 16 * half + (lane >> 2);
      wr_ok[orow][half] = ho < h_out && col_t < TW_VALID && (w0 + col_t) < w_out;
    }
  }
  // NDHWC write-out of the tile staged in the wave's private LDS area: each lane moves two 16-byte chunks per output
  // row, a wave-instruction covers 1 KB of contiguous global memory.  It is DEFERRED by one iteration and issued
  // inside the next slice's MFMA stream (the accumulators are free again once the tile is staged), so its LDS
  // round trip and store issue hide under matrix work instead of sitting between two barriers.
  auto write_out = [&](int tw) {
    unsigned char* yt = reinterpret_cast<unsigned char*>(y + ((size_t)b * t_out + tw) * plane_out * 32);
    const unsigned char* epi_w = lds + 3 * G::SLOT_BYTES + 512 + wave * 8192;
    u32x4 og[OUT_GATE ? 4 : 1][2];
    if constexpr (OUT_GATE) {  // dgrad: ReLU mask of this output slice
      const uint32_t tbase = (uint32_t)tw * (uint32_t)plane_out * 64u;
#pragma unroll
      for (int orow = 0; orow < 4; ++orow)
#pragma unroll
        for (int half = 0; half < 2; ++half)
          og[orow][half] = __builtin_amdgcn_raw_buffer_load_b128(
              ogrsrc, wr_ok[orow][half] ? tbase + wr_off[orow] + half * 1024 : 0x40000000u, 0, 0);
    }
#pragma unroll
    for (int orow = 0; orow < 4; ++orow) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int v = 16 * half + (lane >> 2);       // voxel inside the 32-column segment
        const int c = lane & 3;                      // 16-byte chunk = channels 8c .. 8c+7
        const int pc = c ^ ((v >> 1) & 3);           // swizzle moves whole 16-byte chunks (even slot XOR)
        u32x4 val = *reinterpret_cast<const u32x4*>(epi_w + orow * 2048 + v * 64 + pc * 16);
        if (wr_ok[orow][half]) {
          if constexpr (OUT_GATE) {  // zero the gradient where the consumer's ReLU was inactive
            const u32x4 g = og[orow][half];
            val[0] = gate_word(val[0], g[0]); val[1] = gate_word(val[1], g[1]);
            val[2] = gate_word(val[2], g[2]); val[3] = gate_word(val[3], g[3]);
          }
          *reinterpret_cast<u32x4*>(yt + wr_off[orow] + half * 1024) = val;
        }
        if constexpr (!OUT_GATE) {
          if (mask_out) {
            // 1-bit relu mask of the output (u32 per voxel, bit = channel): this lane's 8 channels are byte c of the
            // voxel's word; the 64 lanes of the instruction write 64 consecutive bytes
            const uint32_t words[4] = {val[0], val[1], val[2], val[3]};
            const uint32_t mb = relu_bits_of_8(words);
            if (wr_ok[orow][half]) {   // planes padded to [ceil8(h_out)][ceil32(w_out)] words (pv_relu_mask_dims)
              const int mw = (w_out + 31) & ~31, mh = (h_out + 7) & ~7;
              reinterpret_cast<unsigned char*>(mask_out)[((((size_t)b * t_out + tw) * mh + (h0 + 4 * wr + orow)) * mw + w0 +
                                                          32 * ws + 16 * half + (lane >> 2)) * 4 + c] = (unsigned char)mb;
            }
          }
        }
      }
    }
  };

#ifdef PV_DIAG_STAMPS
  unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, q0, q1, q2, q3, q4, q5, q6, q7;
#endif
  for (int t = tc0; t < tc1; ++t) {
    PV_STAMP(q0);
    store_slice(t + 2);
    PV_STAMP(q1);
    __syncthreads();
    PV_STAMP(q2);
    if (t + 1 < tc1) load_slice(t + 3);  // prefetch under the MFMAs below
    PV_STAMP(q3);
    if constexpr (X_F32) {
      if (t == tc0) {
        write_xp(tc0);
        write_xp(tc0 + 1);
      }
      write_xp(t + 2);
    }
    PV_STAMP(q4);
    f32x16 acc0;  // bias as the initial accumulator: row(reg j, half hh) = (j&3) + 8*(j>>2) + 4*hh
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(lds_bias + 8 * q + 4 * hh);
      acc0[4 * q] = bq[0]; acc0[4 * q + 1] = bq[1]; acc0[4 * q + 2] = bq[2]; acc0[4 * q + 3] = bq[3];
    }
    f32x16 acc[4] = {acc0, acc0, acc0, acc0};
    // 18 (kt, kw, ks) groups of 6 operand reads + 12 MFMAs, software-pipelined by one group: the ds_read_b128s of
    // group g+1 are issued before the MFMAs of group g so their LDS latency hides under 12 x 32 MFMA cycles
    // (one wave per SIMD: nothing else would cover it).
    const unsigned char* slot_kt[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) slot_kt[kt] = lds + ((t + kt) % 3) * G::SLOT_BYTES + (4 * wr) * G::ROW_BYTES;
    constexpr int NG = 9 * KS;
    bf16x8 bfr[2][6];
    auto read_group = [&](int g, bf16x8* dst) {
      const int kt = g / (3 * KS), kw = (g / KS) % 3, ks = g % KS;
      const unsigned char* p = slot_kt[kt] + voff[kw][ks];
#pragma unroll
      for (int ir = 0; ir < 6; ++ir) dst[ir] = *reinterpret_cast<const bf16x8*>(p + ir * G::ROW_BYTES);
    };
    read_group(0, bfr[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) read_group(g + 1, bfr[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the 6 reads of group g+1 AHEAD of the 12 MFMAs of group g
      const int kt = g / (3 * KS), kw = (g / KS) % 3, ks = g % KS;
#pragma unroll
      for (int ir = 0; ir < 6; ++ir) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int orow = ir - kh;
          if (orow >= 0 && orow < 4)
            acc[orow] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kt * 9 + kh * 3 + kw][ks], bfr[g & 1][ir],
                                                                acc[orow], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!Y_NCDHW) {
        if (g == 1 && t > tc0) {  // previous slice's tile: its stores ride under the remaining 16 groups of MFMAs
          write_out(t - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    PV_STAMP(q5);
    // ---- epilogue: ReLU, bf16, store.  Addresses = scalar base of this (b, t) + a 32-bit per-lane offset
    // that does not depend on t, so nothing but 4 offsets stays live across the march. -----------------
#pragma unroll
    for (int orow = 0; orow < 4; ++orow) {
      if (!Y_NCDHW || st_ok[orow]) {
        f32x16 a = acc[orow];
        if (relu) {
#pragma unroll
          for (int j = 0; j < 16; ++j) a[j] = a[j] > 0.f ? a[j] : 0.f;
        }
        if constexpr (!Y_NCDHW) {
          // stage this 32-voxel x 32-channel tile as [voxel][channel] bf16 (2 KB) in the wave's private LDS area;
          // 8-byte slots XOR-swizzled by the voxel so the 16 lanes of a write group spread over the banks
          unsigned char* epi = lds + 3 * G::SLOT_BYTES + 512 + wave * 8192 + orow * 2048;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u32x2 o;
            o[0] = pack_bf16_pair(a[4 * q], a[4 * q + 1]);
            o[1] = pack_bf16_pair(a[4 * q + 2], a[4 * q + 3]);
            const int slot = (2 * q + hh) ^ (((r >> 1) & 3) << 1);
            *reinterpret_cast<u32x2*>(epi + r * 64 + slot * 8) = o;
          }
        } else {
          const size_t cstride = (size_t)t_out * plane_out;  // elements between channels
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int cj = (j & 3) + 8 * (j >> 2);           // + 4*hh folded into st_off
            unsigned char* yc = reinterpret_cast<unsigned char*>(y + ((size_t)b * c_out + cj) * cstride + (size_t)t * plane_out);
            if (cj + 4 * hh < c_out) *reinterpret_cast<uint16_t*>(yc + st_off[orow]) = f32_to_bf16_bits(a[j]);
          }
        }
      }
    }
    PV_STAMP(q6);
    __syncthreads();  // every wave is done reading slot t%3 before the next store_slice(t+3)
#ifdef PV_DIAG_STAMPS
    PV_STAMP(q7);
    dg[0] += q1 - q0;  // wait for the prefetched slice + convert + LDS writes
    dg[1] += q2 - q1;  // barrier
    dg[2] += q3 - q2;  // issue of the next slice's loads
    dg[3] += q4 - q3;  // xp copy (LDS -> global)
    dg[4] += q5 - q4;  // MFMA groups (+ deferred write-out of the previous tile)
    dg[5] += q6 - q5;  // epilogue: relu, pack, LDS transpose writes
    dg[6] += q7 - q6;  // barrier
    dg[7] += 1;
#endif
  }
  if constexpr (!Y_NCDHW) write_out(tc1 - 1);
#ifdef PV_DIAG_STAMPS
  {
    const int wgl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (lane == 0 && wgl * 4 + wave < PV_DIAG_WAVES)
      for (int i = 0; i < PV_DIAG_SLOTS; ++i) v1_diag[(size_t)(wgl * 4 + wave) * PV_DIAG_SLOTS + i] = dg[i];
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// layout packers
// ---------------------------------------------------------------------------------------------
// x[B,C,T,H,W] f32 -> xp[B,T,H,W,CPAD] bf16; one thread = one voxel (reads are coalesced per channel
// plane, the CPAD*2-byte voxel is written with 16-byte stores)
template <int CPAD>
__global__ __launch_bounds__(256) void pack_ncdhw_to_ndhwc_kernel(const float* __restrict__ x,
                                                                   uint16_t* __restrict__ xp, int c,
                                                                   long long vox_per_sample, long long total_vox) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vox; i += stride) {
    long long bi = i / vox_per_sample;
    long long v = i - bi * vox_per_sample;
    const float* src = x + (size_t)bi * c * vox_per_sample + v;
    uint32_t words[CPAD / 2];
#pragma unroll
    for (int k = 0; k < CPAD / 2; ++k) {
      float a = (2 * k < c) ? src[(size_t)(2 * k) * vox_per_sample] : 0.f;
      float bq = (2 * k + 1 < c) ? src[(size_t)(2 * k + 1) * vox_per_sample] : 0.f;
      words[k] = pack_bf16_pair(a, bq);
    }
    u32x4* dst = reinterpret_cast<u32x4*>(xp + (size_t)i * CPAD);
#pragma unroll
    for (int k = 0; k < CPAD / 8; ++k) {
      u32x4 o = {words[4 * k], words[4 * k + 1], words[4 * k + 2], words[4 * k + 3]};
      dst[k] = o;
    }
  }
}

// A wave holds NCH 16-byte chunks per lane that belong to one contiguous output run of NCH * 1 KB (lane l owns bytes
// [l * NCH * 16, (l + 1) * NCH * 16)).  Written straight from registers every store instruction would touch 64 different
// cache lines; here the run goes through a wave-private LDS patch (XOR-swizzled rows: conflict-free both ways) and
// leaves as NCH instructions of 1 KB each, lane-contiguous.  valid_bytes clips the tail of the tensor.
template <int NCH>
__device__ __forceinline__ void wave_store_run(unsigned char* patch, const u32x4 (&o)[NCH], unsigned char* gdst,
                                               long long valid_bytes) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
    *reinterpret_cast<u32x4*>(patch + lane * (NCH * 16) + ((k ^ (lane & (NCH - 1))) << 4)) = o[k];
  __builtin_amdgcn_wave_barrier();  // LDS executes a wave's instructions in order; this only pins the compiler's order
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int g = i * 64 + lane;  // linear chunk of the run
    const int row = g / NCH, k = g % NCH;
    const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * (NCH * 16) + ((k ^ (row & (NCH - 1))) << 4));
    if ((long long)g * 16 < valid_bytes) *reinterpret_cast<u32x4*>(gdst + (size_t)g * 16) = v;
  }
  __builtin_amdgcn_wave_barrier();
}

// 4 consecutive voxels per thread: 16-byte plane reads, 4*CPAD*2 contiguous bytes written (vps % 4 == 0)
template <int CPAD>
__global__ __launch_bounds__(256) void pack_ncdhw_to_ndhwc_v4_kernel(const float* __restrict__ x,
                                                                      uint16_t* __restrict__ xp, int c,
                                                                      long long vox_per_sample, long long total_quads) {
  constexpr int NCH = 4 * CPAD * 2 / 16;  // 16-byte chunks per lane (4 voxels)
  __shared__ __attribute__((aligned(16))) unsigned char patch[4][NCH * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long qps = vox_per_sample / 4;
  for (long long base = (long long)blockIdx.x * blockDim.x + wave * 64; base < total_quads; base += stride) {
    const long long i = base + lane;
    u32x4 o[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) o[k] = (u32x4){0u, 0u, 0u, 0u};
    if (i < total_quads) {
      const long long bi = i / qps;
      const long long v = (i - bi * qps) * 4;
      const float* src = x + (size_t)bi * c * vox_per_sample + v;
      uint16_t h[4][CPAD];
#pragma unroll
      for (int k = 0; k < CPAD; ++k) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (k < c) a = *reinterpret_cast<const f32x4*>(src + (size_t)k * vox_per_sample);
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q][k] = f32_to_bf16_bits(a[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < CPAD / 8; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[q * (CPAD / 8) + k][e] = (uint32_t)h[q][8 * k + 2 * e] | ((uint32_t)h[q][8 * k + 2 * e + 1] << 16);
    }
    // quads are numbered through the whole [B][voxel] range, so the wave's 64 quads are one contiguous output run
    wave_store_run<NCH>(patch[wave], o, reinterpret_cast<unsigned char*>(xp) + (size_t)base * (NCH * 16),
                        (total_quads - base) * (NCH * 16));
  }
}

// ---- two-term f16 split (round 4): x s = h + l, h = rne_f16(x s), l = rne_f16(x s - h): 22 significant bits in two half-float
// images, so that an f32-accurate product needs THREE matrix-core launches (l h, h l, h h) where the bf16 split needs six, and
// the split pass writes 4 instead of 6 bytes per element.  Half floats have a narrow range: s = 2^(14 - e) is an exact power of
// two from the tensor's largest magnitude (max |x| < 2^e), found by a pass of its own (maxabs_f32_kernel, one atomic per wave);
// elements more than 2^11 below the largest lose low bits of l to the subnormal range -- an absolute error of 2^-39 of the
// largest element.  state[0] = bits of max |x| (in), state[1] = s, state[2] = 1 / s (out, for the caller's un-scaling).
__global__ __launch_bounds__(256) void maxabs_f32_kernel(const float* __restrict__ x, long long n4, uint32_t* __restrict__ state) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  uint32_t m = 0;
  auto fold = [&](const u32x4 v) { m = max(max(m, v[0] & 0x7fffffffu), max(max(v[1] & 0x7fffffffu, v[2] & 0x7fffffffu), v[3] & 0x7fffffffu)); };
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {      // eight independent 16-byte loads in flight per lane
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4*>(x + 4 * (i + j * stride));
#pragma unroll
    for (int j = 0; j < 8; ++j) fold(v[j]);
  }
  for (; i < n4; i += stride) fold(*reinterpret_cast<const u32x4*>(x + 4 * i));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  // one atomic per WORKGROUP (512 in all): same-address atomics serialise at the L2 channel, ~15 ns apiece -- one per wave of a
  // 2 048-workgroup grid was 120 us of this pass
  __shared__ uint32_t wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    if (m) atomicMax(state, m);
  }
}

__device__ __forceinline__ float split2_scale(const uint32_t* __restrict__ state) {
  const uint32_t bits = state[0];
  if (bits == 0u || bits >= 0x7f800000u) return 1.f;      // all zeros / a non-finite element: nothing to protect
  int e = __builtin_amdgcn_frexp_expf(__builtin_bit_cast(float, bits));      // max = f 2^e, 0.5 <= f < 1, e <= 128
  e = e < -112 ? -112 : e;      // (s and 1 / s stay normal floats; rounds 4-5 clamped e to +-100: tensors beyond 2^100 overflowed their h image)
  return ldexpf(1.f, 14 - e);
}

template <int CPAD>
__global__ __launch_bounds__(256, 2) void pack_split2_ncdhw_to_ndhwc_f16_kernel(const float* __restrict__ x,
                                                                                uint16_t* __restrict__ xp_h,
                                                                                uint16_t* __restrict__ xp_l, int c,
                                                                                long long vox_per_sample, long long total_quads,
                                                                                uint32_t* __restrict__ state) {
  constexpr int NCH = 4 * CPAD * 2 / 16;
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) unsigned char patch[4][NCH * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long qps = vox_per_sample / 4;
  const float s = split2_scale(state);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    reinterpret_cast<float*>(state)[1] = s;
    reinterpret_cast<float*>(state)[2] = 1.f / s;
  }
  for (long long base = (long long)blockIdx.x * blockDim.x + wave * 64; base < total_quads; base += stride) {
    const long long i = base + lane;
    u32x4 oh[NCH], ol[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) oh[k] = ol[k] = (u32x4){0u, 0u, 0u, 0u};
    if (i < total_quads) {
      const long long bi = i / qps;
      const long long v = (i - bi * qps) * 4;
      const float* src = x + (size_t)bi * c * vox_per_sample + v;
#pragma unroll
      for (int k = 0; k < CPAD; k += 2) {      // channels k, k + 1 -> one 32-bit word per voxel and image
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (k < c) a0 = *reinterpret_cast<const f32x4*>(src + (size_t)k * vox_per_sample);
        if (k + 1 < c) a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 1) * vox_per_sample);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2_t f = (f32x2_t){a0[q], a1[q]} * s;
          const f16x2_t h = __builtin_convertvector(f, f16x2_t);
          const f32x2_t r = f - __builtin_convertvector(h, f32x2_t);
          const f16x2_t l = __builtin_convertvector(r, f16x2_t);
          const int chunk = q * (CPAD / 8) + (k >> 3), word = (k & 7) >> 1;
          oh[chunk][word] = __builtin_bit_cast(uint32_t, h);
          ol[chunk][word] = __builtin_bit_cast(uint32_t, l);
        }
      }
    }
    const size_t off = (size_t)base * (NCH * 16);
    const long long valid = (total_quads - base) * (NCH * 16);
    wave_store_run<NCH>(patch[wave], oh, reinterpret_cast<unsigned char*>(xp_h) + off, valid);
    wave_store_run<NCH>(patch[wave], ol, reinterpret_cast<unsigned char*>(xp_l) + off, valid);
  }
}

template <int CPAD>
__global__ __launch_bounds__(256) void unpack_ndhwc_to_ncdhw_kernel(const uint16_t* __restrict__ xp,
                                                                     float* __restrict__ x, int c,
                                                                     long long vox_per_sample, long long total_vox) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vox; i += stride) {
    long long bi = i / vox_per_sample;
    long long v = i - bi * vox_per_sample;
    const uint16_t* src = xp + (size_t)i * CPAD;
    float* dst = x + (size_t)bi * c * vox_per_sample + v;
    for (int k = 0; k < c; ++k) dst[(size_t)k * vox_per_sample] = bf16_bits_to_f32(src[k]);
  }
}

// w[Co,Ci,27] f32 -> A fragments [27][KS][64 lanes][8] bf16:
//   lane (r = lane&31, h = lane>>5), element j  <-  W[cout = r][cin = ks*16 + 8*h + j][tap]
// transpose_flip: the dgrad operator, W'[cout' = ci][cin' = co][tap] = W[co][ci][26 - tap]
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp,
                                                           int c_out, int c_in, int ks_count, int transpose_flip) {
  const int total = 27 * ks_count * 64 * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int j = i & 7;
    int lane = (i >> 3) & 63;
    int ks = (i >> 9) % ks_count;
    int tap = (i >> 9) / ks_count;
    int row = lane & 31;                       // MFMA A row = output channel of this operator
    int k = ks * 16 + 8 * (lane >> 5) + j;     // contraction index = input channel of this operator
    float v = 0.f;
    if (!transpose_flip) {
      if (row < c_out && k < c_in) v = w[((size_t)row * c_in + k) * 27 + tap];
    } else {
      // operator maps c_out (dy channels, k) -> c_in (dx channels, row)
      if (row < c_in && k < c_out) v = w[((size_t)k * c_in + row) * 27 + (26 - tap)];
    }
    wp[i] = f32_to_bf16_bits(v);
  }
}

// all (weight, orientation) jobs of a model in one launch: 108 blocks per job (54 for the 32x32x16 fragments, 54 for the
// 16x16x32 fragments of the input-stationary kernel (conv3d_bf16_v3.hip), which exist only for 32 contraction channels)
struct PackTable {
  pv_pack_job job[PV_PACK_MAX_JOBS];
};
constexpr int PACK_BLOCKS_PER_JOB = 108;

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(PackTable tab) {
  const pv_pack_job J = tab.job[blockIdx.x / PACK_BLOCKS_PER_JOB];
  const int blk = blockIdx.x % PACK_BLOCKS_PER_JOB;
  const int kch = J.transpose_flip ? J.c_out : J.c_in;
  const int ks_count = kch <= 16 ? 1 : 2;
  const float* __restrict__ w = J.w;
  if (blk < 54) {
    const int total = 27 * ks_count * 64 * 8;
    for (int i = blk * 256 + threadIdx.x; i < total; i += 54 * 256) {
      const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % ks_count, tap = (i >> 9) / ks_count;
      const int row = lane & 31, k = ks * 16 + 8 * (lane >> 5) + j;
      float v = 0.f;
      if (!J.transpose_flip) {
        if (row < J.c_out && k < J.c_in) v = w[((size_t)row * J.c_in + k) * 27 + tap];
      } else {
        if (row < J.c_in && k < J.c_out) v = w[((size_t)k * J.c_in + row) * 27 + (26 - tap)];
      }
      J.wp[i] = f32_to_bf16_bits(v);
    }
  } else if (ks_count == 2) {
    uint16_t* wp2 = J.wp + (size_t)27 * 2 * 64 * 8;
    const int total = 27 * 2 * 64 * 8;
    for (int i = (blk - 54) * 256 + threadIdx.x; i < total; i += 54 * 256) {
      const int j = i & 7, lane = (i >> 3) & 63, half = (i >> 9) & 1, tap = i >> 10;
      const int row = 16 * half + (lane & 15), k = 8 * (lane >> 4) + j;
      float v = 0.f;
      if (!J.transpose_flip) {
        if (row < J.c_out && k < J.c_in) v = w[((size_t)row * J.c_in + k) * 27 + tap];
      } else {
        if (row < J.c_in && k < J.c_out) v = w[((size_t)k * J.c_in + row) * 27 + (26 - tap)];
      }
      wp2[i] = f32_to_bf16_bits(v);
    }
  }
}

// dy_eff[B,T,H,W,32] bf16 = NDHWC( dy[B,32,T,H,W] ⊙ (y[B,32,T,H,W] > 0) ), both bf16 NCDHW.
// Used to bring fc1's input gradient (flatten order) back to the conv layout.
__global__ __launch_bounds__(256) void repack_gate_ncdhw_to_ndhwc_bf16(const uint16_t* __restrict__ dy,
                                                                        const uint16_t* __restrict__ yv,
                                                                        uint16_t* __restrict__ out, int c,
                                                                        long long vox_per_sample, long long total_vox) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vox; i += stride) {
    long long bi = i / vox_per_sample;
    long long v = i - bi * vox_per_sample;
    size_t base = (size_t)bi * c * vox_per_sample + v;
    uint32_t words[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint32_t lo = 0, hi = 0;
      if (2 * k < c) {
        uint16_t d = dy[base + (size_t)(2 * k) * vox_per_sample];
        uint16_t g = yv ? yv[base + (size_t)(2 * k) * vox_per_sample] : (uint16_t)0x3f80;
        lo = ((g & 0x7fff) != 0 && (g & 0x8000) == 0) ? d : 0;
      }
      if (2 * k + 1 < c) {
        uint16_t d = dy[base + (size_t)(2 * k + 1) * vox_per_sample];
        uint16_t g = yv ? yv[base + (size_t)(2 * k + 1) * vox_per_sample] : (uint16_t)0x3f80;
        hi = ((g & 0x7fff) != 0 && (g & 0x8000) == 0) ? d : 0;
      }
      words[k] = lo | (hi << 16);
    }
    u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)i * 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u32x4 o = {words[4 * k], words[4 * k + 1], words[4 * k + 2], words[4 * k + 3]};
      dst[k] = o;
    }
  }
}

// 4 consecutive voxels per thread: 8-byte plane reads of dy and y, 256 contiguous bytes written (vps % 4 == 0)
__global__ __launch_bounds__(256) void repack_gate_ncdhw_to_ndhwc_bf16_v4(const uint16_t* __restrict__ dy,
                                                                           const uint16_t* __restrict__ yv,
                                                                           uint16_t* __restrict__ out, int c,
                                                                           long long vox_per_sample, long long total_quads) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[4][16 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long qps = vox_per_sample / 4;
  for (long long wbase = (long long)blockIdx.x * blockDim.x + wave * 64; wbase < total_quads; wbase += stride) {
    const long long i = wbase + lane;
    u32x4 o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = (u32x4){0u, 0u, 0u, 0u};
    if (i < total_quads) {
      const long long bi = i / qps;
      const long long v = (i - bi * qps) * 4;
      const size_t base = (size_t)bi * c * vox_per_sample + v;
      uint16_t h[4][32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        u32x2 d = {0u, 0u}, g = {0x3f803f80u, 0x3f803f80u};
        if (k < c) {
          d = *reinterpret_cast<const u32x2*>(dy + base + (size_t)k * vox_per_sample);
          if (yv) g = *reinterpret_cast<const u32x2*>(yv + base + (size_t)k * vox_per_sample);
        }
        d[0] = gate_word(d[0], g[0]);
        d[1] = gate_word(d[1], g[1]);
        h[0][k] = (uint16_t)(d[0] & 0xffffu); h[1][k] = (uint16_t)(d[0] >> 16);
        h[2][k] = (uint16_t)(d[1] & 0xffffu); h[3][k] = (uint16_t)(d[1] >> 16);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[q * 4 + k][e] = (uint32_t)h[q][8 * k + 2 * e] | ((uint32_t)h[q][8 * k + 2 * e + 1] << 16);
    }
    wave_store_run<16>(patch[wave], o, reinterpret_cast<unsigned char*>(out) + (size_t)wbase * 256, (total_quads - wbase) * 256);
  }
}

// relu_mask[v] (u32 per voxel, bit c = channel c > 0) from an NDHWC bf16 [.., 32] activation: the fallback producer for
// launches whose kernel does not emit the mask itself (v3 kernel, gated v1 variants).  4 lanes per voxel.
__global__ __launch_bounds__(256) void relu_mask_from_ndhwc32_kernel(const uint16_t* __restrict__ y, uint32_t* __restrict__ mask,
                                                                     long long total_vox, int h, int w) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int mw = (w + 31) & ~31, mh = (h + 7) & ~7;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vox * 4; i += stride) {
    const int c = (int)(i & 3);
    const u32x4 val = *reinterpret_cast<const u32x4*>(y + (size_t)i * 8);
    const uint32_t words[4] = {val[0], val[1], val[2], val[3]};
    const long long v = i >> 2, bt = v / ((long long)h * w);
    const int rem = (int)(v - bt * h * w), hy = rem / w;
    reinterpret_cast<unsigned char*>(mask)[((bt * mh + hy) * mw + (rem - hy * w)) * 4 + c] = (unsigned char)relu_bits_of_8(words);
  }
}

// input-stationary variant (conv3d_bf16_v3.hip, its own 8 x 32 tiling); 1 = shape not covered
int launch_conv3d_fwd_bf16_v3(const uint16_t* x, const uint16_t* wp2, const float* bias, uint16_t* y,
                              const uint16_t* out_gate, const pv_conv3d_dims* d, int to, int ho, int wo, int relu,
                              int y_ncdhw, hipStream_t st, const uint32_t* out_gate_mask, uint32_t* mask_out);
bool v3_writes_mask(int y_ncdhw, const void* out_gate, const void* out_gate_mask);
// loader-wave first layer (conv3d_bf16_first.hip); 1 = request not covered
int launch_conv3d_first_f32in(const float* x, uint16_t* xp_out, const uint16_t* wp, const float* bias, uint16_t* y,
                              const pv_conv3d_dims* d, int to, int ho, int wo, int relu, hipStream_t st, uint32_t* mask_out);
void launch_pack_weight_v3(const float* w, uint16_t* wp2, int c_out, int c_in, int transpose_flip, hipStream_t st);
constexpr size_t V3_WEIGHT_ELEMS = (size_t)27 * 2 * 64 * 8;   // the 16x16x32 fragments of conv3d_bf16_v3.hip

}  // namespace pv

using namespace pv;

extern "C" {

int pv_relu_mask_dims(int32_t h, int32_t w, int32_t* hp, int32_t* wp) {
  PV_REQUIRE(h > 0 && w > 0 && hp && wp, PV_EINVAL, "pv_relu_mask_dims: bad arguments");
  *hp = (h + 7) & ~7;    // whole 8 x 32 output tiles of the conv kernels
  *wp = (w + 31) & ~31;
  return PV_OK;
}

int pv_bf16_cpad(int32_t c) {
  if (c <= 0) return PV_EINVAL;
  if (c <= 16) return 16;
  if (c <= 32) return 32;
  return PV_ESIZE;
}

int pv_pack_split2_ncdhw_f32_to_ndhwc_f16(const float* x, uint16_t* xp_h, uint16_t* xp_l, float* state, int32_t have_max, int32_t batch,
                                          int32_t c, int32_t t, int32_t h, int32_t w, void* stream) {
  PV_REQUIRE(x && xp_h && xp_l && state, PV_EINVAL, "pv_pack_split2_ncdhw_f32_to_ndhwc_f16: null pointer");
  int cpad = pv_bf16_cpad(c);
  PV_REQUIRE(cpad > 0, PV_ESIZE, "pv_pack_split2_ncdhw_f32_to_ndhwc_f16: c=%d not in 1..32", c);
  if (have_max & 2) cpad = 32;      // 64-byte voxels whatever c (the operand images of pv_conv3d_fwd_f16_f32out)
  have_max &= 1;
  const long long vps = (long long)t * h * w, total = vps * batch;
  PV_REQUIRE(total > 0, PV_EINVAL, "pv_pack_split2_ncdhw_f32_to_ndhwc_f16: empty tensor");
  PV_REQUIRE(vps % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)xp_h % 16 == 0) && ((uintptr_t)xp_l % 16 == 0) &&
                 ((uintptr_t)state % 4 == 0),
             PV_EINVAL, "pv_pack_split2_ncdhw_f32_to_ndhwc_f16: needs t*h*w %% 4 == 0 and 16-byte aligned buffers");
  hipStream_t st = as_stream(stream);
  uint32_t* st_bits = reinterpret_cast<uint32_t*>(state);
  if (!have_max) {      // (have_max: state[0] already holds the bits of max |x| -- pv_relu_gate_max_f32 left them there)
    PV_REQUIRE(hipMemsetAsync(st_bits, 0, sizeof(uint32_t), st) == hipSuccess, PV_ELAUNCH, "pv_pack_split2_ncdhw_f32_to_ndhwc_f16: memset failed");
    const long long n4 = total * c / 4;
    hipLaunchKernelGGL(maxabs_f32_kernel, dim3(std::min<unsigned>(stream_grid((size_t)n4, 256), 2 * kNumCU)), dim3(256), 0, st, x, n4, st_bits);
  }
  const unsigned g4 = stream_grid((size_t)(total / 4), 256);
  if (cpad == 16)
    hipLaunchKernelGGL(pack_split2_ncdhw_to_ndhwc_f16_kernel<16>, dim3(g4), dim3(256), 0, st, x, xp_h, xp_l, c, vps, total / 4, st_bits);
  else
    hipLaunchKernelGGL(pack_split2_ncdhw_to_ndhwc_f16_kernel<32>, dim3(g4), dim3(256), 0, st, x, xp_h, xp_l, c, vps, total / 4, st_bits);
  return check_launch("pv_pack_split2_ncdhw_f32_to_ndhwc_f16");
}

int pv_pack_ncdhw_f32_to_ndhwc_bf16(const float* x, uint16_t* xp, int32_t batch, int32_t c, int32_t t, int32_t h,
                                    int32_t w, void* stream) {
  PV_REQUIRE(x && xp, PV_EINVAL, "pv_pack_ncdhw_f32_to_ndhwc_bf16: null pointer");
  int cpad = pv_bf16_cpad(c);
  PV_REQUIRE(cpad > 0, PV_ESIZE, "pv_pack_ncdhw_f32_to_ndhwc_bf16: c=%d not in 1..32", c);
  long long vps = (long long)t * h * w, total = vps * batch;
  PV_REQUIRE(total > 0, PV_EINVAL, "pv_pack_ncdhw_f32_to_ndhwc_bf16: empty tensor");
  unsigned grid = stream_grid((size_t)total, 256);
  if (vps % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)xp % 16 == 0)) {
    unsigned g4 = stream_grid((size_t)(total / 4), 256);
    if (cpad == 16)
      hipLaunchKernelGGL(pack_ncdhw_to_ndhwc_v4_kernel<16>, dim3(g4), dim3(256), 0, as_stream(stream), x, xp, c, vps, total / 4);
    else
      hipLaunchKernelGGL(pack_ncdhw_to_ndhwc_v4_kernel<32>, dim3(g4), dim3(256), 0, as_stream(stream), x, xp, c, vps, total / 4);
  } else if (cpad == 16)
    hipLaunchKernelGGL(pack_ncdhw_to_ndhwc_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), x, xp, c, vps, total);
  else
    hipLaunchKernelGGL(pack_ncdhw_to_ndhwc_kernel<32>, dim3(grid), dim3(256), 0, as_stream(stream), x, xp, c, vps, total);
  return check_launch("pv_pack_ncdhw_f32_to_ndhwc_bf16");
}

int pv_unpack_ndhwc_bf16_to_ncdhw_f32(const uint16_t* xp, float* x, int32_t batch, int32_t c, int32_t t, int32_t h,
                                      int32_t w, void* stream) {
  PV_REQUIRE(x && xp, PV_EINVAL, "pv_unpack_ndhwc_bf16_to_ncdhw_f32: null pointer");
  int cpad = pv_bf16_cpad(c);
  PV_REQUIRE(cpad > 0, PV_ESIZE, "pv_unpack_ndhwc_bf16_to_ncdhw_f32: c=%d not in 1..32", c);
  long long vps = (long long)t * h * w, total = vps * batch;
  PV_REQUIRE(total > 0, PV_EINVAL, "pv_unpack_ndhwc_bf16_to_ncdhw_f32: empty tensor");
  unsigned grid = stream_grid((size_t)total, 256);
  if (cpad == 16)
    hipLaunchKernelGGL(unpack_ndhwc_to_ncdhw_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), xp, x, c, vps, total);
  else
    hipLaunchKernelGGL(unpack_ndhwc_to_ncdhw_kernel<32>, dim3(grid), dim3(256), 0, as_stream(stream), xp, x, c, vps, total);
  return check_launch("pv_unpack_ndhwc_bf16_to_ncdhw_f32");
}

int pv_repack_gate_ncdhw_to_ndhwc_bf16(const uint16_t* dy, const uint16_t* y_relu_mask, uint16_t* out, int32_t batch,
                                       int32_t c, int32_t t, int32_t h, int32_t w, void* stream) {
  PV_REQUIRE(dy && out, PV_EINVAL, "pv_repack_gate_ncdhw_to_ndhwc_bf16: null pointer");
  PV_REQUIRE(c > 0 && c <= 32, PV_ESIZE, "pv_repack_gate_ncdhw_to_ndhwc_bf16: c=%d not in 1..32", c);
  long long vps = (long long)t * h * w, total = vps * batch;
  PV_REQUIRE(total > 0, PV_EINVAL, "pv_repack_gate_ncdhw_to_ndhwc_bf16: empty tensor");
  if (vps % 4 == 0 && ((uintptr_t)dy % 8 == 0) && ((uintptr_t)y_relu_mask % 8 == 0))
    hipLaunchKernelGGL(repack_gate_ncdhw_to_ndhwc_bf16_v4, dim3(stream_grid((size_t)(total / 4), 256)), dim3(256), 0,
                       as_stream(stream), dy, y_relu_mask, out, c, vps, total / 4);
  else
    hipLaunchKernelGGL(repack_gate_ncdhw_to_ndhwc_bf16, dim3(stream_grid((size_t)total, 256)), dim3(256), 0,
                       as_stream(stream), dy, y_relu_mask, out, c, vps, total);
  return check_launch("pv_repack_gate_ncdhw_to_ndhwc_bf16");
}

size_t pv_conv3d_packed_weight_elems(int32_t k_channels) {
  int cpad = pv_bf16_cpad(k_channels);
  if (cpad <= 0) return 0;
  // v1 fragments (32x32x16 A operands) followed by the v3 fragments (16x16x32, 32-channel contractions only)
  return (size_t)27 * (cpad / 16) * 64 * 8 + (cpad == 32 ? V3_WEIGHT_ELEMS : 0);
}

int pv_conv3d_pack_weight_bf16(const float* w, uint16_t* wp, int32_t c_out, int32_t c_in, int transpose_flip,
                               void* stream) {
  PV_REQUIRE(w && wp, PV_EINVAL, "pv_conv3d_pack_weight_bf16: null pointer");
  PV_REQUIRE(c_out > 0 && c_out <= 32 && c_in > 0 && c_in <= 32, PV_ESIZE,
             "pv_conv3d_pack_weight_bf16: channels (%d,%d) must be in 1..32", c_out, c_in);
  int kch = transpose_flip ? c_out : c_in;  // contraction channels of the packed operator
  int ks = pv_bf16_cpad(kch) / 16;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(27 * ks * 2), dim3(256), 0, as_stream(stream), w, wp, c_out, c_in, ks,
                     transpose_flip ? 1 : 0);
  if (ks == 2) launch_pack_weight_v3(w, wp + (size_t)27 * 2 * 64 * 8, c_out, c_in, transpose_flip, as_stream(stream));
  return check_launch("pv_conv3d_pack_weight_bf16");
}

int pv_conv3d_pack_weights_multi_bf16(const pv_pack_job* jobs, int32_t n_jobs, void* stream) {
  PV_REQUIRE(jobs && n_jobs > 0 && n_jobs <= PV_PACK_MAX_JOBS, PV_EINVAL, "pv_conv3d_pack_weights_multi_bf16: 1..%d jobs per call",
             PV_PACK_MAX_JOBS);
  PackTable tab;
  for (int i = 0; i < n_jobs; ++i) {
    PV_REQUIRE(jobs[i].w && jobs[i].wp, PV_EINVAL, "pv_conv3d_pack_weights_multi_bf16: null pointer in job %d", i);
    PV_REQUIRE(jobs[i].c_out > 0 && jobs[i].c_out <= 32 && jobs[i].c_in > 0 && jobs[i].c_in <= 32, PV_ESIZE,
               "pv_conv3d_pack_weights_multi_bf16: channels (%d,%d) must be in 1..32", jobs[i].c_out, jobs[i].c_in);
    tab.job[i] = jobs[i];
  }
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)(n_jobs * PACK_BLOCKS_PER_JOB)), dim3(256), 0, as_stream(stream),
                     tab);
  return check_launch("pv_conv3d_pack_weights_multi_bf16");
}

int pv_conv3d_fwd_bf16(const uint16_t* x, const uint16_t* gate, const uint16_t* wp, const float* bias, uint16_t* y,
                       const uint16_t* out_gate, const uint32_t* out_gate_mask, uint32_t* relu_mask_out,
                       const pv_conv3d_dims* d, int relu, int y_ncdhw, void* stream) {
  PV_REQUIRE(d && x && wp && y, PV_EINVAL, "pv_conv3d_fwd_bf16: null pointer");
  PV_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_in <= 32 && d->c_out > 0 && d->c_out <= 32, PV_ESIZE,
             "pv_conv3d_fwd_bf16: channels (%d -> %d) must be in 1..32", d->c_in, d->c_out);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_t <= 2 && d->pad_h >= 0 && d->pad_h <= 2 && d->pad_w >= 0 && d->pad_w <= 2,
             PV_EINVAL, "pv_conv3d_fwd_bf16: padding must be 0..2");
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  PV_REQUIRE(to > 0 && ho > 0 && wo > 0, PV_ESIZE, "pv_conv3d_fwd_bf16: input smaller than the kernel");
  PV_REQUIRE(d->batch <= 65535, PV_ESIZE, "pv_conv3d_fwd_bf16: batch too large for grid.z");
  PV_REQUIRE(!(out_gate && y_ncdhw), PV_EINVAL, "pv_conv3d_fwd_bf16: out_gate needs the NDHWC output layout");
  PV_REQUIRE(!out_gate_mask || out_gate, PV_EINVAL,
             "pv_conv3d_fwd_bf16: out_gate_mask accompanies out_gate (kernels without a mask path read the bf16 tensor)");
  PV_REQUIRE(!relu_mask_out || (!y_ncdhw && d->c_out <= 32), PV_EINVAL,
             "pv_conv3d_fwd_bf16: relu_mask_out needs the NDHWC output layout");
  const int cpad = pv_bf16_cpad(d->c_in);
  PV_REQUIRE((size_t)d->t_in * d->h_in * d->w_in * cpad * 2 <= 0x40000000ull, PV_ESIZE,
             "pv_conv3d_fwd_bf16: one sample exceeds 1 GiB (buffer-addressing limit of this kernel)");
  const int n_rowblk = (ho + TR - 1) / TR;
  const int n_colblk = (wo + TW_VALID - 1) / TW_VALID;
  // split the time march only when the (sample, tile) grid alone cannot fill 256 CUs
  long long tiles = (long long)d->batch * n_rowblk * n_colblk;
  int n_tchunk = 1;
  if (tiles < 256) {
    n_tchunk = (int)((256 + tiles - 1) / tiles);
    int max_chunks = (to + 1) / 2;  // at least 2 output slices per chunk
    if (max_chunks < 1) max_chunks = 1;
    if (n_tchunk > max_chunks) n_tchunk = max_chunks;
  }
  const int t_chunk = (to + n_tchunk - 1) / n_tchunk;
  n_tchunk = (to + t_chunk - 1) / t_chunk;
  dim3 grid((unsigned)(n_rowblk * n_colblk), (unsigned)n_tchunk, (unsigned)d->batch);
  hipStream_t st = as_stream(stream);
  // relu_mask_out: written by the kernel itself where it can (v3 plain forward, v1 NDHWC epilogue), else by a pass over y
  auto mask_fallback = [&](int rc) -> int {
    if (rc != PV_OK || !relu_mask_out) return rc;
    const long long total_vox = (long long)d->batch * to * ho * wo;
    hipLaunchKernelGGL(relu_mask_from_ndhwc32_kernel, dim3(stream_grid((size_t)total_vox * 4, 256)), dim3(256), 0, st, y,
                       relu_mask_out, total_vox, ho, wo);
    return check_launch("pv_conv3d_fwd_bf16(relu mask)");
  };
  if (cpad == 32 && !gate) {  // 32 -> 32 channel layers: the input-stationary kernel (two workgroups per CU)
    const bool in_kernel = relu_mask_out && v3_writes_mask(y_ncdhw, out_gate, out_gate_mask);
    const int rc = launch_conv3d_fwd_bf16_v3(x, wp + (size_t)27 * 2 * 64 * 8, bias, y, out_gate, d, to, ho, wo, relu,
                                             y_ncdhw, st, out_gate_mask, in_kernel ? relu_mask_out : nullptr);
    if (rc != 1) return in_kernel ? rc : mask_fallback(rc);
  }
  uint32_t* v1_mask = (relu_mask_out && !out_gate && !y_ncdhw) ? relu_mask_out : nullptr;
#define PV_LAUNCH_CONV(CP, HG, YN, OG)                                                                           \
  hipLaunchKernelGGL((conv3d_fwd_bf16_kernel<CP, HG, YN, OG>), grid, dim3(256), 0, st, x, gate, wp, bias, y, out_gate, \
                     d->t_in, d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, relu ? 1 : 0, n_colblk,        \
                     t_chunk, d->c_out, 0, (uint16_t*)nullptr, v1_mask)
#define PV_LAUNCH_CONV2(CP, HG)                                      \
  do {                                                               \
    if (y_ncdhw) PV_LAUNCH_CONV(CP, HG, true, false);                \
    else if (out_gate) PV_LAUNCH_CONV(CP, HG, false, true);          \
    else PV_LAUNCH_CONV(CP, HG, false, false);                       \
  } while (0)
  if (cpad == 16) {
    if (gate) PV_LAUNCH_CONV2(16, true); else PV_LAUNCH_CONV2(16, false);
  } else {
    if (gate) PV_LAUNCH_CONV2(32, true); else PV_LAUNCH_CONV2(32, false);
  }
#undef PV_LAUNCH_CONV2
#undef PV_LAUNCH_CONV
  const int rc = check_launch("pv_conv3d_fwd_bf16");
  return v1_mask ? rc : mask_fallback(rc);
}

int pv_conv3d_fwd_bf16_f32in(const float* x, uint16_t* xp_out, const uint16_t* wp, const float* bias, uint16_t* y,
                             uint32_t* relu_mask_out, const pv_conv3d_dims* d, int relu, void* stream) {
  PV_REQUIRE(d && x && wp && y, PV_EINVAL, "pv_conv3d_fwd_bf16_f32in: null pointer");
  PV_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_in <= 16 && d->c_out > 0 && d->c_out <= 32, PV_ESIZE,
             "pv_conv3d_fwd_bf16_f32in: channels (%d -> %d) must be in 1..16 -> 1..32", d->c_in, d->c_out);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_t <= 2 && d->pad_h >= 0 && d->pad_h <= 2 && d->pad_w >= 0 && d->pad_w <= 2,
             PV_EINVAL, "pv_conv3d_fwd_bf16_f32in: padding must be 0..2");
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  PV_REQUIRE(to > 0 && ho > 0 && wo > 0, PV_ESIZE, "pv_conv3d_fwd_bf16_f32in: input smaller than the kernel");
  PV_REQUIRE(d->batch <= 65535, PV_ESIZE, "pv_conv3d_fwd_bf16_f32in: batch too large for grid.z");
  PV_REQUIRE((size_t)d->c_in * d->t_in * d->h_in * d->w_in * 4 <= 0x40000000ull, PV_ESIZE,
             "pv_conv3d_fwd_bf16_f32in: one sample exceeds 1 GiB (buffer-addressing limit of this kernel)");
  PV_REQUIRE(((uintptr_t)x % 4 == 0) && (!xp_out || (uintptr_t)xp_out % 16 == 0), PV_EINVAL,
             "pv_conv3d_fwd_bf16_f32in: unaligned operand");
  PV_REQUIRE(d->pad_w == 0, PV_EINVAL, "pv_conv3d_fwd_bf16_f32in: built for pad_w == 0 (quads of voxels start inside the image)");
  if (launch_conv3d_first_f32in(x, xp_out, wp, bias, y, d, to, ho, wo, relu, as_stream(stream), relu_mask_out) == 0)
    return check_launch("pv_conv3d_fwd_bf16_f32in(loader waves)");
  const int n_rowblk = (ho + TR - 1) / TR;
  const int n_colblk = (wo + TW_VALID - 1) / TW_VALID;
  long long tiles = (long long)d->batch * n_rowblk * n_colblk;
  int n_tchunk = 1;
  if (tiles < 256) {
    n_tchunk = (int)((256 + tiles - 1) / tiles);
    int max_chunks = (to + 1) / 2;
    if (max_chunks < 1) max_chunks = 1;
    if (n_tchunk > max_chunks) n_tchunk = max_chunks;
  }
  const int t_chunk = (to + n_tchunk - 1) / n_tchunk;
  n_tchunk = (to + t_chunk - 1) / t_chunk;
  dim3 grid((unsigned)(n_rowblk * n_colblk), (unsigned)n_tchunk, (unsigned)d->batch);
  hipLaunchKernelGGL((conv3d_fwd_bf16_kernel<16, false, false, false, true>), grid, dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint16_t*>(x), (const uint16_t*)nullptr, wp, bias, y, (const uint16_t*)nullptr,
                     d->t_in, d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, relu ? 1 : 0, n_colblk, t_chunk,
                     d->c_out, d->c_in, xp_out, relu_mask_out);
  return check_launch("pv_conv3d_fwd_bf16_f32in");
}

#ifdef PV_DIAG_STAMPS
int pv_diag_read_v1(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::v1_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
