// bf16 MFMA weight gradient of the 3x3x3 Conv3D, second form: loader waves + LDS-direct staging.
//
//   dW[co][tap][ci] = sum_voxels dY[v][co] * X[v + tap][ci]          (dY already carries its ReLU derivative)
//
// conv3d_wgrad_bf16.hip stages a slice through registers: 18 dwordx4 loads per lane right after the barrier, 18
// ds_write_b128 in front of the next one.  In-kernel stamps (tools/diag_stamps.py wgrad_v1) put a quarter of a slice into
// those two bursts and the barriers around them, and with one wave per SIMD nothing else runs meanwhile.  Dealing LDS-direct
// loads between the k-steps of the same waves (first form of this file) did not help either: every such instruction holds
// the issuing wave for ~75 clocks (13 per slice = a fifth of the slice's matrix time).  So the work is split by WAVE:
//   * waves 0..3 (one per SIMD) only multiply: transposed LDS reads (ds_read_b64_tr_b16) + MFMAs, one barrier per slice;
//   * waves 4..7 (one per SIMD, beside a multiplying wave; 212 registers each, so two waves fit a SIMD) only stage:
//     global -> LDS direct (buffer_load_dwordx4 ... lds, no staging registers, no ds_write pass), two slices ahead of the
//     multiplication, `s_waitcnt vmcnt(0)` before they join the slice barrier.  Their issue stalls cost nobody anything.
//   * the tile is 8 rows x 32 columns (X slot 10 x 34 voxels): FIVE X slots and TWO dY slots fit in LDS (142 KB), so the
//     loaders always have a slot nobody reads;
//   * a workgroup walks over ALL column tiles of its 8 rows (two for 58..64-pixel layers) with the same accumulators: one
//     slab per (sample, row block), 256 workgroups = one per CU in one round, one slab write and half the slab-reduce
//     traffic of a 32-column grid; at a tile switch the ring cannot hold the whole next triple of X slices, the loaders
//     fetch the missing one behind an extra barrier (once per tile).
// The LDS side of an LDS-direct load is linear by lane, so the XOR swizzle of the image is applied to the SOURCE chunk a
// lane fetches (as conv3d_bf16_v3.hip does).  Addressing as in v3: lane part (column, chunk, or the out-of-range mark) in the
// VGPR offset, wave-uniform part (slice, row) in the SGPR offset, a wave-uniform "not there" selects a zero-sized descriptor
// (the load then writes zeros).  The transposed reads, the tap -> wave deal (7,7,7,6 + ones-tap = dbias; 16 padded input
// channels: two taps per accumulator) and the slab + fixed-order reduce are those of conv3d_wgrad_bf16.hip.
#include "pv_common.h"

namespace pv {

constexpr int W2TR = 8, W2TRI = 10, W2TW = 32, W2XW = 34;
constexpr int W2_SLAB_ELEMS = 28 * 32 * 32;
constexpr uint32_t W2_INVALID = 0x40000000u;

typedef __attribute__((address_space(3))) s16x4 w2_lds_s16x4;
typedef __attribute__((address_space(3))) unsigned char* w2_lds_ptr;
typedef int w2_i32x4 __attribute__((ext_vector_type(4)));

// One LDS-direct load (16 bytes per lane, LDS destination = lds_addr + 16 * lane) as inline assembly.  Through the builtin
// (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc orders the LDS write against the transposed LDS reads of the k-loop, whose
// intrinsic carries no memory operand: it put `s_waitcnt vmcnt(0)` behind every staging instruction, i.e. a whole memory
// latency into every k-step (the first build of this kernel ran at half the speed of the register-staged one).  As an asm
// statement the load is invisible to that pass; its completion is waited for by the explicit vmcnt(0) + barrier at the top
// of the next slice, and this loop issues no other vector-memory instruction the compiler would have to count.
__device__ __forceinline__ void w2_lds_dma16(uint32_t lds_addr, uint32_t voff, w2_i32x4 rsrc, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc),
               "s"(soff)
               : "memory");
}
__device__ __forceinline__ w2_i32x4 w2_rsrc(const void* base, int bytes) {   // raw buffer descriptor, `bytes` records
  const uintptr_t a = (uintptr_t)base;
  return (w2_i32x4){(int)(uint32_t)a, (int)((a >> 32) & 0xffffu), bytes, 0x00020000};
}

#ifdef PV_DIAG_STAMPS
__device__ unsigned long long wgrad2_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif

template <int NACC, int Q>
__device__ __forceinline__ void w2_interleave() {
  if constexpr (Q < NACC) {
    constexpr int R = 2 + 2 * NACC;
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, R / NACC + (Q < R % NACC ? 1 : 0), 0);
    w2_interleave<NACC, Q + 1>();
  }
}

// F16: the operand images hold IEEE half floats (the two-term split of the f32 path, pv_pack_split2_...): the same data
// movement, v_mfma_f32_32x32x16_f16 instead of ..._bf16 and 1.0 as a half in the ones-tap
// PACK12 (CPAD == 16, at most 12 real input channels: the first layer's 11 satellite channels): the N axis of the product is
// (tap, 4-channel piece) with THREE pieces per tap instead of four -- 27 x 12 + the ones-tap = 82 pieces = 11 column tiles of
// 32 where the paired form multiplies 14 (two taps x 16 padded channels each): three accumulators per wave instead of four.
// A transposed read lets every lane address its own 4-channel piece, so a column tile may mix taps freely: piece P = 8 T +
// 4 (lane group & 1) + pi of tile T belongs to tap P / 3, channels 4 (P % 3) .. + 3; P = 81 is the ones-tap, larger ones read
// zeros.
template <int CPAD, bool F16 = false, bool PACK12 = false>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_bf16_v2_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, float* __restrict__ slabs, int t_in, int h_in, int w_in,
    int t_out, int h_out, int w_out, int pad_t, int pad_h, int pad_w, int n_colblk, int t_chunk, int rpb) {
  // rpb: output rows per workgroup, 1 .. 8 (W2TR).  56 output rows at B = 32 are 7 blocks of 8 = 224 workgroups on 256 CUs; as
  // 8 blocks of 7 every CU has one and each walks 14 of the 16 k-steps of a slice (the launcher decides, wgrad_v2_grid).
  constexpr int NCH = CPAD / 8;       // 16-byte chunks per X voxel
  constexpr int VPR = 16 / NCH;
  constexpr int VOXB = CPAD * 2;
  constexpr int ROWB = W2XW * VOXB;   // 2176 / 1088
  constexpr int SLOTB = W2TRI * ROWB;
  constexpr int XVPP = 1024 / VOXB;   // X voxels per 1-KB piece: 16 / 32
  constexpr int XPIECES = (W2XW + XVPP - 1) / XVPP;            // pieces per X row: 3 / 2
  constexpr int XTAIL_LANES = (W2XW - (XPIECES - 1) * XVPP) * NCH;   // lanes of the last piece: 8 / 4
  constexpr int DROWB = W2TW * 64;    // dY tile: always 32 channels
  constexpr int DSLOTB = W2TR * DROWB;
  // [X ring: 5 slots][128 B of bf16 ones + 128 B of zeros: the B operand of the ones-tap (dbias) and of the empty tap
  // slots, fetched like any other fragment][dY ring: 2 slots]
  constexpr int NXS = 5;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NXS * SLOTB + 256 + 2 * DSLOTB];
  unsigned char* lds_const = lds + NXS * SLOTB;
  unsigned char* lds_dy = lds_const + 256;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;   // waves 4..7 stage, waves 0..3 multiply
  const int wave = wave8 & 3;
  // transposed-read roles of this lane
  const int grp = lane >> 4;        // 16-lane group
  const int qi = (lane & 15) >> 2;  // block row (voxel) this lane addresses
  const int pi = lane & 3;          // 4-channel piece this lane addresses
  const int hh = grp >> 1;          // k half (voxels 8*hh ..)
  const int cb = 16 * (grp & 1);    // channel base of the group

  const int h0 = blockIdx.x * rpb;
  const int b = blockIdx.z;
  const int tc0 = blockIdx.y * t_chunk;
  const int tc1 = min(tc0 + t_chunk, t_out);
  const int wg_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* slab = slabs + (size_t)wg_id * W2_SLAB_ELEMS;

  static_assert(!PACK12 || CPAD == 16, "PACK12 packs the 16-channel image");
  constexpr bool PAIRED = CPAD == 16;      // two taps per accumulator (PACK12: eight 4-channel pieces of consecutive taps)
  constexpr int NACC = PACK12 ? 3 : (PAIRED ? 4 : 7);
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  // Item = (column tile cb, output slice t): items run cb-major.  X slices of tile cb are numbered u = cb * tn + (s - tc0),
  // s = tc0 .. tc1 + 1 (tn = nt + 2 of them) and live in ring slot u % 5; dY tiles v = cb * nt + (t - tc0) in slot v & 1.
  // Item (cb, t) multiplies dY v with X u0, u0 + 1, u0 + 2, u0 = cb * tn + (t - tc0).
  const int nt = tc1 - tc0, tn = nt + 2;
  const int n_items = nt > 0 ? nt * n_colblk : 0;
  const int u_end = tn * n_colblk;   // X slices in all
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(w2_lds_ptr)lds);

  if (n_items > 0 && loader) {
    // =============================== staging waves =====================================================================
    const uint32_t x_plane_b = (uint32_t)h_in * w_in * CPAD * 2u, x_row_b = (uint32_t)w_in * CPAD * 2u;
    const uint32_t d_plane_b = (uint32_t)h_out * w_out * 64u, d_row_b = (uint32_t)w_out * 64u;
    const size_t x_sample = (size_t)t_in * h_in * w_in * CPAD, d_sample = (size_t)t_out * h_out * w_out * 32;
    const void* const x_base = x + (size_t)b * x_sample;
    const void* const d_base = dy + (size_t)b * d_sample;
    const int x_bytes = (int)(x_sample * 2), d_bytes = (int)(d_sample * 2);
    // rows of this wave: X rows wave, wave + 4, wave + 8 (< 10); dY rows wave, wave + 4
    uint32_t xrow_src[3];
    bool xrow_in[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int hi = h0 - pad_h + wave + 4 * i;
      xrow_in[i] = (unsigned)hi < (unsigned)h_in && wave + 4 * i < rpb + 2;
      xrow_src[i] = xrow_in[i] ? (uint32_t)hi * x_row_b : 0u;
    }
    uint32_t drow_src[2];
    bool drow_in[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ho = h0 + wave + 4 * i;
      drow_in[i] = ho < h_out && wave + 4 * i < rpb;
      drow_src[i] = drow_in[i] ? (uint32_t)ho * d_row_b : 0u;
    }
    const bool has_third_row = wave + 8 < W2TRI;
    // X piece (row, pc): lane -> voxel pc * XVPP + lane / NCH at position lane % NCH, which holds source chunk
    // position ^ swz(voxel) (the swizzle has period 16 voxels, so every piece of a row shares the lane pattern)
    auto x_slice = [&](int u) {   // X slice number u of the whole walk
      const int cb = u / tn, s = tc0 + (u - cb * tn), w0 = cb * W2TW;
      const int ti = s - pad_t;
      const bool t_ok = (unsigned)ti < (unsigned)t_in;
      const uint32_t slot = lds_base + (uint32_t)((u % NXS) * SLOTB);
#pragma unroll
      for (int pc = 0; pc < XPIECES; ++pc) {
        const int v = pc * XVPP + lane / NCH;
        const int src = (lane % NCH) ^ ((v / VPR) % NCH);
        const int wi = w0 - pad_w + v;
        const uint32_t lane_off = ((unsigned)wi < (unsigned)w_in && v < W2XW) ? (uint32_t)(wi * CPAD + src * 8) * 2u : W2_INVALID;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i == 2 && !has_third_row) continue;
          const bool ok = t_ok && xrow_in[i];
          if (pc + 1 < XPIECES || lane < XTAIL_LANES)
            w2_lds_dma16(slot + (uint32_t)((wave + 4 * i) * ROWB + pc * 1024), lane_off, w2_rsrc(x_base, ok ? x_bytes : 0),
                         ok ? (uint32_t)ti * x_plane_b + xrow_src[i] : 0u);
        }
      }
    };
    // dY piece (row, pc): voxel pc * 16 + lane / 4, position lane % 4 holds source chunk position ^ ((voxel >> 2) & 3)
    auto d_slice = [&](int v_seq) {
      const int cb = v_seq / nt, t = tc0 + (v_seq - cb * nt), w0 = cb * W2TW;
      const uint32_t slot = lds_base + (uint32_t)(NXS * SLOTB + 256 + (v_seq & 1) * DSLOTB);
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int v = pc * 16 + (lane >> 2);
        const int src = (lane & 3) ^ ((v >> 2) & 3);
        const uint32_t lane_off = (w0 + v) < w_out ? (uint32_t)((w0 + v) * 32 + src * 8) * 2u : W2_INVALID;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool ok = drow_in[i];
          w2_lds_dma16(slot + (uint32_t)((wave + 4 * i) * DROWB + pc * 1024), lane_off, w2_rsrc(d_base, ok ? d_bytes : 0),
                       ok ? (uint32_t)t * d_plane_b + drow_src[i] : 0u);
        }
      }
    };

    // prologue: X slices 0 .. 3 and dY tile 0; then, inside item j (X u0 .. u0 + 2 being read), everything up to X u0 + 4
    // (its slot held u0 - 1, which the previous item was the last to read) and dY tile j + 1 (slot of tile j - 1)
    int next_u = 0;
    for (; next_u < 4 && next_u < u_end; ++next_u) x_slice(next_u);
    d_slice(0);
    for (int j = 0; j < n_items; ++j) {
      const int cb = j / nt, u0 = cb * tn + (j - cb * nt);
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): everything this wave requested has landed
      __syncthreads();                      // item j may start; item j - 1 is done with its operands
      if (next_u < u0 + 3) {                // first item of a further tile: X u0 + 2 could not be fetched earlier
        for (; next_u < u0 + 3; ++next_u) x_slice(next_u);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
      }
      for (; next_u <= u0 + 4 && next_u < u_end; ++next_u) x_slice(next_u);
      if (j + 1 < n_items) d_slice(j + 1);
    }
  } else if (n_items > 0) {
    // =============================== multiplying waves =================================================================
    if (tid < 64) reinterpret_cast<uint32_t*>(lds_const)[tid] = tid < 32 ? (F16 ? 0x3c003c00u : 0x3f803f80u) : 0u;

    // ---- per-lane byte offsets of the transposed reads (identical to conv3d_wgrad_bf16.hip) ----------------------
    int aoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = 8 * hh + 4 * s + qi;
      const int chunk = (cb + 4 * pi) >> 3;
      aoff[s] = v * 64 + ((chunk ^ ((v >> 2) & 3)) << 4) + (pi & 1) * 8;
    }
    int boff[NACC][2];
    int bkt[NACC];
    int my_tap[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      int tap, c4 = 0;
      if constexpr (PACK12) {
        const int piece = 8 * (wave + 4 * i) + 4 * (grp & 1) + pi;      // 0 .. 95; 81 = the ones-tap, beyond it nothing
        tap = piece / 3;
        c4 = piece - 3 * tap;
        if (piece > 81) tap = 28;
      } else {
        tap = PAIRED ? 2 * (wave + 4 * i) + (grp & 1) : wave + 4 * i;
      }
      my_tap[i] = tap;
      if (tap > 26) tap = 26;
      const int kt = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      bkt[i] = kt;
      const int cbx = PAIRED ? 0 : cb;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int v = kw + 8 * hh + 4 * s + qi;
        const int chunk = PACK12 ? (c4 >> 1) : ((cbx + 4 * pi) >> 3);
        const int sub = PACK12 ? (c4 & 1) : (pi & 1);
        boff[i][s] = kh * ROWB + v * VOXB + ((chunk ^ ((v / VPR) % NCH)) << 4) + sub * 8;
      }
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int last_tap = (PAIRED || PACK12) ? my_tap[NACC - 1] : (wave == 3 ? 27 : 0);
    const unsigned char* const_frag = last_tap < 27 ? nullptr : lds_const + (last_tap == 27 ? 0 : 128) + (lane & 15) * 8;

#ifdef PV_DIAG_STAMPS
    unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, s0, s1, s2;
#endif
    for (int j = 0; j < n_items; ++j) {
      const int cbj = j / nt, u0 = cbj * tn + (j - cbj * nt);
      PV_STAMP(s0);
      __syncthreads();                       // the loaders' data for item j is in place
      if (cbj > 0 && j == cbj * nt) __syncthreads();   // (tile switch: the loaders fetch X u0 + 2 behind the first barrier)
      PV_STAMP(s1);
      int slot_of_kt[3];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) slot_of_kt[kt] = ((u0 + kt) % NXS) * SLOTB;
      const unsigned char* dy_slot = lds_dy + (j & 1) * DSLOTB;
      int bslot[NACC];
#pragma unroll
      for (int i = 0; i < NACC; ++i) bslot[i] = bkt[i] == 0 ? slot_of_kt[0] : (bkt[i] == 1 ? slot_of_kt[1] : slot_of_kt[2]);
      // 16 k-steps (8 rows x 2 column groups of 16 voxels); each = 2 + 2 NACC transposed reads + NACC MFMAs, software-
      // pipelined by one k-step: the reads of k-step s+1 are dealt between the MFMAs of k-step s
      auto read_step = [&](int st, s16x4 (&ra)[2], s16x4 (&rb)[NACC][2]) {
        const int rho = st >> 1, cg = st & 1;
        const unsigned char* ap = dy_slot + rho * DROWB + cg * 16 * 64;
        ra[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4*)(ap + aoff[0]));
        ra[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4*)(ap + aoff[1]));
        const int xo = rho * ROWB + cg * 16 * VOXB;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          const unsigned char* bp = lds + bslot[i] + xo;
          const unsigned char* p0 = bp + boff[i][0];
          const unsigned char* p1 = bp + boff[i][1];
          if (i == NACC - 1) {  // the last slot may hold the ones-tap (27) or run past it: the select is on the ADDRESS
            p0 = const_frag ? const_frag : p0;
            p1 = const_frag ? const_frag : p1;
          }
          rb[i][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4*)p0);
          rb[i][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_lds_s16x4*)p1);
        }
      };
      auto mfma_step = [&](const s16x4 (&ra)[2], const s16x4 (&rb)[NACC][2]) {
        const s16x8 a8 = {ra[0][0], ra[0][1], ra[0][2], ra[0][3], ra[1][0], ra[1][1], ra[1][2], ra[1][3]};
        typedef _Float16 w2_f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          s16x8 t8 = {rb[i][0][0], rb[i][0][1], rb[i][0][2], rb[i][0][3], rb[i][1][0], rb[i][1][1], rb[i][1][2], rb[i][1][3]};
          if constexpr (F16)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w2_f16x8, a8), __builtin_bit_cast(w2_f16x8, t8), acc[i], 0, 0, 0);
          else
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, t8), acc[i], 0, 0, 0);
        }
      };
      s16x4 ra0[2], ra1[2], rb0[NACC][2], rb1[NACC][2];
      // rows >= rpb of the tile belong to the next workgroup: the loaders leave them zero, so their k-steps add nothing; the
      // LAST row's two k-steps are skipped outright when the workgroup has fewer than 8 rows (one wave-uniform branch around
      // the last unrolled iteration -- a run-time trip count in this software-pipelined loop cost 819 spills)
      const bool skip_last_row = rpb < W2TR;
      read_step(0, ra0, rb0);
#pragma unroll
      for (int st = 0; st < W2TR * 2; st += 2) {
        if (st == W2TR * 2 - 2 && skip_last_row) break;
        __builtin_amdgcn_sched_barrier(0);
        read_step(st + 1, ra1, rb1);
        mfma_step(ra0, rb0);
        w2_interleave<NACC, 0>();
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < W2TR * 2) {
          read_step(st + 2, ra0, rb0);
          mfma_step(ra1, rb1);
          w2_interleave<NACC, 0>();
        } else {
          mfma_step(ra1, rb1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef PV_DIAG_STAMPS
      PV_STAMP(s2);
      dg[0] += s1 - s0;   // barrier(s): waiting for the loaders / the other multiplying waves
      dg[1] += s2 - s1;   // 16 k-steps
      dg[2] += 1;
#endif
    }
#ifdef PV_DIAG_STAMPS
    if (lane == 0 && wg_id * 4 + wave < PV_DIAG_WAVES)
      for (int i = 0; i < PV_DIAG_SLOTS; ++i) wgrad2_diag[(size_t)(wg_id * 4 + wave) * PV_DIAG_SLOTS + i] = dg[i];
#endif
  }
  if (loader) return;   // (no barrier follows)

  // ---- write the partial slab: [tapslot][co][ci], C layout: col = ci = lane&31, row = co ----------
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    int tapslot = PAIRED ? 2 * (wave + 4 * i) + ((lane & 31) >> 4) : wave + 4 * i;  // 27 = ones-tap
    int ci = PAIRED ? (lane & 15) : (lane & 31);
    if constexpr (PACK12) {      // column (lane & 31) of tile wave + 4 i = piece 8 T + column / 4, channel 4 (piece % 3) + column % 4
      const int piece = 8 * (wave + 4 * i) + ((lane & 31) >> 2);
      tapslot = piece / 3;
      ci = 4 * (piece - 3 * tapslot) + (lane & 3);
      if (piece > 81) tapslot = 28;
    }
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

// one workgroup (8 waves, 142 / 87 KB of LDS) per CU; a workgroup owns 8 output rows of a sample over ALL columns.  Cut the
// time march only when there are fewer row blocks than CUs: rounds x (slices per chunk + the two extra X slices per tile)
void wgrad_v2_grid(const pv_conv3d_dims* d, int* n_rowblk, int* n_colblk, int* n_tchunk, int* t_chunk, int* rows_per_blk) {
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  *n_rowblk = (ho + W2TR - 1) / W2TR;
  // rows per workgroup: as few as cover the image with that many blocks (a 9-row image is two blocks of 5: 10 of the 16 k-steps
  // of a slice); and one more block where that shortens every workgroup and fills CUs the 8-row split leaves idle (56 rows at
  // B = 32: 7 blocks of 8 on 224 CUs -> 8 blocks of 7 on 256)
  if ((long long)d->batch * *n_rowblk < 256 && (long long)d->batch * (*n_rowblk + 1) <= 256 &&
      (ho + *n_rowblk) / (*n_rowblk + 1) < (ho + *n_rowblk - 1) / *n_rowblk)
    *n_rowblk += 1;
  *rows_per_blk = (ho + *n_rowblk - 1) / *n_rowblk;
  *n_colblk = (wo + W2TW - 1) / W2TW;
  const long long tiles = (long long)d->batch * *n_rowblk;
  int max_chunks = (to + 1) / 2;
  if (max_chunks < 1) max_chunks = 1;
  int ntc = 1;
  long long best = -1;
  for (int c = 1; c <= max_chunks && tiles * c <= 16 * 256; ++c) {
    const int tch = (to + c - 1) / c, nch = (to + tch - 1) / tch;
    const long long rounds = (tiles * nch + 255) / 256;
    const long long cost = rounds * (tch + 2);
    if (best < 0 || cost < best) best = cost, ntc = nch;
  }
  *t_chunk = (to + ntc - 1) / ntc;
  *n_tchunk = (to + *t_chunk - 1) / *t_chunk;
}

size_t wgrad_v2_workspace_bytes(const pv_conv3d_dims* d) {
  int nrb, ncb, ntc, tch, rpb;
  wgrad_v2_grid(d, &nrb, &ncb, &ntc, &tch, &rpb);
  return (size_t)d->batch * nrb * ntc * W2_SLAB_ELEMS * sizeof(float);
}

// 0 = launched (n_slabs set), 1 = shape / alignment not covered (the caller uses the register-staged kernel)
int launch_conv3d_wgrad_bf16_v2(const uint16_t* x, const uint16_t* dy, float* slabs, const pv_conv3d_dims* d, int to, int ho,
                                int wo, hipStream_t st, int* n_slabs, bool f16) {
  if (((uintptr_t)x % 16) != 0 || ((uintptr_t)dy % 16) != 0) return 1;
  int nrb, ncb, ntc, tch, rpb;
  wgrad_v2_grid(d, &nrb, &ncb, &ntc, &tch, &rpb);
  *n_slabs = d->batch * nrb * ntc;
  dim3 grid((unsigned)nrb, (unsigned)ntc, (unsigned)d->batch);
#define PV_W2(CP, HALF)                                                                                                       \
  hipLaunchKernelGGL((conv3d_wgrad_bf16_v2_kernel<CP, HALF>), grid, dim3(512), 0, st, x, dy, slabs, d->t_in, d->h_in, d->w_in, to, \
                     ho, wo, d->pad_t, d->pad_h, d->pad_w, ncb, tch, rpb)
#define PV_W2P(HALF)                                                                                                        \
  hipLaunchKernelGGL((conv3d_wgrad_bf16_v2_kernel<16, HALF, true>), grid, dim3(512), 0, st, x, dy, slabs, d->t_in, d->h_in, d->w_in, \
                     to, ho, wo, d->pad_t, d->pad_h, d->pad_w, ncb, tch, rpb)
  if (pv_bf16_cpad(d->c_in) == 16) {
    // (PV_WGRAD_NO_PACK12: the paired form for channel counts it also covers -- the cross-check of tests/test_gpu_conv.py)
    const bool pack12 = d->c_in <= 12 && !getenv("PV_WGRAD_NO_PACK12");
    if (pack12) { if (f16) PV_W2P(true); else PV_W2P(false); }
    else { if (f16) PV_W2(16, true); else PV_W2(16, false); }
  } else { if (f16) PV_W2(32, true); else PV_W2(32, false); }
#undef PV_W2P
#undef PV_W2
  return 0;
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
  const int i = blockIdx.x * 64 + lane;  // index into [28][32][32]; W2_SLAB_ELEMS % 64 == 0
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int k = grp; k < n_slabs; k += 16 * WR_GROUPS) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (k + WR_GROUPS * j < n_slabs) ? slabs[(size_t)(k + WR_GROUPS * j) * W2_SLAB_ELEMS + i] : 0.f;
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

// dY gated by the ReLU derivative of the layer's own output, as a copy: dst = dy where y > 0 (as bf16: sign clear and not +0),
// else 0.  The loader-wave kernel above stages dY by LDS-direct loads and cannot gate on the way; callers that hand over a
// gate (layers whose consumer did not pre-gate the gradient: the sat+NWP towers' last layers, eval-time probes) pay this one
// streaming pass instead of the register-staged kernel of rounds 1-2 (removed in round 6: 380 lines that only those calls reached).
__global__ __launch_bounds__(256) void wgrad_gate_copy_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ y,
                                                              u32x4* __restrict__ dst, size_t n8) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const u32x4 g = dy[i], a = y[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] & (relu_pair01(a[j]) * 0xffffu);      // (bit 0 / bit 16 -> 0xffff in that half)
    dst[i] = o;
  }
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_conv3d_bwd_weight_bf16_workspace_bytes(const pv_conv3d_dims* d, size_t* bytes) {
  PV_REQUIRE(d && bytes, PV_EINVAL, "pv_conv3d_bwd_weight_bf16_workspace_bytes: null pointer");
  PV_REQUIRE(d->batch > 0 && d->t_in > 0 && d->h_in > 0 && d->w_in > 0, PV_EINVAL,
             "pv_conv3d_bwd_weight_bf16_workspace_bytes: bad dims");
  // the partial slabs, and behind them room for a gated copy of dY (used only when a gate is handed over)
  const long long to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  const size_t dy_bytes = (to > 0 && ho > 0 && wo > 0) ? (size_t)d->batch * to * ho * wo * pv_bf16_cpad(d->c_out) * 2 : 0;
  *bytes = ((wgrad_v2_workspace_bytes(d) + 255) & ~(size_t)255) + dy_bytes;
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
  PV_REQUIRE(d->batch <= 65535, PV_ESIZE, "pv_conv3d_bwd_weight_bf16: batch too large for grid.z");
  PV_REQUIRE((size_t)d->t_in * d->h_in * d->w_in * 64 <= 0x40000000ull, PV_ESIZE,
             "pv_conv3d_bwd_weight_bf16: one sample exceeds 1 GiB (buffer-addressing limit of this kernel)");
  const size_t slab_bytes = (wgrad_v2_workspace_bytes(d) + 255) & ~(size_t)255;
  PV_REQUIRE(workspace_bytes >= wgrad_v2_workspace_bytes(d), PV_ESIZE, "pv_conv3d_bwd_weight_bf16: workspace too small");
  PV_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)workspace % 16) == 0, PV_EINVAL,
             "pv_conv3d_bwd_weight_bf16: x, dy and the workspace must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  if (y_relu_mask) {
    const size_t n = (size_t)d->batch * to * ho * wo * pv_bf16_cpad(d->c_out);      // bf16 elements of dY (a multiple of 16)
    PV_REQUIRE(workspace_bytes >= slab_bytes + n * 2, PV_ESIZE,
               "pv_conv3d_bwd_weight_bf16: a gate needs room for the gated copy of dY behind the slabs (workspace_bytes query)");
    PV_REQUIRE(((uintptr_t)y_relu_mask % 16) == 0, PV_EINVAL, "pv_conv3d_bwd_weight_bf16: the gate must be 16-byte aligned");
    uint16_t* gated = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + slab_bytes);
    hipLaunchKernelGGL(wgrad_gate_copy_kernel, dim3(stream_grid(n / 8, 256)), dim3(256), 0, st,
                       reinterpret_cast<const u32x4*>(dy), reinterpret_cast<const u32x4*>(y_relu_mask),
                       reinterpret_cast<u32x4*>(gated), n / 8);
    dy = gated;
  }
  int n2 = 0;
  PV_REQUIRE(launch_conv3d_wgrad_bf16_v2(x, dy, (float*)workspace, d, to, ho, wo, st, &n2, false) == 0, PV_EINVAL,
             "pv_conv3d_bwd_weight_bf16: x and dy must be 16-byte aligned");
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(W2_SLAB_ELEMS / 64), dim3(64 * WR_GROUPS), 0, st, (const float*)workspace, n2,
                     dw, dbias, d->c_out, d->c_in);
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
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(W2_SLAB_ELEMS / 64), dim3(64 * WR_GROUPS), 0, st, (const float*)workspace, n2,
                     dw, dbias, d->c_out, d->c_in);
  return check_launch("pv_conv3d_bwd_weight_f16");
}

}  // extern "C"

#ifdef PV_DIAG_STAMPS
extern "C" int pv_diag_read_wgrad2(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::wgrad2_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
