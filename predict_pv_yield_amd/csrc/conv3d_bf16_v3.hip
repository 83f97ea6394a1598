// bf16 MFMA Conv3D for 32 -> 32 channel layers, INPUT-STATIONARY time march (forward L1..L3 and every dgrad).
//
// Its predecessor (two waves per SIMD, 16 couts per wave, one output slice at a time from the three input slices it needs;
// removed in round 3) read every B fragment (activations, 32 cin x 16 voxels) for only the 3 kh taps of one kt plane, and
// the two waves of a cout pair read the same fragments -- 864 ds_read_b128 per slice and CU = 6912 LDS cycles, exactly the
// 6912 matrix-pipe cycles per SIMD: LDS bandwidth co-limited it (measured matrix pipe busy 60 %).
//
// Here the march is turned around: each step takes ONE INPUT slice s and adds its contribution to the three output
// slices s, s-1, s-2 (tap planes kt = 0, 1, 2) that are in flight in registers.  A fragment is read once and feeds up
// to 3 kh x 3 kt MFMAs: a third of the LDS operand traffic.  To make room for 3 x 32 accumulator registers inside the
// 256-register budget of two waves per SIMD:
//   * the kt = 2 tap plane of the weights lives in LDS (3 fragment reads per (slice, kw)); the NDHWC-writing variants keep
//     kt = 1 there as well (W1_LDS: 9 instead of 18 weight fragments in registers, which is what lets the epilogue's pending
//     stores and gate loads stay in registers without spills), the NCDHW-writing variant keeps kt = 0, 1 in registers;
//   * input slices are staged global -> LDS directly (buffer_load_dwordx4 ... lds, gfx950), no staging registers: the
//     LDS destination of a wave instruction is 1 KB linear by lane, so the XOR swizzle of the image is applied to the
//     SOURCE address of each lane instead;
//   * the finished tile goes from registers straight to global memory (8 B per lane = 4 consecutive couts of a voxel;
//     the two waves of a cout pair fill the two 32-byte halves of every voxel) -- no LDS staging, no second barrier.
// One barrier per slice, and because a step only needs the slice being read plus the one being fetched, the ring is
// TWO slots deep.  That shrinks the LDS footprint enough to run two independent 4-wave workgroups per CU (tile 8 rows x
// 32 columns each, one wave of each workgroup per SIMD): the two workgroups drift apart, so while one sits at its
// barrier or converts / stores a finished tile the other keeps the matrix pipe busy -- without the lock-step a single
// 8-wave workgroup imposes.  Weight fragments: pack_weight_v3_kernel below.
#include "pv_common.h"

namespace pv {

constexpr int V3_TR = 8, V3_TRI = 10, V3_TW = 34, V3_TW_VALID = 32;  // input row 34 voxels -> 32 output columns
constexpr int V3_VOXB = 64, V3_ROWB = V3_TW * V3_VOXB, V3_SLOTB = V3_TRI * V3_ROWB;
constexpr int V3_W2B = 9 * 2 * 64 * 16;  // tap plane kt = 2: [9 taps][2 cout halves][64 lanes] x 16 B
constexpr uint32_t V3_INVALID = 0x40000000u;
constexpr int V3_PATCH_CS = 4 * 64 + 16;        // NCDHW epilogue patch: bytes per cout ([4 rows][32 voxels] bf16 + pad)
constexpr int V3_PATCHB = 16 * V3_PATCH_CS;     // per wave: 16 couts

#ifdef PV_DIAG_STAMPS
__device__ unsigned long long v3_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#define PV_V3_DIAG(i, a, b) dg[i] += (b) - (a)
#else
#define PV_V3_DIAG(i, a, b) do { } while (0)
#endif

// Packed bf16-pair helpers of the epilogue (one instruction per pair; hipcc turns the equivalent builtins into two compares,
// two selects and a byte permute).  As 16-bit integers a bf16 is > 0 exactly when the signed value is.
__device__ __forceinline__ uint32_t v3_pk_max(uint32_t x, uint32_t floor2) {   // signed 16-bit max on both halves
  uint32_t r;   // floor2 = 0: ReLU (negative floats -> +0); floor2 = 0x80008000 (most negative): no-op
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "s"(floor2));
  return r;
}
__device__ __forceinline__ uint32_t v3_pk_gate(uint32_t x, uint32_t g) {   // keep each half of x where that half of g is > 0
  uint32_t m, r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(m) : "v"(g));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "s"(0x00010001u));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
}

// the same from two bits of a mask byte: halves of x stay where bits `lo`, `lo + 1` of b are set
__device__ __forceinline__ uint32_t v3_pk_gate_bits(uint32_t x, uint32_t b, int lo) {
  uint32_t m = __builtin_amdgcn_ubfe(b, lo, 2);      // (b1 b0)
  m = (m * 0x8001u) & 0x00010001u;                    // b0 -> bit 0, b1 -> bit 16 (a 24-bit multiply)
  uint32_t r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
}
// One input slice's contribution to output slices s - kt, kt in [KT_LO, KT_HI] (compile-time: head and tail steps of
// the march feed fewer slices).  P = (s - tc0) % 3 names the accumulator slot of output slice s.
// FIRST (KW == 0 call of a step with KT_LO == 0): the tap (kt 0, kh 0, kw 0) is the first contribution output slice s ever
// receives, so its MFMA takes the bias vector as the C operand instead of the accumulator -- the 32 accumulator registers
// of a finished tile need no re-initialisation moves.
// side(j), j = 6 KW + ir: one vector-memory instruction of the step's staging / write-out work, issued between two rows of
// MFMAs (see PV_V3_STEP): a burst of them right behind the barrier kept the wave at the issue of its 10..17 memory
// instructions for a fifth of the step (in-kernel stamps), the queue of the texture-address unit being a few entries deep.
// W1_LDS: the kt = 1 tap plane of the weights is read from LDS like kt = 2 (w2 - V3_W2B), only kt = 0 stays in registers.
// F16: the operands are half floats (the f32-accurate two-term form, conv3d_f16x2.hip): same fragment layouts, the f16 instruction.
typedef _Float16 v3_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 v3_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v3_f16x8, a), __builtin_bit_cast(v3_f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int P, int KT_LO, int KT_HI, int KW, bool W1_LDS, bool F16, typename Side>
__device__ __forceinline__ void v3_accumulate(const unsigned char* slot, const int (&voff)[3],
                                              const bf16x8 (&wfrag)[W1_LDS ? 9 : 18], const unsigned char* w2,
                                              f32x4 (&acc)[3][4][2], const f32x4& b4, Side&& side) {
  {
    constexpr int kw = KW;
    bf16x8 cur[2], nxt[2], wk2[3], wk1[W1_LDS ? 3 : 1];
    if (KT_HI == 2) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) wk2[kh] = *reinterpret_cast<const bf16x8*>(w2 + (kh * 3 + kw) * 2048);
    }
    if (W1_LDS && KT_LO <= 1 && KT_HI >= 1) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) wk1[kh] = *reinterpret_cast<const bf16x8*>(w2 - V3_W2B + (kh * 3 + kw) * 2048);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) cur[half] = *reinterpret_cast<const bf16x8*>(slot + voff[kw] + half * (16 * V3_VOXB));
#pragma unroll
    for (int ir = 0; ir < 6; ++ir) {
      if (ir < 5) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
          nxt[half] = *reinterpret_cast<const bf16x8*>(slot + (ir + 1) * V3_ROWB + voff[kw] + half * (16 * V3_VOXB));
      }
      side(6 * KW + ir);
#pragma unroll
      for (int kt = KT_LO; kt <= KT_HI; ++kt) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int orow = ir - kh;
          if (orow >= 0 && orow < 4) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
              acc[(P - kt + 3) % 3][orow][half] = v3_mfma<F16>(
                  kt == 2 ? wk2[kh] : ((W1_LDS && kt == 1) ? wk1[kh] : wfrag[kt * 9 + kh * 3 + kw]), cur[half],
                  (kt == 0 && kh == 0 && KW == 0) ? b4 : acc[(P - kt + 3) % 3][orow][half]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int half = 0; half < 2; ++half) cur[half] = nxt[half];
    }
  }
}

// Y_NCDHW (the last conv layer, whose output fc1 consumes in the reference's flatten order): the finished tile is
// transposed through a wave-private LDS patch ([cout][row][voxel]) and leaves as 16-byte pieces of a (cout, row) line
// (w_out % 8 == 0, checked by the launcher).
// OUT_GATE: the dgrad epilogue zeroes dx where the gating activation (the producer layer's ReLU output, bf16 NDHWC, same
// shape as y) is not > 0.
//
// Write-out and gate read (round 3).  An accumulator tile gives a lane 4 couts of one voxel (8 bytes); stored like that,
// the 64 lanes of an instruction touch 64 different cache lines, and in-kernel stamps (tools/diag_stamps.py) showed the
// waves parked at the ISSUE of these instructions for a third (forward) to a half (gated dgrad: the gate was fetched the
// same way) of a step -- the texture-address unit works through an instruction line by line.  Now the two column halves
// of a tile row swap registers between 16-lane rows (v_permlane16_swap: odd rows of one <-> even rows of the other), after
// which a lane holds 8 consecutive couts = 16 bytes of a voxel and an instruction covers 32 half-lines instead of 64
// eighth-lines: 4 dwordx4 stores (and 4 dwordx4 gate loads in the same lane geometry) per step instead of 8 + 8 dwordx2.
// The gate is applied to the packed bf16 pairs in that swapped geometry (3 packed integer ops per pair), ReLU likewise.
// F32OUT (round 5, the f32 model's forward / dgrad on the 16-bit cores): half-float operand images (one term of a two-term
// split each), the accumulators leave as they are -- f32 NDHWC, 16 bytes per lane and accumulator (4 couts of a voxel), no bias,
// no activation: one of the three partial products conv3d_f16x2.hip's sum pass adds up.  Eight stores per tile in the side slots
// the bf16 form uses for four stores + four gate loads (the same 32 registers).
// OUTM 2: half-float operands, HALF-FLOAT output = the accumulators times 2^-12 (V3_F16OUT_SCALE), through the bf16 form's packed
// epilogue: the two small partial products (x_l w_h, x_h w_l: 2^-11 of the result, so 11 more bits of them are all the sum can use)
// written and read back at half the bytes.  |x_l| <= 4, |w_h| < 2^14, 864 terms: below 2^26 before, 2^14 after the scale.
constexpr float V3_F16OUT_SCALE = 1.f / 4096.f;
typedef _Float16 v3_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t v3_pack_f16_pair(float a, float b) {      // one v_cvt_pk_f16_f32 (RNE)
  const pv_f32x2_t v = {a * V3_F16OUT_SCALE, b * V3_F16OUT_SCALE};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, v3_f16x2));
}

// MASKM (round 6): the 1-bit ReLU masks of the C ABI (u32 per voxel, bit c = channel c > 0, planes padded to whole 8 x 32 tiles:
// pv_relu_mask_dims).  1: the forward also WRITES the mask of its output -- a lane holds 8 consecutive couts of a voxel after the
// row swap, i.e. one byte of the voxel's word, stored in the side slots the gated form uses for its gate loads.  2: the gated
// dgrad reads that byte INSTEAD of the 16 bytes of the bf16 activation (out_gate is not touched): the conv family draws 0.5 of
// the HBM peak while it multiplies (bench.py roofline.all_conv_hbm_frac), and a gate read is a third of a dgrad launch's bytes.
template <bool OUT_GATE, bool Y_NCDHW, int OUTM = 0, int MASKM = 0>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_bf16_v3_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ wp2, const float* __restrict__ bias,
    uint16_t* __restrict__ y, const uint16_t* __restrict__ out_gate, int t_in, int h_in, int w_in, int t_out,
    int h_out, int w_out, int pad_t, int pad_h, int pad_w, int relu, int n_colblk, int t_chunk, int c_out,
    uint32_t* __restrict__ mask = nullptr, int mask_hp = 0, int mask_wp = 0) {
  // ring of 2 slices | [kt = 1 weight plane] | kt = 2 weight plane | 32 bias floats | [NCDHW patches]
  // NDHWC variants: 80 512 B, two workgroups per CU use 161 024 of the 163 840 B; the NCDHW variant keeps the kt = 1 plane
  // in registers (its patches take the room) and issues its memory work in bursts
  constexpr bool F32OUT = OUTM == 1, F16 = OUTM != 0;
  constexpr bool W1_LDS = !Y_NCDHW;
  constexpr int NWREG = W1_LDS ? 9 : 18;
  __shared__ __attribute__((aligned(1024))) unsigned char
      lds[2 * V3_SLOTB + (W1_LDS ? 2 : 1) * V3_W2B + 128 + (Y_NCDHW ? 4 * V3_PATCHB : 0)];
  unsigned char* lds_w2 = lds + 2 * V3_SLOTB + (W1_LDS ? V3_W2B : 0);
  float* lds_bias = reinterpret_cast<float*>(lds_w2 + V3_W2B);
  static_assert(!(OUT_GATE && Y_NCDHW), "the gated epilogue writes NDHWC");
  static_assert(!(F16 && (OUT_GATE || Y_NCDHW)), "the partial-product forms write plain NDHWC");
  static_assert(MASKM == 0 || (OUTM == 0 && !Y_NCDHW && (MASKM == 1) == !OUT_GATE), "mask forms: plain forward writes, gated dgrad reads");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = wave & 1;   // output-channel half (16 couts)
  const int wr = wave >> 1;  // row quad
  const int vox = lane & 15, kg = lane >> 4;

  const int rowblk = blockIdx.x / n_colblk;
  const int colblk = blockIdx.x - rowblk * n_colblk;
  const int h0 = rowblk * V3_TR;
  const int w0 = colblk * V3_TW_VALID;
  const int b = blockIdx.z;
  const int tc0 = blockIdx.y * t_chunk;
  const int tc1 = min(tc0 + t_chunk, t_out);
  if (tc1 - tc0 < 2) return;  // (the launcher never forms single-slice chunks: such launches take the v1 kernel)

  if (tid < 32) lds_bias[tid] = (bias && tid < c_out) ? bias[tid] : 0.f;

  // ---- staging: global -> LDS direct.  A wave instruction fills 16 voxels x 64 B = 1 KB (lane -> voxel lane/4, chunk
  // position lane%4); a row of 34 voxels = 2 full instructions + one with lanes 0..7.  Wave w stages rows w, w+4, w+8.
  // The lane fetches the source chunk that the swizzle maps to its position (the swizzle has period 8 voxels, so the
  // three segments of a row share it). -----------------------------------------------------------------------------
  const int svox = lane >> 2;
  const int ssrc = (lane & 3) ^ (((svox >> 2) & 1) << 1);
  uint32_t lane_voff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int swi = w0 - pad_w + 16 * j + svox;
    lane_voff[j] = ((unsigned)swi < (unsigned)w_in) ? (uint32_t)(swi * 32 + ssrc * 8) * 2u : V3_INVALID;
  }
  const uint32_t x_plane_b = (uint32_t)h_in * w_in * 64u, x_row_b = (uint32_t)w_in * 64u;
  const size_t sample_elems = (size_t)t_in * h_in * w_in * 32;
  // piece j = 3 i + p of slice s: row wave + 4 i of the slot, 16-voxel segment p (p = 2: the last two voxels).
  // Addressing keeps the loop free of vector address arithmetic and of scalar-register pressure (once the loop runs out of
  // SGPRs hipcc keeps wave-uniform values in vector registers and turns uniform tests into exec-masked branches): the lane
  // part of an address (column, chunk; or the out-of-range mark) is the instruction's VGPR offset and never changes; the
  // wave-uniform part (time slice + row) goes into the SGPR offset, which the hardware adds WITHOUT range checking it; a
  // wave-uniform "nothing there" (row outside the image, time padding, no such slice) selects a descriptor of ZERO
  // records, for which every lane is out of range: an LDS-direct load then writes zeros (what padding is), a store is
  // dropped.
  uint32_t row_src[3];
  bool row_in[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hi = h0 - pad_h + 4 * i + wave;
    row_in[i] = (unsigned)hi < (unsigned)h_in;
    row_src[i] = row_in[i] ? (uint32_t)hi * x_row_b : 0u;
  }
  const bool wave_has_third_row = wave + 8 < V3_TRI;
  const void* const x_sample = x + (size_t)b * sample_elems;
  const int x_sample_b = (int)(sample_elems * 2);
  auto load_piece = [&](int s, int j, bool live) {   // live: slice s exists (wave-uniform)
    const int i = j / 3, pc = j - 3 * i;
    if (i == 2 && !wave_has_third_row) return;
    const int ti = s - pad_t;
    const bool ok = live && row_in[i] && (unsigned)ti < (unsigned)t_in;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x_sample, 0, ok ? x_sample_b : 0, 0x00020000);
    unsigned char* dst = lds + (s & 1) * V3_SLOTB + (wave + 4 * i) * V3_ROWB + pc * 1024;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    if (pc < 2 || lane < 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, lane_voff[pc],
                                               ok ? (uint32_t)ti * x_plane_b + row_src[i] : 0u, 0, 0);
  };
  auto load_slice = [&](int s) {
#pragma unroll
    for (int j = 0; j < 9; ++j) load_piece(s, j, true);
  };

  load_slice(tc0);  // first: its HBM latency hides under the weight loads below

  // ---- weights: tap planes kt = 0, 1 resident in registers (18 A fragments, 16 couts x 32 cin), kt = 2 in LDS ----
  bf16x8 wfrag[NWREG];
#pragma unroll
  for (int tap = 0; tap < NWREG; ++tap)
    wfrag[tap] = *reinterpret_cast<const bf16x8*>(wp2 + ((size_t)(tap * 2 + ch) * 64 + lane) * 8);
  {
    unsigned char* wdst = lds_w2 - (W1_LDS ? V3_W2B : 0);
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(wp2 + (size_t)NWREG * 2 * 64 * 8);
    for (int i = tid; i < (W1_LDS ? 2 : 1) * V3_W2B / 16; i += 256) reinterpret_cast<u32x4*>(wdst)[i] = wsrc[i];
  }
  const unsigned char* w2 = lds_w2 + ch * 1024 + lane * 16;

  // ---- per-lane LDS read offsets of the B operand: voxel 16*half + vox + kw, 16-byte chunk kg ---------------------
  // (the column half 16 voxels further right has the same swizzle phase: + 1024 bytes, an immediate of the ds_read)
  int voff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int v = vox + kw;
    voff[kw] = v * V3_VOXB + ((kg ^ (((v >> 2) & 1) << 1)) << 4);
  }

  // ---- write-out geometry.  Accumulator [orow][half] gives lane (vox, kg) couts 16ch + 4kg .. +3 of voxel (row h0 + 4wr +
  // orow, column w0 + 16half + vox).  After the row swap of the two halves the lane holds couts 16ch + 8(kg>>1) .. +7 of
  // voxel (same row, column w0 + 16(kg&1) + vox): 16 bytes. --------------------------------------------------------------
  const int plane_out = h_out * w_out;
  uint32_t st_off;
  {
    const int col_t = 16 * (kg & 1) + vox;
    const bool ok = (w0 + col_t) < w_out;
    st_off = ok ? ((uint32_t)((h0 + 4 * wr) * w_out + w0 + col_t) * 32u + 16u * ch + 8u * (kg >> 1)) * 2u : V3_INVALID;
  }
  // F32OUT: accumulator [orow][half] as it is: couts 16ch + 4kg .. +3 (16 bytes) of voxel (row, column w0 + 16half + vox), 128 bytes per voxel
  constexpr uint32_t VOX_OUT_B = F32OUT ? 128u : 64u;
  uint32_t st_off_f[2] = {V3_INVALID, V3_INVALID};
  if constexpr (F32OUT) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col_t = 16 * half + vox;
      if ((w0 + col_t) < w_out) st_off_f[half] = (uint32_t)((h0 + 4 * wr) * w_out + w0 + col_t) * 128u + 64u * ch + 16u * kg;
    }
  }
  const size_t out_sample_b = (size_t)t_out * plane_out * VOX_OUT_B;
  // wave-uniform part of an output address: slice o, tile row orow (SGPR offset); rows below the image select the
  // zero-sized descriptor
  const uint32_t out_plane_b = (uint32_t)plane_out * VOX_OUT_B, out_row_b = (uint32_t)w_out * VOX_OUT_B;
  const int rows_left = h_out - (h0 + 4 * wr);   // tile rows orow < rows_left exist
  // a wave whose four tile rows all lie below the image (the second row quad of the last tile when h_out % 8 is 1..4: 60- and
  // 58-row planes) multiplies nothing -- its stores are dropped anyway; it still stages its share and keeps the barriers.  The
  // conv kernels are power-limited (profiles/r06/NOTES.md section 5): a matrix instruction not issued is clock for the others
  const bool quad_live = rows_left > 0;
  void* const y_sample = reinterpret_cast<unsigned char*>(y) + (size_t)b * out_sample_b;
  const void* const og_sample = (OUT_GATE ? out_gate : y) + (size_t)b * t_out * plane_out * 32;
  // finished tile in the store geometry (16 bytes per lane and tile row), stored one step later so the stores never sit in
  // front of a wait; its gate, fetched in the same geometry during the last two thirds of the step that finishes the tile
  // (the tile of the previous step has left by then, so the two never hold registers at the same time)
  // NCDHW: lane -> 16-byte piece (lane & 3) of line (cout 4i + lane/16, row (lane/4) & 3) for store instruction i
  unsigned char* patch = lds + 2 * V3_SLOTB + (W1_LDS ? 2 : 1) * V3_W2B + 128 + (Y_NCDHW ? (tid >> 6) * V3_PATCHB : 0);
  const int n_orow = (lane >> 2) & 3, n_piece = lane & 3, n_co = lane >> 4;
  const size_t cstride = (size_t)t_out * plane_out;
  const bool n_ok = (h0 + 4 * wr + n_orow) < h_out && (w0 + n_piece * 8) < w_out;
  const size_t n_base = ((size_t)b * c_out + 16 * ch + n_co) * cstride + (size_t)(h0 + 4 * wr + n_orow) * w_out + w0 + n_piece * 8;
  u32x4 pend[4];
  u32x4 og[(OUT_GATE && MASKM != 2) ? 4 : 1];
  uint32_t mbyte[MASKM != 0 ? 4 : 1];      // MASKM 1: the finished tile's mask bytes (stored with the tile); 2: the gate's
  // a lane's byte of the mask: voxel (row h0 + 4 wr + orow, column w0 + 16 (kg & 1) + vox), byte 2 ch + (kg >> 1) of its word
  const uint32_t mk_off = (uint32_t)(((h0 + 4 * wr) * mask_wp + w0 + 16 * (kg & 1) + vox) * 4 + 2 * ch + (kg >> 1));
  const uint32_t mk_plane_b = (uint32_t)mask_hp * mask_wp * 4u, mk_row_b = (uint32_t)mask_wp * 4u;
  unsigned char* const mk_sample = reinterpret_cast<unsigned char*>(mask) + (size_t)b * t_out * mk_plane_b;
  const int mk_sample_b = (int)((size_t)t_out * mk_plane_b);
  // F32OUT: a finished tile leaves straight from its accumulators at the end of the step that finishes it (8 stores; held for the
  // next step's side slots like the bf16 tile, its 32 registers + the new slice's accumulators do not fit: 181 spills).  The
  // wait at the top of the next step then lets exactly these 8 newest operations stay in flight (vmcnt counts in issue order)
  auto store_acc = [&](const f32x4& v, int o, int orow, int half, bool live) {
    const bool ok = live && orow < rows_left;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y_sample, 0, ok ? (int)out_sample_b : 0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, st_off_f[half],
                                           ok ? (uint32_t)o * out_plane_b + (uint32_t)orow * out_row_b : 0u, 0);
  };
  auto store_row = [&](int o, int orow, bool live) {   // NDHWC only; live: the tile exists (wave-uniform)
    const bool ok = live && orow < rows_left;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y_sample, 0, ok ? (int)out_sample_b : 0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(pend[orow], rs, st_off, ok ? (uint32_t)o * out_plane_b + (uint32_t)orow * out_row_b : 0u, 0);
  };
  auto store_mask_row = [&](int o, int orow, bool live) {      // MASKM 1 (the padded plane holds every tile row)
    if constexpr (MASKM == 1) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mk_sample, 0, live ? mk_sample_b : 0, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b8((unsigned char)mbyte[orow], rs, mk_off,
                                           live ? (uint32_t)o * mk_plane_b + (uint32_t)orow * mk_row_b : 0u, 0);
    }
  };
  auto load_gate_row = [&](int o, int orow) {
    if constexpr (MASKM == 2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mk_sample, 0, mk_sample_b, 0x00020000);
      mbyte[orow] = __builtin_amdgcn_raw_buffer_load_b8(rs, mk_off, (uint32_t)o * mk_plane_b + (uint32_t)orow * mk_row_b, 0);
    } else if constexpr (OUT_GATE) {
      const bool ok = orow < rows_left;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)og_sample, 0, ok ? (int)out_sample_b : 0, 0x00020000);
      og[orow] = __builtin_amdgcn_raw_buffer_load_b128(rs, st_off, ok ? (uint32_t)o * out_plane_b + (uint32_t)orow * out_row_b : 0u, 0);
    }
  };
  auto store_pending = [&](int o) {
    if constexpr (Y_NCDHW) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(patch + (4 * i + n_co) * V3_PATCH_CS + n_orow * 64 + n_piece * 16);
        if (n_ok && 16 * ch + 4 * i + n_co < c_out)
          *reinterpret_cast<u32x4*>(y + n_base + (size_t)(4 * i) * cstride + (size_t)o * plane_out) = v;
      }
    } else if constexpr (F32OUT) {
      // (every tile has left in the step that finished it)
    } else {
#pragma unroll
      for (int orow = 0; orow < 4; ++orow) store_row(o, orow, true);
#pragma unroll
      for (int orow = 0; orow < 4; ++orow) store_mask_row(o, orow, true);
    }
  };
  // accumulators of the three output slices in flight: acc[(o - tc0) % 3][row][half], 16 couts x 16 voxels each; the first
  // MFMA of an output slice takes the bias of couts 16*ch + 4*kg + reg as its C operand
  f32x4 acc[3][4][2];

  // ---- prologue ------------------------------------------------------------------------------------------------
  __syncthreads();  // lds_bias, lds_w2 written
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_bias + 16 * ch + 4 * kg);
  const int s_last = tc1 + 1;  // last input slice any output slice of this chunk needs
  const uint32_t relu_floor = relu ? 0u : 0x80008000u;

#ifdef PV_DIAG_STAMPS
  unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, q0, q1, q2, q3, q4, q5, q6, q7;
#endif

  // One step of the march.  The loads of slice s+1 (into the slot slice s-1 occupied) and the stores of output slice s-3
  // are issued right after the barrier and have the whole step to complete; the wait at the top of the next step then
  // costs nothing.  The gate of the tile a step finishes is requested after the first third of that step and applied when the
  // tile is converted at its end.
#define PV_V3_STEP(P, KT_LO, KT_HI)                                                                               \
  {                                                                                                               \
    PV_STAMP(q0);                                                                                                 \
    /* vmcnt(0): this wave's LDS-direct loads of slice s (and the gate) landed; F32OUT: vmcnt(8), see store_acc */ \
    if (F32OUT) __builtin_amdgcn_s_waitcnt(0x0f78); else __builtin_amdgcn_s_waitcnt(0x0f70);                      \
    PV_STAMP(q1);                                                                                                 \
    __syncthreads();                    /* ... everybody's; and every wave is done reading slice s-1 */            \
    PV_STAMP(q2);                                                                                                 \
    /* the step's vector-memory work, one instruction per MFMA row (side job j = 6 kw + ir): first the 4 stores of the   \
       tile the previous step finished (their registers are free again after four rows; NCDHW: one burst, the tile comes \
       from the LDS patch), then the gate of the tile THIS step finishes (consumed at the end of the step; never live    \
       together with the previous tile), then the 9 staging pieces of slice s+1 (into the slot slice s-1 occupied) */    \
    const bool do_load = s + 1 <= s_last, do_store = s - 3 >= tc0;                                                \
    auto side = [&](int j) {                                                                                      \
      if (F32OUT && j < 8) {                                                                                      \
      } else if (j < 4) {                                                                                         \
        if (!Y_NCDHW) store_row(s - 3, j, do_store);                                                              \
      } else if (j < 8) {                                                                                         \
        if (OUT_GATE && KT_HI == 2) load_gate_row(s - 2, j - 4);                                                  \
        if (MASKM == 1) store_mask_row(s - 3, j - 4, do_store);                                                   \
      } else if (j < 17) {                                                                                        \
        load_piece(s + 1, j - 8, do_load);                                                                        \
      }                                                                                                           \
    };                                                                                                            \
    auto no_side = [](int) {};                                                                                    \
    if (Y_NCDHW) { /* burst form: the tile comes out of the LDS patch, the staging pieces follow */               \
      if (do_store) store_pending(s - 3);                                                                         \
      if (do_load) load_slice(s + 1);                                                                             \
    }                                                                                                             \
    PV_STAMP(q3);                                                                                                 \
    const unsigned char* slot = lds + (s & 1) * V3_SLOTB + (4 * wr) * V3_ROWB;                                    \
    /* an input slice that lies in the time padding (dgrad: pad_t = 2 -> the two head and the two tail steps of a    \
       whole march) is all zeros: its MFMAs are skipped (wave-uniform branch), the bookkeeping of the step is not.    \
       Only the head / tail instances test it: an interior step of a chunk never reads padding of a whole march's    \
       ends unless the march has fewer than three slices, and its code stays branch-free */                          \
    const bool slice_live = quad_live && (((KT_LO) == 0 && (KT_HI) == 2) || (unsigned)(s - pad_t) < (unsigned)t_in); \
    if (slice_live && Y_NCDHW) {                                                                                  \
      v3_accumulate<P, KT_LO, KT_HI, 0, W1_LDS, F16>(slot, voff, wfrag, w2, acc, b4, no_side);                         \
      PV_STAMP(q4);                                                                                               \
      v3_accumulate<P, KT_LO, KT_HI, 1, W1_LDS, F16>(slot, voff, wfrag, w2, acc, b4, no_side);                         \
      v3_accumulate<P, KT_LO, KT_HI, 2, W1_LDS, F16>(slot, voff, wfrag, w2, acc, b4, no_side);                         \
    } else if (slice_live) {                                                                                      \
      v3_accumulate<P, KT_LO, KT_HI, 0, W1_LDS, F16>(slot, voff, wfrag, w2, acc, b4, side);                            \
      PV_STAMP(q4);                                                                                               \
      v3_accumulate<P, KT_LO, KT_HI, 1, W1_LDS, F16>(slot, voff, wfrag, w2, acc, b4, side);                            \
      v3_accumulate<P, KT_LO, KT_HI, 2, W1_LDS, F16>(slot, voff, wfrag, w2, acc, b4, side);                            \
    } else {                                                                                                      \
      if (!Y_NCDHW) { /* (first: F32OUT stores the previous tile from the accumulators armed below) */            \
        _Pragma("unroll") for (int j = 0; j < 18; ++j) side(j);                                                   \
      }                                                                                                           \
      if ((KT_LO) == 0) { /* output slice s receives nothing from this step: arm its accumulators */              \
        _Pragma("unroll") for (int r4 = 0; r4 < 4; ++r4) _Pragma("unroll") for (int half = 0; half < 2; ++half)   \
          acc[P][r4][half] = b4;                                                                                  \
      }                                                                                                           \
      PV_STAMP(q4);                                                                                               \
    }                                                                                                             \
    PV_STAMP(q6);                                                                                                 \
    if (KT_HI == 2 && F32OUT) { /* output slice s-2 is complete: its accumulators are the stores' data */         \
      /* (through a copy the compiler may not fold: stored straight from the accumulators, the matrix instructions of  \
         the interior steps stop accumulating in place and two slots' worth of registers goes to scratch) */          \
      _Pragma("unroll") for (int orow = 0; orow < 4; ++orow) _Pragma("unroll") for (int half = 0; half < 2; ++half) { \
        f32x4 tv = acc[((P) + 1) % 3][orow][half];                                                                \
        asm volatile("" : "+v"(tv));                                                                              \
        store_acc(tv, s - 2, orow, half, true);                                                                   \
      }                                                                                                           \
    } else if (KT_HI == 2) { /* output slice s-2 is complete: convert it into the store geometry */               \
      _Pragma("unroll") for (int orow = 0; orow < 4; ++orow) {                                                    \
        u32x2 o[2];                                                                                               \
        _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                  \
          const f32x4 a = acc[((P) + 1) % 3][orow][half];                                                         \
          o[half][0] = OUTM == 2 ? v3_pack_f16_pair(a[0], a[1]) : pack_bf16_pair(a[0], a[1]);                     \
          o[half][1] = OUTM == 2 ? v3_pack_f16_pair(a[2], a[3]) : pack_bf16_pair(a[2], a[3]);                     \
          o[half][0] = v3_pk_max(o[half][0], relu_floor);                                                         \
          o[half][1] = v3_pk_max(o[half][1], relu_floor);                                                         \
          if constexpr (Y_NCDHW) {                                                                                \
            unsigned char* pp = patch + (4 * kg) * V3_PATCH_CS + orow * 64 + (16 * half + vox) * 2;                 \
            *reinterpret_cast<uint16_t*>(pp) = (uint16_t)o[half][0];                                              \
            *reinterpret_cast<uint16_t*>(pp + V3_PATCH_CS) = (uint16_t)(o[half][0] >> 16);                        \
            *reinterpret_cast<uint16_t*>(pp + 2 * V3_PATCH_CS) = (uint16_t)o[half][1];                            \
            *reinterpret_cast<uint16_t*>(pp + 3 * V3_PATCH_CS) = (uint16_t)(o[half][1] >> 16);                    \
          }                                                                                                       \
        }                                                                                                         \
        if constexpr (!Y_NCDHW) {                                                                                 \
          /* odd 16-lane rows of half 0 <-> even rows of half 1: rows 0 / 2 now hold column half 0, rows 1 / 3 half 1, \
             every lane 8 consecutive couts (first the word pair that came from kg even, then the one from kg odd) */ \
          const auto r0 = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);                       \
          const auto r1 = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);                       \
          u32x4 v = {r0[0], r1[0], r0[1], r1[1]};                                                                 \
          if constexpr (MASKM == 2) {                                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) v[j] = v3_pk_gate_bits(v[j], mbyte[orow], 2 * j);       \
          } else if constexpr (OUT_GATE) {                                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) v[j] = v3_pk_gate(v[j], og[orow][j]);                   \
          }                                                                                                       \
          if constexpr (MASKM == 1) mbyte[orow] = relu_byte_of_pairs(v);                                                \
          pend[orow] = v;                                                                                         \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    PV_STAMP(q7);                                                                                                 \
    PV_V3_DIAG(0, q0, q1); /* vmcnt(0) */                                                                         \
    PV_V3_DIAG(1, q1, q2); /* barrier */                                                                          \
    PV_V3_DIAG(2, q2, q3); /* (NCDHW: tile stores) */                                                             \
    PV_V3_DIAG(3, q3, q4); /* MFMA kw = 0 + gate request + staging pieces */                                      \
    PV_V3_DIAG(5, q4, q6); /* MFMA kw = 1, 2 + staging pieces + tile stores */                                    \
    PV_V3_DIAG(6, q6, q7); /* convert + swap + gate */                                                           \
    PV_V3_DIAG(7, q0, q0 + 1); /* steps */                                                                        \
  }

  int s = tc0;
  PV_V3_STEP(0, 0, 0);
  ++s;
  PV_V3_STEP(1, 0, 1);
  ++s;
  for (;;) {  // interior steps: all three tap planes
    if (s >= tc1) break;
    PV_V3_STEP(2, 0, 2);
    ++s;
    if (s >= tc1) break;
    PV_V3_STEP(0, 0, 2);
    ++s;
    if (s >= tc1) break;
    PV_V3_STEP(1, 0, 2);
    ++s;
  }
  // tail: s = tc1 feeds output slices tc1-1, tc1-2; s = tc1+1 only tc1-1
  const int ptail = (tc1 - tc0) % 3;
  if (ptail == 0) {
    PV_V3_STEP(0, 1, 2);
    ++s;
    PV_V3_STEP(1, 2, 2);
  } else if (ptail == 1) {
    PV_V3_STEP(1, 1, 2);
    ++s;
    PV_V3_STEP(2, 2, 2);
  } else {
    PV_V3_STEP(2, 1, 2);
    ++s;
    PV_V3_STEP(0, 2, 2);
  }
#undef PV_V3_STEP
  store_pending(tc1 - 1);  // the tile the last step finished
#ifdef PV_DIAG_STAMPS
  {
    const int wgl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (lane == 0 && wgl * 4 + wave < PV_DIAG_WAVES)
      for (int i = 0; i < PV_DIAG_SLOTS; ++i) v3_diag[(size_t)(wgl * 4 + wave) * PV_DIAG_SLOTS + i] = dg[i];
  }
#endif
}

// w[Co,Ci,27] f32 -> 16x16x32 A fragments [27][2 cout halves][64 lanes][8]:
//   lane (co = lane&15, kg = lane>>4), element j  <-  W[cout = 16*half + co][cin = 8*kg + j][tap]
// transpose_flip: the dgrad operator, W'[cout' = ci][cin' = co][tap] = W[co][ci][26 - tap]
__global__ __launch_bounds__(256) void pack_weight_v3_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp2,
                                                              int c_out, int c_in, int transpose_flip) {
  const int total = 27 * 2 * 64 * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, half = (i >> 9) & 1, tap = i >> 10;
    const int row = 16 * half + (lane & 15);  // output channel of this operator
    const int k = 8 * (lane >> 4) + j;        // input channel of this operator
    float v = 0.f;
    if (!transpose_flip) {
      if (row < c_out && k < c_in) v = w[((size_t)row * c_in + k) * 27 + tap];
    } else {
      if (row < c_in && k < c_out) v = w[((size_t)k * c_in + row) * 27 + (26 - tap)];
    }
    wp2[i] = f32_to_bf16_bits(v);
  }
}

void launch_pack_weight_v3(const float* w, uint16_t* wp2, int c_out, int c_in, int transpose_flip, hipStream_t st) {
  hipLaunchKernelGGL(pack_weight_v3_kernel, dim3(54), dim3(256), 0, st, w, wp2, c_out, c_in, transpose_flip ? 1 : 0);
}

// Returns 1 (not a PV_* code) when the shape does not fit this kernel (fewer than 2 output slices per time chunk): the caller falls back
// to the one-wave-per-SIMD kernel of conv3d_bf16.hip.  The 1-bit relu masks of the C ABI: the plain NDHWC forward writes a requested
// mask itself (MASKM 1; v3_writes_mask()), the gated dgrad reads a mask that accompanies out_gate instead of the bf16 tensor (MASKM 2);
// the NCDHW-writing forward leaves a requested mask to the caller's pass over y.
bool v3_writes_mask(int y_ncdhw, const void* out_gate, const void*) { return !y_ncdhw && !out_gate; }

// grid of a v3 launch (tiles x time chunks x samples); false when no chunking leaves every chunk two output slices
static bool v3_grid(const pv_conv3d_dims* d, int to, int ho, int wo, dim3* grid_out, int* n_colblk_out, int* t_chunk_out) {
  const int n_rowblk = (ho + V3_TR - 1) / V3_TR;
  const int n_colblk = (wo + V3_TW_VALID - 1) / V3_TW_VALID;
  // two workgroups per CU: split the time march only when the (sample, tile) grid alone cannot fill 512 slots
  const long long tiles = (long long)d->batch * n_rowblk * n_colblk;
  // (a chunk of t output slices marches t + 2 input slices, the two extra ones with 1/3 and 2/3 of the taps: ~t + 1.5
  // steps; the cut that minimises rounds of 512 workgroups x steps wins -- 448 tiles are NOT cut: 1 x 11.5 < 2 x 6.5)
  int n_tchunk = 1;
  {
    long long best = -1;
    for (int c = 1; c <= (to / 2 > 0 ? to / 2 : 1) && tiles * c <= 8 * 512; ++c) {
      const int tch = (to + c - 1) / c, nch = (to + tch - 1) / tch;
      const long long rounds = (tiles * nch + 511) / 512;
      const long long cost = rounds * (2 * tch + 3);
      if (best < 0 || cost < best) best = cost, n_tchunk = nch;
    }
  }
  int t_chunk = (to + n_tchunk - 1) / n_tchunk;
  n_tchunk = (to + t_chunk - 1) / t_chunk;
  if (to - (n_tchunk - 1) * t_chunk < 2) {  // a single-slice remainder: fold it into longer chunks
    ++t_chunk;
    n_tchunk = (to + t_chunk - 1) / t_chunk;
    if (to - (n_tchunk - 1) * t_chunk < 2) return false;
  }
  *grid_out = dim3((unsigned)(n_rowblk * n_colblk), (unsigned)n_tchunk, (unsigned)d->batch);
  *n_colblk_out = n_colblk;
  *t_chunk_out = t_chunk;
  return true;
}

int launch_conv3d_fwd_bf16_v3(const uint16_t* x, const uint16_t* wp2, const float* bias, uint16_t* y,
                              const uint16_t* out_gate, const pv_conv3d_dims* d, int to, int ho, int wo, int relu,
                              int y_ncdhw, hipStream_t st, const uint32_t* out_gate_mask, uint32_t* mask_out) {
  if (to < 2) return 1;
  if (y_ncdhw && (wo % 8 != 0 || ((uintptr_t)y % 16) != 0)) return 1;  // 16-byte pieces of an output line
  if (!y_ncdhw && (((uintptr_t)y % 16) != 0 || ((uintptr_t)out_gate % 16) != 0)) return 1;
  dim3 grid;
  int n_colblk, t_chunk;
  if (!v3_grid(d, to, ho, wo, &grid, &n_colblk, &t_chunk)) return 1;
  const int mhp = (ho + V3_TR - 1) / V3_TR * V3_TR, mwp = (wo + V3_TW_VALID - 1) / V3_TW_VALID * V3_TW_VALID;      // pv_relu_mask_dims
#define PV_LAUNCH_V3(OG, YN, MM, MPTR)                                                                                  \
  hipLaunchKernelGGL((conv3d_fwd_bf16_v3_kernel<OG, YN, 0, MM>), grid, dim3(256), 0, st, x, wp2, bias, y, out_gate, d->t_in, \
                     d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, relu ? 1 : 0, n_colblk, t_chunk, d->c_out, \
                     MPTR, mhp, mwp)
  if (y_ncdhw) PV_LAUNCH_V3(false, true, 0, (uint32_t*)nullptr);
  else if (out_gate && out_gate_mask) PV_LAUNCH_V3(true, false, 2, const_cast<uint32_t*>(out_gate_mask));
  else if (out_gate) PV_LAUNCH_V3(true, false, 0, (uint32_t*)nullptr);
  else if (mask_out) PV_LAUNCH_V3(false, false, 1, mask_out);
  else PV_LAUNCH_V3(false, false, 0, (uint32_t*)nullptr);
#undef PV_LAUNCH_V3
  return check_launch("pv_conv3d_fwd_bf16(v3)");
}

bool v3_f32out_covers(const pv_conv3d_dims* d, int to, int ho, int wo) {
  dim3 grid;
  int n_colblk, t_chunk;
  return to >= 2 && (size_t)to * ho * wo * 128 <= 0x7fffffffull && v3_grid(d, to, ho, wo, &grid, &n_colblk, &t_chunk);
}

// One partial product of the f32-accurate form: x, wp2 half floats (a term of a two-term split each), y f32 NDHWC [B,To,Ho,Wo,32]
// = the raw accumulators.  Returns 1 when the shape does not fit (the caller keeps the f32 kernels).
// f16_out: y is a half-float image holding the accumulators times 2^-12 (OUTM 2) instead of the f32 accumulators.
int launch_conv3d_fwd_f16_f32out_v3(const uint16_t* x, const uint16_t* wp2, void* y, int f16_out, const pv_conv3d_dims* d, int to,
                                    int ho, int wo, hipStream_t st) {
  if (!v3_f32out_covers(d, to, ho, wo) || ((uintptr_t)y % 16) != 0 || ((uintptr_t)x % 16) != 0) return 1;
  dim3 grid;
  int n_colblk, t_chunk;
  if (!v3_grid(d, to, ho, wo, &grid, &n_colblk, &t_chunk)) return 1;
#define PV_LAUNCH_V3P(M)                                                                                                      \
  hipLaunchKernelGGL((conv3d_fwd_bf16_v3_kernel<false, false, M>), grid, dim3(256), 0, st, x, wp2, (const float*)nullptr,     \
                     reinterpret_cast<uint16_t*>(y), (const uint16_t*)nullptr, d->t_in, d->h_in, d->w_in, to, ho, wo, d->pad_t, \
                     d->pad_h, d->pad_w, 0, n_colblk, t_chunk, d->c_out)
  if (f16_out) PV_LAUNCH_V3P(2); else PV_LAUNCH_V3P(1);
#undef PV_LAUNCH_V3P
  return check_launch("pv_conv3d_fwd_f16_f32out(v3)");
}

}  // namespace pv

#ifdef PV_DIAG_STAMPS
extern "C" int pv_diag_read_v3(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::v3_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
