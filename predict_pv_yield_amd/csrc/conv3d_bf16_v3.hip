// bf16 MFMA Conv3D for 32 -> 32 channel layers, INPUT-STATIONARY time march (forward L1..L2 and every dgrad).
//
// conv3d_bf16_v2.hip computes one output slice at a time from the three input slices it needs: every B fragment
// (activations, 32 cin x 16 voxels) read from LDS feeds only the 3 kh taps of one kt plane, and the two waves of a
// cout pair read the same fragments -- 864 ds_read_b128 per slice and CU = 6912 LDS cycles, exactly the 6912 matrix-pipe
// cycles per SIMD: LDS bandwidth co-limits the kernel (measured matrix pipe busy 60 %).
//
// Here the march is turned around: each step takes ONE INPUT slice s and adds its contribution to the three output
// slices s, s-1, s-2 (tap planes kt = 0, 1, 2) that are in flight in registers.  A fragment is read once and feeds up
// to 3 kh x 3 kt MFMAs: a third of the LDS operand traffic.  To make room for 3 x 32 accumulator registers inside the
// 256-register budget of two waves per SIMD:
//   * the kt = 2 tap plane of the weights lives in LDS (3 fragment reads per (slice, kw)), kt = 0, 1 stay in registers;
//   * input slices are staged global -> LDS directly (buffer_load_dwordx4 ... lds, gfx950), no staging registers: the
//     LDS destination of a wave instruction is 1 KB linear by lane, so the XOR swizzle of the image is applied to the
//     SOURCE address of each lane instead;
//   * the finished tile goes from registers straight to global memory (8 B per lane = 4 consecutive couts of a voxel;
//     the two waves of a cout pair fill the two 32-byte halves of every voxel) -- no LDS staging, no second barrier.
// One barrier per slice, and because a step only needs the slice being read plus the one being fetched, the ring is
// TWO slots deep.  That shrinks the LDS footprint enough to run two independent 4-wave workgroups per CU (tile 8 rows x
// 32 columns each, one wave of each workgroup per SIMD): the two workgroups drift apart, so while one sits at its
// barrier or converts / stores a finished tile the other keeps the matrix pipe busy -- without the lock-step a single
// 8-wave workgroup imposes.  Same swizzled LDS image and weight fragments (pack_weight_v2_kernel) as v2.
#include "pv_common.h"

namespace pv {

constexpr int V3_TR = 8, V3_TRI = 10, V3_TW = 34, V3_TW_VALID = 32;  // input row 34 voxels -> 32 output columns
constexpr int V3_VOXB = 64, V3_ROWB = V3_TW * V3_VOXB, V3_SLOTB = V3_TRI * V3_ROWB;
constexpr int V3_W2B = 9 * 2 * 64 * 16;  // tap plane kt = 2: [9 taps][2 cout halves][64 lanes] x 16 B
constexpr uint32_t V3_INVALID = 0x40000000u;
constexpr int V3_PATCH_CS = 4 * 64 + 16;        // NCDHW epilogue patch: bytes per cout ([4 rows][32 voxels] bf16 + pad)
constexpr int V3_PATCHB = 16 * V3_PATCH_CS;     // per wave: 16 couts

// ReLU-derivative gate of a bf16 pair: 2 bits (low half > 0, high half > 0); and its application to a bf16 pair
__device__ __forceinline__ uint32_t v3_gate_bits(uint32_t g) { return relu_bits_of_pair(g); }
__device__ __forceinline__ uint32_t v3_apply_gate(uint32_t x, uint32_t bits) {
  return x & (((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u));
}

// ---- 1-bit ReLU masks ------------------------------------------------------------------------------------------
// relu_mask[b][t][h][w] (u32 per voxel): bit c = (activation of channel c > 0).  A forward launch writes it next to
// its bf16 output; the dgrad whose epilogue applies that ReLU's derivative reads 4 bytes per voxel instead of the 64-byte
// bf16 voxel (the gated dgrad was 16 us slower per launch than the ungated one: 85 MB of extra reads at B = 32).
// In this kernel a lane (vox, kg) of wave (ch, wr) owns channels 16ch + 4kg .. +3 of the 8 voxels (orow, half) of its
// tile rows: 8 nibbles = `gbits`, nibble index 2 * orow + half.  Memory wants, per voxel, the 16 bits of a channel
// half: the four kg lanes (16 lanes apart) transpose their 4 x 4 bytes (byte = the two nibbles of one orow) with two
// ds_bpermute rounds, after which lane kg holds row orow = kg complete.  The transpose is its own inverse, so the
// consumer runs the same routine on what it loaded.
__device__ __forceinline__ uint32_t v3_kg_transpose(uint32_t x, int lane) {
  const bool b0 = (lane >> 4) & 1, b1 = (lane >> 5) & 1;
  const uint32_t ev = (x & 0xffu) | ((x >> 8) & 0xff00u), od = ((x >> 8) & 0xffu) | ((x >> 16) & 0xff00u);
  const uint32_t keep = b0 ? od : ev, send = b0 ? ev : od;
  const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)send);
  const uint32_t lo = b0 ? recv : keep, hi = b0 ? keep : recv;
  const uint32_t y = (lo & 0xffu) | ((hi & 0xffu) << 8) | (((lo >> 8) & 0xffu) << 16) | (((hi >> 8) & 0xffu) << 24);
  const uint32_t keep2 = b1 ? (y >> 16) : (y & 0xffffu), send2 = b1 ? (y & 0xffffu) : (y >> 16);
  const uint32_t recv2 = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)send2);
  return b1 ? (recv2 | (keep2 << 16)) : (keep2 | (recv2 << 16));
}
// low nibbles of the 4 bytes of x <-> 16 contiguous bits
__device__ __forceinline__ uint32_t v3_nibbles_to_u16(uint32_t x) {
  uint32_t y = x & 0x0f0f0f0fu;
  y = (y | (y >> 4)) & 0x00ff00ffu;
  return (y | (y >> 8)) & 0xffffu;
}
__device__ __forceinline__ uint32_t v3_u16_to_nibbles(uint32_t v) {
  uint32_t y = v & 0xffffu;
  y = (y | (y << 8)) & 0x00ff00ffu;
  return (y | (y << 4)) & 0x0f0f0f0fu;
}

// One input slice's contribution to output slices s - kt, kt in [KT_LO, KT_HI] (compile-time: head and tail steps of
// the march feed fewer slices).  P = (s - tc0) % 3 names the accumulator slot of output slice s.
template <int P, int KT_LO, int KT_HI, int KW>
__device__ __forceinline__ void v3_accumulate(const unsigned char* slot, const int (&voff)[3], const bf16x8 (&wfrag)[18],
                                              const unsigned char* w2, f32x4 (&acc)[3][4][2]) {
  {
    constexpr int kw = KW;
    bf16x8 cur[2], nxt[2], wk2[3];
    if (KT_HI == 2) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) wk2[kh] = *reinterpret_cast<const bf16x8*>(w2 + (kh * 3 + kw) * 2048);
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
#pragma unroll
      for (int kt = KT_LO; kt <= KT_HI; ++kt) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int orow = ir - kh;
          if (orow >= 0 && orow < 4) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
              acc[(P - kt + 3) % 3][orow][half] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  kt == 2 ? wk2[kh] : wfrag[kt * 9 + kh * 3 + kw], cur[half], acc[(P - kt + 3) % 3][orow][half], 0, 0, 0);
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
// OUT_GATE: 0 = none, 1 = bf16 tensor of the gating activation, 2 = its 1-bit relu mask (out_gate then points to u32
// words).  MASK_OUT: also write the relu mask of THIS launch's output (forward, NDHWC).
template <int OUT_GATE, bool Y_NCDHW, bool MASK_OUT = false>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_bf16_v3_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ wp2, const float* __restrict__ bias,
    uint16_t* __restrict__ y, const uint16_t* __restrict__ out_gate, int t_in, int h_in, int w_in, int t_out,
    int h_out, int w_out, int pad_t, int pad_h, int pad_w, int relu, int n_colblk, int t_chunk, int c_out,
    uint32_t* __restrict__ mask_out = nullptr) {
  // ring of 2 slices | kt = 2 weight plane | 32 bias floats   (78 KB: two workgroups per CU)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * V3_SLOTB + V3_W2B + 128 + (Y_NCDHW ? 4 * V3_PATCHB : 0)];
  unsigned char* lds_w2 = lds + 2 * V3_SLOTB;
  float* lds_bias = reinterpret_cast<float*>(lds_w2 + V3_W2B);
  static_assert(!(OUT_GATE && Y_NCDHW), "the gated epilogue writes NDHWC");
  static_assert(!(MASK_OUT && (Y_NCDHW || OUT_GATE)), "the relu mask is written by the plain NDHWC forward");

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
  if (tc1 - tc0 < 2) return;  // the launcher sends single-slice chunks to the v2 kernel

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
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * sample_elems), 0, (int)(sample_elems * 2), 0x00020000);
  auto load_slice = [&](int s) {
    const int ti = s - pad_t;
    const bool t_ok = (unsigned)ti < (unsigned)t_in;
    const uint32_t toff = (uint32_t)min(max(ti, 0), t_in - 1) * x_plane_b;
    unsigned char* dst = lds + (s & 1) * V3_SLOTB + wave * V3_ROWB;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (wave + 4 * i < V3_TRI) {
        const int hi = h0 - pad_h + 4 * i + wave;
        const bool row_ok = t_ok && (unsigned)hi < (unsigned)h_in;
        const uint32_t srow = toff + (uint32_t)min(max(hi, 0), h_in - 1) * x_row_b + (row_ok ? 0u : V3_INVALID);
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + 4 * i * V3_ROWB), 16, lane_voff[0] + srow, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + 4 * i * V3_ROWB + 1024), 16, lane_voff[1] + srow, 0,
                                                 0, 0);
        if (lane < 8)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + 4 * i * V3_ROWB + 2048), 16, lane_voff[2] + srow,
                                                   0, 0, 0);
      }
    }
  };

  load_slice(tc0);  // first: its HBM latency hides under the weight loads below

  // ---- weights: tap planes kt = 0, 1 resident in registers (18 A fragments, 16 couts x 32 cin), kt = 2 in LDS ----
  bf16x8 wfrag[18];
#pragma unroll
  for (int tap = 0; tap < 18; ++tap)
    wfrag[tap] = *reinterpret_cast<const bf16x8*>(wp2 + ((size_t)(tap * 2 + ch) * 64 + lane) * 8);
  for (int i = tid; i < V3_W2B / 16; i += 256)
    reinterpret_cast<u32x4*>(lds_w2)[i] = reinterpret_cast<const u32x4*>(wp2 + (size_t)18 * 2 * 64 * 8)[i];
  const unsigned char* w2 = lds_w2 + ch * 1024 + lane * 16;

  // ---- per-lane LDS read offsets of the B operand: voxel 16*half + vox + kw, 16-byte chunk kg ---------------------
  // (the column half 16 voxels further right has the same swizzle phase: + 1024 bytes, an immediate of the ds_read)
  int voff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int v = vox + kw;
    voff[kw] = v * V3_VOXB + ((kg ^ (((v >> 2) & 1) << 1)) << 4);
  }

  // ---- write-out geometry: lane (vox, kg) of accumulator [orow][half] holds couts 16ch + 4kg .. +3 of voxel
  // (row h0 + 4wr + orow, column w0 + 16half + vox): 8 bytes ----------------------------------------------------
  const int plane_out = h_out * w_out;
  uint32_t st_off[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int col_t = 16 * half + vox;
    const bool ok = (w0 + col_t) < w_out;
    st_off[half] = ok ? ((uint32_t)((h0 + 4 * wr) * w_out + w0 + col_t) * 32u + 16u * ch + 4u * kg) * 2u : V3_INVALID;
  }
  const size_t out_sample_b = (size_t)t_out * plane_out * 64;
  const __amdgpu_buffer_rsrc_t yrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)b * t_out * plane_out * 32), 0, (int)out_sample_b, 0x00020000);
  const __amdgpu_buffer_rsrc_t ogrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((OUT_GATE == 1 ? out_gate : y) + (size_t)b * t_out * plane_out * 32), 0, (int)out_sample_b, 0x00020000);
  // relu-mask addressing (read: OUT_GATE == 2, write: MASK_OUT): lane (vox, kg) handles tile row orow = kg, the voxels
  // of columns vox and 16 + vox, 16-bit half ch of the per-voxel word
  // The mask planes are padded to whole tiles ([B][T][ceil8(H)][ceil32(W)] words, pv_relu_mask_dims), so every lane's
  // address is in range and the second column half is a constant 64 bytes further.
  const int mask_w = n_colblk * V3_TW_VALID, mask_plane = ((h_out + V3_TR - 1) / V3_TR) * V3_TR * mask_w;
  const size_t mask_sample_b = (size_t)t_out * mask_plane * 4;
  const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((OUT_GATE == 2 ? reinterpret_cast<const unsigned char*>(out_gate) : reinterpret_cast<const unsigned char*>(mask_out)) +
              (size_t)b * mask_sample_b),
      0, (OUT_GATE == 2 || MASK_OUT) ? (int)mask_sample_b : 0, 0x00020000);
  const uint32_t m_off = (uint32_t)((h0 + 4 * wr + kg) * mask_w + w0 + vox) * 4u + 2u * ch;
  auto row_off = [&](int o, int orow) -> uint32_t {  // wave-uniform part of the byte offset of (slice o, tile row orow)
    const bool ok = (h0 + 4 * wr + orow) < h_out;
    return ok ? (uint32_t)o * (uint32_t)plane_out * 64u + (uint32_t)orow * (uint32_t)w_out * 64u : V3_INVALID;
  };

  u32x2 pend[4][2];  // finished tile (bf16 pairs), stored one step later so the stores never sit in front of a wait
  u32x2 og[OUT_GATE == 1 ? 4 : 1][2];
  // MASK_OUT: the finished tile's relu bits of (row kg, column vox) in the low and (row kg, column 16 + vox) in the high
  // half, stored with the tile; OUT_GATE 2: the same two 16-bit words as loaded
  uint32_t mpend = 0u, mpend_hi = 0u;
  uint32_t gbits = 0u;  // the tile's gate, 4 bits per (row, half): the 16 gate registers live only through the kw = 0 phase
  // NCDHW: lane -> 16-byte piece (lane & 3) of line (cout 4i + lane/16, row (lane/4) & 3) for store instruction i
  unsigned char* patch = lds + 2 * V3_SLOTB + V3_W2B + 128 + (Y_NCDHW ? (tid >> 6) * V3_PATCHB : 0);
  const int n_orow = (lane >> 2) & 3, n_piece = lane & 3, n_co = lane >> 4;
  const size_t cstride = (size_t)t_out * plane_out;
  const bool n_ok = (h0 + 4 * wr + n_orow) < h_out && (w0 + n_piece * 8) < w_out;
  const size_t n_base = ((size_t)b * c_out + 16 * ch + n_co) * cstride + (size_t)(h0 + 4 * wr + n_orow) * w_out + w0 + n_piece * 8;
  auto store_pending = [&](int o) {
    if constexpr (Y_NCDHW) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(patch + (4 * i + n_co) * V3_PATCH_CS + n_orow * 64 + n_piece * 16);
        if (n_ok && 16 * ch + 4 * i + n_co < c_out)
          *reinterpret_cast<u32x4*>(y + n_base + (size_t)(4 * i) * cstride + (size_t)o * plane_out) = v;
      }
    } else {
#pragma unroll
      for (int orow = 0; orow < 4; ++orow) {
        const uint32_t ro = row_off(o, orow);
#pragma unroll
        for (int half = 0; half < 2; ++half)
          __builtin_amdgcn_raw_buffer_store_b64(pend[orow][half], yrsrc, st_off[half] + ro, 0, 0);
      }
      if constexpr (MASK_OUT) {
        // relu bits of the tile being stored, taken from the very registers that are stored (nothing extra is carried
        // across the step); the kg transpose runs here, at the top of the step, where register pressure is lowest
        uint32_t obits = 0u;
#pragma unroll
        for (int orow = 0; orow < 4; ++orow)
#pragma unroll
          for (int half = 0; half < 2; ++half)
          {
            const uint32_t t = relu_pair01(pend[orow][half][0]) | (relu_pair01(pend[orow][half][1]) << 2);   // bits 0, 16, 2, 18
            obits |= ((t | (t >> 15)) & 0xfu) << (4 * (2 * orow + half));
          }
        const uint32_t tr = v3_kg_transpose(obits, lane);  // bytes [kg']: (nibble of column vox | nibble of column 16 + vox << 4)
        const uint32_t mo = m_off + (uint32_t)o * (uint32_t)mask_plane * 4u;
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v3_nibbles_to_u16(tr), mrsrc, mo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v3_nibbles_to_u16(tr >> 4), mrsrc, mo + 64u, 0, 0);
      }
    }
  };
  auto load_gate = [&](int o) {
    if constexpr (OUT_GATE == 1) {
#pragma unroll
      for (int orow = 0; orow < 4; ++orow) {
        const uint32_t ro = row_off(o, orow);
#pragma unroll
        for (int half = 0; half < 2; ++half) og[orow][half] = __builtin_amdgcn_raw_buffer_load_b64(ogrsrc, st_off[half] + ro, 0, 0);
      }
    } else if constexpr (OUT_GATE == 2) {
      const uint32_t mo = m_off + (uint32_t)o * (uint32_t)mask_plane * 4u;
      mpend = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(mrsrc, mo, 0, 0);
      mpend_hi = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(mrsrc, mo + 64u, 0, 0);
    }
  };

  // accumulators of the three output slices in flight: acc[(o - tc0) % 3][row][half], 16 couts x 16 voxels each,
  // initialised with the bias of couts 16*ch + 4*kg + reg
  f32x4 acc[3][4][2];

  // ---- prologue ------------------------------------------------------------------------------------------------
  __syncthreads();  // lds_bias, lds_w2 written
  {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_bias + 16 * ch + 4 * kg);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int half = 0; half < 2; ++half) acc[j][r4][half] = b4;
  }
  const int s_last = tc1 + 1;  // last input slice any output slice of this chunk needs

  // One step of the march.  The loads of slice s+1 (into the slot slice s-1 occupied) and the stores of output slice s-3
  // are issued right after the barrier and have the whole step to complete; the wait at the top of the next step then
  // costs nothing.
#define PV_V3_STEP(P, KT_LO, KT_HI)                                                                               \
  {                                                                                                               \
    __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0): this wave's LDS-direct loads of slice s have landed */        \
    __syncthreads();                    /* ... everybody's; and every wave is done reading slice s-1 */            \
    if (s + 1 <= s_last) load_slice(s + 1);                                                                       \
    if (s - 3 >= tc0) store_pending(s - 3);                                                                       \
    if (OUT_GATE == 1 && KT_HI == 2) load_gate(s - 2);                                                            \
    const unsigned char* slot = lds + (s & 1) * V3_SLOTB + (4 * wr) * V3_ROWB;                                    \
    /* an input slice that lies in the time padding (dgrad: pad_t = 2 -> the two head and the two tail steps of a    \
       whole march) is all zeros: its MFMAs are skipped (wave-uniform branch), the bookkeeping of the step is not.    \
       Only the head / tail instances test it: an interior step of a chunk never reads padding of a whole march's    \
       ends unless the march has fewer than three slices, and its code stays branch-free */                          \
    const bool slice_live = ((KT_LO) == 0 && (KT_HI) == 2) || (unsigned)(s - pad_t) < (unsigned)t_in;             \
    if (slice_live) v3_accumulate<P, KT_LO, KT_HI, 0>(slot, voff, wfrag, w2, acc);                                \
    if constexpr (OUT_GATE == 1 && KT_HI == 2) {                                                                  \
      gbits = 0u;                                                                                                 \
      _Pragma("unroll") for (int orow = 0; orow < 4; ++orow) _Pragma("unroll") for (int half = 0; half < 2; ++half) \
        gbits |= (v3_gate_bits(og[orow][half][0]) | (v3_gate_bits(og[orow][half][1]) << 2)) << (4 * (2 * orow + half)); \
    }                                                                                                             \
    if constexpr (OUT_GATE == 2 && KT_HI == 2) { /* row kg of both column halves -> this lane's 8 nibbles */       \
      gbits = v3_kg_transpose(v3_u16_to_nibbles(mpend) | (v3_u16_to_nibbles(mpend_hi) << 4), lane);                \
    }                                                                                                             \
    if (slice_live) {                                                                                             \
      v3_accumulate<P, KT_LO, KT_HI, 1>(slot, voff, wfrag, w2, acc);                                              \
      v3_accumulate<P, KT_LO, KT_HI, 2>(slot, voff, wfrag, w2, acc);                                              \
    }                                                                                                             \
    if (KT_HI == 2) { /* output slice s-2 is complete: convert it, re-arm its accumulators */                     \
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_bias + 16 * ch + 4 * kg);                              \
      _Pragma("unroll") for (int orow = 0; orow < 4; ++orow) _Pragma("unroll") for (int half = 0; half < 2; ++half) { \
        f32x4 a = acc[((P) + 1) % 3][orow][half];                                                                 \
        acc[((P) + 1) % 3][orow][half] = b4;                                                                      \
        if (relu) {                                                                                               \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) a[j] = a[j] > 0.f ? a[j] : 0.f;                           \
        }                                                                                                         \
        u32x2 o;                                                                                                  \
        o[0] = pack_bf16_pair(a[0], a[1]);                       \
        o[1] = pack_bf16_pair(a[2], a[3]);                       \
        if constexpr (OUT_GATE != 0) {                                                                            \
          o[0] = v3_apply_gate(o[0], gbits >> (4 * (2 * orow + half)));                                           \
          o[1] = v3_apply_gate(o[1], gbits >> (4 * (2 * orow + half) + 2));                                       \
        }                                                                                                         \
        if constexpr (Y_NCDHW) {                                                                                  \
          unsigned char* pp = patch + (4 * kg) * V3_PATCH_CS + orow * 64 + (16 * half + vox) * 2;                   \
          *reinterpret_cast<uint16_t*>(pp) = (uint16_t)o[0];                                                      \
          *reinterpret_cast<uint16_t*>(pp + V3_PATCH_CS) = (uint16_t)(o[0] >> 16);                                \
          *reinterpret_cast<uint16_t*>(pp + 2 * V3_PATCH_CS) = (uint16_t)o[1];                                    \
          *reinterpret_cast<uint16_t*>(pp + 3 * V3_PATCH_CS) = (uint16_t)(o[1] >> 16);                            \
        } else {                                                                                                  \
          pend[orow][half] = o;                                                                                   \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    if constexpr (OUT_GATE == 2 && KT_HI >= 1) load_gate(s - 1); /* mask of the slice the NEXT step finishes */   \
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
}

// Returns 1 (not a PV_* code) when the shape does not fit this kernel (fewer than 2 output slices per time chunk): the caller falls back
// to the v2 kernel.
// out_gate_mask (u32 per voxel of y) takes precedence over out_gate; mask_out (may be NULL) receives the relu mask of y
// and is only honoured for the plain NDHWC forward (no gate): the caller checks v3_writes_mask().
bool v3_writes_mask(int y_ncdhw, const void* out_gate, const void* out_gate_mask) { return !y_ncdhw && !out_gate && !out_gate_mask; }

int launch_conv3d_fwd_bf16_v3(const uint16_t* x, const uint16_t* wp2, const float* bias, uint16_t* y,
                              const uint16_t* out_gate, const pv_conv3d_dims* d, int to, int ho, int wo, int relu,
                              int y_ncdhw, hipStream_t st, const uint32_t* out_gate_mask, uint32_t* mask_out) {
  if (to < 2) return 1;
  if (y_ncdhw && (wo % 8 != 0 || ((uintptr_t)y % 16) != 0)) return 1;  // 16-byte pieces of an output line
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
    if (to - (n_tchunk - 1) * t_chunk < 2) return 1;
  }
  dim3 grid((unsigned)(n_rowblk * n_colblk), (unsigned)n_tchunk, (unsigned)d->batch);
  const uint16_t* og_ptr = out_gate_mask ? reinterpret_cast<const uint16_t*>(out_gate_mask) : out_gate;
#define PV_LAUNCH_V3(OG, YN, MO)                                                                                      \
  hipLaunchKernelGGL((conv3d_fwd_bf16_v3_kernel<OG, YN, MO>), grid, dim3(256), 0, st, x, wp2, bias, y, og_ptr, d->t_in,   \
                     d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, relu ? 1 : 0, n_colblk, t_chunk, d->c_out, \
                     mask_out)
  if (y_ncdhw) PV_LAUNCH_V3(0, true, false);
  else if (out_gate_mask) PV_LAUNCH_V3(2, false, false);
  else if (out_gate) PV_LAUNCH_V3(1, false, false);
  else if (mask_out) PV_LAUNCH_V3(0, false, true);
  else PV_LAUNCH_V3(0, false, false);
#undef PV_LAUNCH_V3
  return check_launch("pv_conv3d_fwd_bf16(v3)");
}

}  // namespace pv
