// First Conv3D layer of the bf16 path straight from the reference's f32 NCDHW input, second form: loader waves.
// replaces: F.relu(sat_conv0(sat_data.float())) of predict_pv_yield/models/conv3d/model.py:113-118 (c_in <= 16).
//
// conv3d_bf16.hip's X_F32 variant does everything in four waves: fetch 11 channel planes, round to bf16, write the NDHWC
// image into LDS, copy it out for the weight gradient, multiply, transpose the tile through LDS, store.  In-kernel stamps
// (tools/diag_stamps.py first) showed the matrix phase at half of a slice: with one wave per SIMD every other phase is
// dead time for the matrix pipe.  Here the work is split by wave, as in conv3d_wgrad_bf16_v2.hip:
//   * waves 4..7 stage: per slice 160 tasks of (input row, 4 consecutive voxels) x all channels = one dwordx4 load per real
//     channel, rounded to bf16 and written as the swizzled NDHWC16 image (the one conv3d_bf16.hip reads) into a FOUR-slot
//     ring, one slice ahead of the multiplication; the NDHWC bf16 copy of the input that the weight gradient reads (xp_out)
//     leaves from the same registers (128 contiguous bytes per task);
//   * waves 0..3 multiply (v_mfma_f32_32x32x16_bf16, weights resident: 27 A fragments, a wave owns 4 output rows x 32
//     columns) and store: bias is the first MFMA's C operand, ReLU is a packed integer max on the bf16 pairs, and the tile
//     leaves WITHOUT an LDS transpose -- the two 32-lane halves of a wave swap registers (v_permlane32_swap) so that a lane
//     holds 8 consecutive output channels = 16 bytes of a voxel;
//   * one barrier per output slice.
// Bit-identical to pv_pack_ncdhw_f32_to_ndhwc_bf16 + pv_conv3d_fwd_bf16 (same rounding points, same order of the 27 x 16
// products per output).  pad_w == 0: other requests keep the one-role kernel.  A requested relu mask (round 6) leaves with the tile.
#include "pv_common.h"

namespace pv {

constexpr int F_TR = 8, F_TRI = 10, F_TW = 64, F_TW_VALID = 62;
constexpr int F_VOXB = 32, F_ROWB = F_TW * F_VOXB, F_SLOTB = F_TRI * F_ROWB;   // 2048, 20480
constexpr uint32_t F_INVALID = 0x40000000u;

#ifdef PV_DIAG_STAMPS
__device__ unsigned long long first_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif

__device__ __forceinline__ uint32_t f_pk_max(uint32_t x, uint32_t floor2) {   // signed 16-bit max on both halves
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "s"(floor2));
  return r;
}

__global__ __launch_bounds__(512, 2) void conv3d_first_f32in_kernel(
    const float* __restrict__ x, const uint16_t* __restrict__ wp, const float* __restrict__ bias, uint16_t* __restrict__ y,
    uint16_t* __restrict__ xp_out, int c_in, int t_in, int h_in, int w_in, int t_out, int h_out, int w_out, int pad_t,
    int pad_h, int relu, int n_colblk, int t_chunk, int c_out, uint32_t* __restrict__ mask_out, int mask_hp, int mask_wp) {
  // ring of 4 slices | 256 B of zeros (tap reads of the last columns run 2 voxels past a slot) | 32 bias floats
  __shared__ __attribute__((aligned(256))) unsigned char lds[4 * F_SLOTB + 256 + 128];
  float* lds_bias = reinterpret_cast<float*>(lds + 4 * F_SLOTB + 256);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;

  const int rowblk = blockIdx.x / n_colblk;
  const int colblk = blockIdx.x - rowblk * n_colblk;
  const int h0 = rowblk * F_TR;
  const int w0 = colblk * F_TW_VALID;
  const int b = blockIdx.z;
  const int tc0 = blockIdx.y * t_chunk;
  const int tc1 = min(tc0 + t_chunk, t_out);
  if (tc0 >= tc1) return;

  if (tid < 64) reinterpret_cast<uint32_t*>(lds + 4 * F_SLOTB)[tid] = 0u;
  if (tid >= 64 && tid < 96) lds_bias[tid - 64] = (bias && tid - 64 < c_out) ? bias[tid - 64] : 0.f;

  if (loader) {
    // =============================== staging waves =====================================================================
    // task = (tile row 0..9, quad of 4 consecutive voxels): 160 per slice, 40 per wave (lanes 0..39)
    const int task = wave * 40 + lane;
    const bool has = lane < 40;
    const int f_row = task >> 4, f_col = 4 * (task & 15);
    const size_t f32_sample_b = (size_t)c_in * t_in * h_in * w_in * 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(x) + (size_t)b * f32_sample_b), 0, (int)f32_sample_b, 0x00020000);
    const uint32_t f_plane_b = (uint32_t)t_in * h_in * w_in * 4u, f_slice_b = (uint32_t)h_in * w_in * 4u, f_row_b = (uint32_t)w_in * 4u;
    const int hi = h0 - pad_h + f_row;
    const bool row_in = (unsigned)hi < (unsigned)h_in;
    // (a quad starts inside the image or is wholly outside it: w_in % 4 == 0 is checked by the launcher)
    const uint32_t lane_base = (has && row_in && w0 + f_col < w_in) ? (uint32_t)hi * f_row_b + (uint32_t)(w0 + f_col) * 4u : F_INVALID;
    const bool own_last_row = (h0 + F_TR >= h_out), own_last_col = (w0 + F_TW_VALID >= w_out), own_last_t = (tc1 >= t_out);
    // two register sets: the loads of a slice are issued TWO slices before it is written to the ring (a slice's 11 loads per
    // lane take longer than one slice of matrix work once every CU streams)
    f32x4 stA[16], stB[16];
    auto load_slice = [&](int s, f32x4 (&st)[16]) {
      const int ti = s - pad_t;
      const uint32_t toff = (unsigned)ti < (unsigned)t_in ? (uint32_t)ti * f_slice_b : F_INVALID;
#pragma unroll
      for (int ch = 0; ch < 16; ++ch) {
        if (ch < c_in)   // uniform: absent channels are never fetched
          st[ch] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_base + toff + (uint32_t)ch * f_plane_b, 0, 0));
        else
          st[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto store_slice = [&](int s, const f32x4 (&st)[16]) {
      if (!has) return;
      unsigned char* slot = lds + (s & 3) * F_SLOTB + f_row * F_ROWB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int scol = f_col + q;
        const bool col_ok = w0 + scol < w_in;
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
          u32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pack_bf16_pair(st[8 * sc + 2 * e][q], st[8 * sc + 2 * e + 1][q]);
          if (!col_ok) v = (u32x4){0u, 0u, 0u, 0u};
          // swizzle of conv3d_bf16.hip's CPAD-16 image: chunk position = sc ^ ((voxel / 8) % 2)
          *reinterpret_cast<u32x4*>(slot + scol * F_VOXB + ((sc ^ ((scol >> 3) & 1)) << 4)) = v;
        }
      }
    };
    // xp_out (the NDHWC bf16 image of the input for the weight gradient) leaves from a ring slot that is COMPLETE (written
    // before the last barrier), lane-contiguous: lane -> (voxel lane / 2, chunk lane % 2) of a 32-voxel run, i.e. one
    // instruction writes 1 KB of xp; the rows of a slot are dealt to the four loader waves (5 half-rows each).  Every input
    // voxel is written by exactly one workgroup: the tile it belongs to as an OUTPUT position; the last row / column block
    // and time chunk also own the 2-voxel rim.
    auto write_xp = [&](int s) {
      const int ti = s - pad_t;
      if (!xp_out || (unsigned)ti >= (unsigned)t_in || !(s < tc1 || own_last_t)) return;
      const unsigned char* slot = lds + (s & 3) * F_SLOTB;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int hr = wave * 5 + i;               // half-row 0..19
        const int rowi = hr >> 1, scol = 32 * (hr & 1) + (lane >> 1), sc = lane & 1;
        const int hi2 = h0 - pad_h + rowi, swi = w0 + scol;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slot + rowi * F_ROWB + scol * F_VOXB + ((sc ^ ((scol >> 3) & 1)) << 4));
        if ((unsigned)hi2 < (unsigned)h_in && swi < w_in && (rowi < F_TR || own_last_row) && (scol < F_TW_VALID || own_last_col))
          *reinterpret_cast<u32x4*>(xp_out + ((((size_t)b * t_in + ti) * h_in + hi2) * w_in + swi) * 16 + sc * 8) = v;
      }
    };
    // prologue: slices tc0 .. tc0 + 2 into the ring, tc0 + 3 and tc0 + 4 in flight
    load_slice(tc0, stA);
    load_slice(tc0 + 1, stB);
    store_slice(tc0, stA);
    load_slice(tc0 + 2, stA);
    store_slice(tc0 + 1, stB);
    load_slice(tc0 + 3, stB);
    store_slice(tc0 + 2, stA);
    load_slice(tc0 + 4, stA);
    // iteration t (after the barrier that lets slice t be multiplied): slice t + 3 -> ring (slot of slice t - 1, free now),
    // loads of slice t + 5, xp of a complete slice.  Register sets alternate: slice parity relative to tc0 picks the set.
    for (int t = tc0; t < tc1; t += 2) {
      __syncthreads();
      if (t == tc0) {
        write_xp(tc0);
        write_xp(tc0 + 1);
      }
      store_slice(t + 3, stB);
      load_slice(t + 5, stB);
      write_xp(t + 2);
      if (t + 1 < tc1) {
        __syncthreads();
        store_slice(t + 4, stA);
        load_slice(t + 6, stA);
        write_xp(t + 3);
      }
    }
    return;
  }

  // =============================== multiplying waves ===================================================================
  const int r = lane & 31, hh = lane >> 5;
  const int ws = wave & 1;   // column segment (32 columns)
  const int wr = wave >> 1;  // row half (4 rows)
  bf16x8 wfrag[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) wfrag[tap] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)tap * 64 + lane) * 8);
  // per-lane LDS read offsets (bytes inside a slot row 0): voxel 32 ws + r + kw, chunk hh
  int voff[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int v = 32 * ws + r + kw;
    voff[kw] = v * F_VOXB + ((hh ^ ((v >> 3) & 1)) << 4);
  }
  // store geometry after the half swap: lane (r, hh) holds couts 8 hh .. 8 hh + 7 (first store) and 16 + 8 hh .. (second)
  const int plane_out = h_out * w_out;
  const int col_t = 32 * ws + r;
  const bool col_ok = col_t < F_TW_VALID && (w0 + col_t) < w_out;
  const uint32_t st_off = col_ok ? ((uint32_t)((h0 + 4 * wr) * w_out + w0 + col_t) * 32u + 8u * hh) * 2u : F_INVALID;
  const size_t out_sample_b = (size_t)t_out * plane_out * 64;
  void* const y_sample = y + (size_t)b * t_out * plane_out * 32;
  const int rows_left = h_out - (h0 + 4 * wr);
  const uint32_t out_plane_b = (uint32_t)plane_out * 64u, out_row_b = (uint32_t)w_out * 64u;
  const uint32_t relu_floor = relu ? 0u : 0x80008000u;
  // relu mask of the output (u32 per voxel, bit c = channel c > 0; pv_relu_mask_dims): after the half swap lane (r, hh) holds
  // bytes hh and 2 + hh of its voxel's word; the hh = 1 lanes hand theirs over and the hh = 0 lanes store the word
  const bool mk_ok = mask_out && hh == 0 && col_t < F_TW_VALID && (w0 + col_t) < mask_wp;
  const uint32_t mk_off = mk_ok ? (uint32_t)((h0 + 4 * wr) * mask_wp + w0 + col_t) * 4u : F_INVALID;
  const uint32_t mk_plane_b = (uint32_t)mask_hp * mask_wp * 4u;
  void* const mk_sample = reinterpret_cast<unsigned char*>(mask_out) + (size_t)b * t_out * mk_plane_b;

#ifdef PV_DIAG_STAMPS
  unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, q0, q1, q2, q3;
#endif
  f32x16 bias16;
  bool have_bias = false;
  for (int t = tc0; t < tc1; ++t) {
    PV_STAMP(q0);
    __syncthreads();
    PV_STAMP(q1);
    if (!have_bias) {   // (lds_bias is visible after the first barrier) row(reg j, half hh) = (j&3) + 8*(j>>2) + 4*hh
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(lds_bias + 8 * q + 4 * hh);
        bias16[4 * q] = bq[0]; bias16[4 * q + 1] = bq[1]; bias16[4 * q + 2] = bq[2]; bias16[4 * q + 3] = bq[3];
      }
      have_bias = true;
    }
    f32x16 acc[4];
    const unsigned char* slot_kt[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) slot_kt[kt] = lds + ((t + kt) & 3) * F_SLOTB + (4 * wr) * F_ROWB;
    // 9 (kt, kw) groups of 6 operand reads + 12 MFMAs, software-pipelined by one group
    bf16x8 bfr[2][6];
    auto read_group = [&](int g, bf16x8* dst) {
      const int kt = g / 3, kw = g % 3;
      const unsigned char* p = slot_kt[kt] + voff[kw];
#pragma unroll
      for (int ir = 0; ir < 6; ++ir) dst[ir] = *reinterpret_cast<const bf16x8*>(p + ir * F_ROWB);
    };
    read_group(0, bfr[0]);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      if (g + 1 < 9) read_group(g + 1, bfr[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int kt = g / 3, kw = g % 3;
#pragma unroll
      for (int ir = 0; ir < 6; ++ir) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int orow = ir - kh;
          if (orow >= 0 && orow < 4)   // the first tap an output row meets (group 0, kh 0) starts from the bias
            acc[orow] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[kt * 9 + kh * 3 + kw], bfr[g & 1][ir],
                                                                (g == 0 && kh == 0) ? bias16 : acc[orow], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    PV_STAMP(q2);
    // ---- epilogue: bf16, ReLU on the packed pairs, half swap, two 16-byte stores per tile row ------------------------
#pragma unroll
    for (int orow = 0; orow < 4; ++orow) {
      uint32_t P[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        P[q][0] = f_pk_max(pack_bf16_pair(acc[orow][4 * q], acc[orow][4 * q + 1]), relu_floor);
        P[q][1] = f_pk_max(pack_bf16_pair(acc[orow][4 * q + 2], acc[orow][4 * q + 3]), relu_floor);
      }
      const bool ok = orow < rows_left;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y_sample, 0, ok ? (int)out_sample_b : 0, 0x00020000);
      const uint32_t so = ok ? (uint32_t)t * out_plane_b + (uint32_t)orow * out_row_b : 0u;
      uint32_t mword = 0;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {   // cout quads (0, 1) and (2, 3): upper half of the first <-> lower half of the second
        const auto a = __builtin_amdgcn_permlane32_swap(P[2 * pr][0], P[2 * pr + 1][0], false, false);
        const auto c = __builtin_amdgcn_permlane32_swap(P[2 * pr][1], P[2 * pr + 1][1], false, false);
        const u32x4 v = {a[0], c[0], a[1], c[1]};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, st_off + 32u * pr, so, 0);
        if (mask_out) mword |= (relu_byte_of_pairs(v) & 0xffu) << (16 * pr);
      }
      if (mask_out) {      // (wave-uniform)
        mword |= (uint32_t)__shfl_xor((int)mword, 32, 64) << 8;
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(mk_sample, 0, (int)((size_t)t_out * mk_plane_b), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(mword, mrs, mk_off, (uint32_t)t * mk_plane_b + (uint32_t)orow * mask_wp * 4u, 0);
      }
    }
#ifdef PV_DIAG_STAMPS
    PV_STAMP(q3);
    dg[0] += q1 - q0;   // barrier (waiting for the loaders)
    dg[1] += q2 - q1;   // 9 groups of MFMAs
    dg[2] += q3 - q2;   // epilogue
    dg[3] += 1;
#endif
  }
#ifdef PV_DIAG_STAMPS
  {
    const int wgl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (lane == 0 && wgl * 4 + wave < PV_DIAG_WAVES)
      for (int i = 0; i < PV_DIAG_SLOTS; ++i) first_diag[(size_t)(wgl * 4 + wave) * PV_DIAG_SLOTS + i] = dg[i];
  }
#endif
}

// 0 = launched, 1 = not covered (the caller uses conv3d_bf16.hip's one-role kernel)
int launch_conv3d_first_f32in(const float* x, uint16_t* xp_out, const uint16_t* wp, const float* bias, uint16_t* y,
                              const pv_conv3d_dims* d, int to, int ho, int wo, int relu, hipStream_t st, uint32_t* mask_out) {
  if (d->pad_w != 0 || d->w_in % 4 != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 || ((uintptr_t)xp_out % 16) != 0)
    return 1;
  if (to < 1) return 1;
  const int n_rowblk = (ho + F_TR - 1) / F_TR;
  const int n_colblk = (wo + F_TW_VALID - 1) / F_TW_VALID;
  const long long tiles = (long long)d->batch * n_rowblk * n_colblk;
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
  hipLaunchKernelGGL(conv3d_first_f32in_kernel, grid, dim3(512), 0, st, x, wp, bias, y, xp_out, d->c_in, d->t_in, d->h_in,
                     d->w_in, to, ho, wo, d->pad_t, d->pad_h, relu ? 1 : 0, n_colblk, t_chunk, d->c_out, mask_out, (ho + 7) & ~7,
                     (wo + 31) & ~31);
  return 0;
}

}  // namespace pv

#ifdef PV_DIAG_STAMPS
extern "C" int pv_diag_read_first(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::first_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
