// bf16 MFMA Conv3D, two-waves-per-SIMD variant for 32 -> 32 channel layers (forward L1..L2 and every dgrad).
//
// conv3d_bf16.hip keeps all 54 weight fragments of a 32x32-output-channel tile in one wave's registers, which
// forces one wave per SIMD: every barrier, LDS round trip and store of that wave idles the matrix pipe
// (PMC: matrix pipe busy 52 % of wave cycles).  Here the OUTPUT CHANNELS are split over a pair of waves instead:
// each wave owns 16 output channels x (4 rows x 32 columns) and uses v_mfma_f32_16x16x32_bf16, whose A operand
// (16 couts x 32 cin) is 4 VGPRs per tap -> 27 fragments = 108 VGPRs, so a wave fits in 256 registers and TWO
// waves share each SIMD: while one sits at a barrier / waits for LDS / stores its tile, the other feeds the
// matrix pipe.  No cross-wave reduction is needed (the split is over M, not K).
//
//   workgroup = 8 waves = (row quad wr) x (column segment ws) x (cout half ch); tile 8 rows x 64 columns, marching
//   over time with the same 3-slice LDS ring, raw-buffer staging, bias-initialised accumulators and LDS-staged
//   coalesced NDHWC write-out as the v1 kernel.  The B operand (activations, 32 cin x 16 voxels) is ONE
//   ds_read_b128 per fragment (lane = voxel x 16-byte channel chunk); the XOR swizzle chunk ^= ((voxel>>2)&1)<<1
//   makes that read conflict-free for every tap shift (brute-forced over all 16 alignments and the 4 lane groups).
#include "pv_common.h"

namespace pv {

constexpr int V2_TR = 8, V2_TRI = 10, V2_TW = 64, V2_TW_VALID = 62;
constexpr int V2_VOXB = 64, V2_ROWB = V2_TW * V2_VOXB, V2_SLOTB = V2_TRI * V2_ROWB;

__device__ __forceinline__ uint32_t v2_gate_word(uint32_t x, uint32_t g) {
  uint32_t lo = ((g & 0x7fffu) != 0u && (g & 0x8000u) == 0u) ? 0x0000ffffu : 0u;
  uint32_t hi = ((g & 0x7fff0000u) != 0u && (g & 0x80000000u) == 0u) ? 0xffff0000u : 0u;
  return x & (lo | hi);
}

template <bool OUT_GATE>
__global__ __launch_bounds__(512, 2) void conv3d_fwd_bf16_v2_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ wp2, const float* __restrict__ bias,
    uint16_t* __restrict__ y, const uint16_t* __restrict__ out_gate, int t_in, int h_in, int w_in, int t_out,
    int h_out, int w_out, int pad_t, int pad_h, int pad_w, int relu, int n_colblk, int t_chunk, int c_out) {
  // ring of 3 slices | 256 B zeros | 32 bias floats (+pad) | 4 pair tiles x 4 rows x 2 KB epilogue staging
  __shared__ __attribute__((aligned(256))) unsigned char lds[3 * V2_SLOTB + 512 + 32768];
  float* lds_bias = reinterpret_cast<float*>(lds + 3 * V2_SLOTB + 256);
  unsigned char* lds_epi = lds + 3 * V2_SLOTB + 512;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = wave & 1;         // output-channel half (16 couts)
  const int ws = (wave >> 1) & 1;  // column segment (32 columns)
  const int wr = wave >> 2;        // row quad
  const int pair = wave >> 1;      // the two waves of a pair share the same 4 x 32 voxels
  const int vox = lane & 15, kg = lane >> 4;

  const int rowblk = blockIdx.x / n_colblk;
  const int colblk = blockIdx.x - rowblk * n_colblk;
  const int h0 = rowblk * V2_TR;
  const int w0 = colblk * V2_TW_VALID;
  const int b = blockIdx.z;
  const int tc0 = blockIdx.y * t_chunk;
  const int tc1 = min(tc0 + t_chunk, t_out);
  if (tc0 >= tc1) return;

  if (tid < 64) reinterpret_cast<uint32_t*>(lds + 3 * V2_SLOTB)[tid] = 0u;
  if (tid < 32) lds_bias[tid] = (bias && tid < c_out) ? bias[tid] : 0.f;

  // ---- weights: 27 A fragments (16 couts x 32 cin) of this wave's cout half, resident in registers ----------
  bf16x8 wfrag[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
    wfrag[tap] = *reinterpret_cast<const bf16x8*>(wp2 + ((size_t)(tap * 2 + ch) * 64 + lane) * 8);

  // ---- per-lane LDS read offsets of the B operand: voxel 32*ws + 16*half + vox + kw, 16-byte chunk kg ---------
  int voff[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int v = 32 * ws + 16 * half + vox + kw;
      voff[kw][half] = v * V2_VOXB + ((kg ^ (((v >> 2) & 1) << 1)) << 4);
    }

  // ---- staging (512 threads: 5 chunks of 16 B per thread and slice; rows wave-uniform, raw buffer loads) -----
  constexpr uint32_t INVALID = 0x40000000u;
  constexpr int NLOAD = V2_TRI * V2_TW * 4 / 512;  // 5
  const int srow0 = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int srem = tid & 255;
  const int scol = srem >> 2, sc = srem & 3;
  const int swi = w0 - pad_w + scol;
  const bool scol_ok = (unsigned)swi < (unsigned)w_in;
  const int lds_lane = scol * V2_VOXB + ((sc ^ (((scol >> 2) & 1) << 1)) << 4);
  const uint32_t x_plane_b = (uint32_t)h_in * w_in * 64u, x_row_b = (uint32_t)w_in * 64u;
  const size_t sample_elems = (size_t)t_in * h_in * w_in * 32;
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * sample_elems), 0, (int)(sample_elems * 2), 0x00020000);
  const uint32_t lane_voff = scol_ok ? (uint32_t)(swi * 32 + sc * 8) * 2u : INVALID;

  u32x4 stage[NLOAD];
  auto load_slice = [&](int s) {
    const int ti = s - pad_t;
    const bool t_ok = (unsigned)ti < (unsigned)t_in;
    const uint32_t toff = (uint32_t)min(max(ti, 0), t_in - 1) * x_plane_b;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int hi = h0 - pad_h + 2 * i + srow0;
      const bool row_ok = t_ok && (unsigned)hi < (unsigned)h_in;
      const uint32_t srow = toff + (uint32_t)min(max(hi, 0), h_in - 1) * x_row_b + (row_ok ? 0u : INVALID);
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_voff + srow, 0, 0);
    }
  };
  auto store_slice = [&](int s) {
    unsigned char* slot = lds + (s % 3) * V2_SLOTB + srow0 * V2_ROWB + lds_lane;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) *reinterpret_cast<u32x4*>(slot + 2 * i * V2_ROWB) = stage[i];
  };

  const int plane_out = h_out * w_out;
  // ---- write-out geometry: the pair's 4 x (32 voxels x 64 B) tile is written by its two waves, 2 rows each;
  // lane -> (voxel 16*half + lane/4, 16-byte chunk lane%4): one wave-instruction = 1 KB of contiguous NDHWC memory
  uint32_t wr_off[2];
  bool wr_ok[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int orow = 2 * ch + k;
    const int ho = h0 + 4 * wr + orow;
    const int wo0 = w0 + 32 * ws + (lane >> 2);
    wr_off[k] = ((uint32_t)(ho * w_out + wo0) * 32u + 8u * (lane & 3)) * 2u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col_t = 32 * ws + 16 * half + (lane >> 2);
      wr_ok[k][half] = ho < h_out && col_t < V2_TW_VALID && (w0 + col_t) < w_out;
    }
  }
  const __amdgpu_buffer_rsrc_t ogrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((OUT_GATE ? out_gate : y) + (size_t)b * t_out * plane_out * 32), 0, (int)((size_t)t_out * plane_out * 64),
      0x00020000);
  unsigned char* epi_pair = lds_epi + pair * 8192;
  auto write_out = [&](int tw) {
    unsigned char* yt = reinterpret_cast<unsigned char*>(y + ((size_t)b * t_out + tw) * plane_out * 32);
    u32x4 og[OUT_GATE ? 2 : 1][2];
    if constexpr (OUT_GATE) {
      const uint32_t tbase = (uint32_t)tw * (uint32_t)plane_out * 64u;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int half = 0; half < 2; ++half)
          og[k][half] = __builtin_amdgcn_raw_buffer_load_b128(
              ogrsrc, wr_ok[k][half] ? tbase + wr_off[k] + half * 1024 : INVALID, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int orow = 2 * ch + k;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int v = 16 * half + (lane >> 2);
        const int c = lane & 3;
        const int pc = c ^ ((v >> 1) & 3);
        u32x4 val = *reinterpret_cast<const u32x4*>(epi_pair + orow * 2048 + v * 64 + pc * 16);
        if (wr_ok[k][half]) {
          if constexpr (OUT_GATE) {
            const u32x4 g = og[k][half];
            val[0] = v2_gate_word(val[0], g[0]); val[1] = v2_gate_word(val[1], g[1]);
            val[2] = v2_gate_word(val[2], g[2]); val[3] = v2_gate_word(val[3], g[3]);
          }
          *reinterpret_cast<u32x4*>(yt + wr_off[k] + half * 1024) = val;
        }
      }
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  load_slice(tc0);
  store_slice(tc0);
  load_slice(tc0 + 1);
  store_slice(tc0 + 1);
  load_slice(tc0 + 2);

  for (int t = tc0; t < tc1; ++t) {
    store_slice(t + 2);
    __syncthreads();  // slice t+2 visible; the partner's staged tile of slice t-1 visible
    if (t + 1 < tc1) load_slice(t + 3);
    if (t > tc0) write_out(t - 1);  // previous slice's tile: stores ride under this slice's MFMAs

    // accumulators: [row][half] 16 couts x 16 voxels, initialised with the bias of couts 16*ch + 4*kg + reg
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_bias + 16 * ch + 4 * kg);
    f32x4 acc[4][2];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
      for (int half = 0; half < 2; ++half) acc[r4][half] = b4;

#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const unsigned char* slot = lds + ((t + kt) % 3) * V2_SLOTB + (4 * wr) * V2_ROWB;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        bf16x8 bfr[6][2];
#pragma unroll
        for (int ir = 0; ir < 6; ++ir)
#pragma unroll
          for (int half = 0; half < 2; ++half)
            bfr[ir][half] = *reinterpret_cast<const bf16x8*>(slot + ir * V2_ROWB + voff[kw][half]);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const bf16x8 a = wfrag[kt * 9 + kh * 3 + kw];
#pragma unroll
          for (int orow = 0; orow < 4; ++orow)
#pragma unroll
            for (int half = 0; half < 2; ++half)
              acc[orow][half] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[orow + kh][half], acc[orow][half], 0, 0, 0);
        }
      }
    }

    // ---- stage the tile: lane holds 4 consecutive couts (8 B) of voxel (16*half + vox) --------------------------
    __syncthreads();  // every wave finished reading slot t%3 AND the previous write_out finished reading lds_epi
#pragma unroll
    for (int orow = 0; orow < 4; ++orow)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 a = acc[orow][half];
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = a[j] > 0.f ? a[j] : 0.f;
        }
        u32x2 o;
        o[0] = pack_bf16_pair(a[0], a[1]);
        o[1] = pack_bf16_pair(a[2], a[3]);
        const int v = 16 * half + vox;
        const int slot8 = (4 * ch + kg) ^ (((v >> 1) & 3) << 1);  // 8-byte slot = couts (16ch + 4kg ..+3), even-XOR swizzle
        *reinterpret_cast<u32x2*>(epi_pair + orow * 2048 + v * 64 + slot8 * 8) = o;
      }
  }
  __syncthreads();
  write_out(tc1 - 1);
}

// w[Co,Ci,27] f32 -> 16x16x32 A fragments [27][2 cout halves][64 lanes][8]:
//   lane (co = lane&15, kg = lane>>4), element j  <-  W[cout = 16*half + co][cin = 8*kg + j][tap]
__global__ __launch_bounds__(256) void pack_weight_v2_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp2,
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

int launch_conv3d_fwd_bf16_v2(const uint16_t* x, const uint16_t* wp2, const float* bias, uint16_t* y,
                              const uint16_t* out_gate, const pv_conv3d_dims* d, int to, int ho, int wo, int relu,
                              int n_rowblk, int n_colblk, int n_tchunk, int t_chunk, hipStream_t st) {
  dim3 grid((unsigned)(n_rowblk * n_colblk), (unsigned)n_tchunk, (unsigned)d->batch);
  if (out_gate)
    hipLaunchKernelGGL(conv3d_fwd_bf16_v2_kernel<true>, grid, dim3(512), 0, st, x, wp2, bias, y, out_gate, d->t_in, d->h_in,
                       d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, relu ? 1 : 0, n_colblk, t_chunk, d->c_out);
  else
    hipLaunchKernelGGL(conv3d_fwd_bf16_v2_kernel<false>, grid, dim3(512), 0, st, x, wp2, bias, y, out_gate, d->t_in, d->h_in,
                       d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w, relu ? 1 : 0, n_colblk, t_chunk, d->c_out);
  return check_launch("pv_conv3d_fwd_bf16(v2)");
}

void launch_pack_weight_v2(const float* w, uint16_t* wp2, int c_out, int c_in, int transpose_flip, hipStream_t st) {
  hipLaunchKernelGGL(pack_weight_v2_kernel, dim3(54), dim3(256), 0, st, w, wp2, c_out, c_in, transpose_flip ? 1 : 0);
}

}  // namespace pv
