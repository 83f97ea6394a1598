// Forward and data gradient of the f32 model's 32 -> 32 channel 3x3x3 layers on the 16-bit matrix cores at f32 accuracy
// (round 5).  Reference op: nn.Conv3d in float32, predict_pv_yield/models/conv3d/model.py:80-90,113-120.
//
// The f32 matrix instruction runs at the vector-ALU rate (conv3d_general_f32.hip: 0.8-1.1 ms per layer pass at B = 32).  Here
// both operands are split in two half-float terms, x s_x = x_h + x_l and w s_w = w_h + w_l (22 significant bits each, s a power of
// two that brings the tensor's largest magnitude below 2^14; subnormal residuals are honoured by the f16 matrix instruction,
// tools/probes/mfma_f16_denorm.hip), and the three products above 2^-22 of the result,
//     x_l w_h,  x_h w_l,  x_h w_h,
// each run as ONE launch of the bf16 model's input-stationary kernel in its half-float form (conv3d_bf16_v3.hip, F32OUT: the
// f32 accumulators leave as they are, NDHWC).  sum3_ndhwc_to_ncdhw_kernel adds the three partial tensors smallest first,
// un-scales (exact: powers of two), adds the bias, applies the ReLU (forward) or the consumer's ReLU gate (dgrad), transposes
// to the f32 model's NCDHW tensors and leaves the largest magnitude for the NEXT split (no pass of its own for the maximum).
// The x_l w_l term is dropped (2^-22 of the result, below the two-term split's own error); accumulation is f32 inside the
// matrix instruction as in the f32 kernels.  The weight gradient of these layers has run in the same way since round 4
// (pv_conv3d_bwd_weight_f16); it now takes the operand images the forward made instead of splitting x again.
#include "pv_common.h"

namespace pv {

int launch_conv3d_fwd_f16_f32out_v3(const uint16_t* x, const uint16_t* wp2, void* y, int f16_out, const pv_conv3d_dims* d, int to,
                                    int ho, int wo, hipStream_t st);

bool v3_f32out_covers(const pv_conv3d_dims* d, int to, int ho, int wo);

constexpr int F2_WFRAG = 27 * 2 * 64 * 8;      // one operator's 16x16x32 A fragments (conv3d_bf16_v3.hip)

__device__ __forceinline__ float f2_scale_of(uint32_t maxbits) {      // = split2_scale of conv3d_bf16.hip
  if (maxbits == 0u || maxbits >= 0x7f800000u) return 1.f;
  int e = __builtin_amdgcn_frexp_expf(__builtin_bit_cast(float, maxbits));      // max = f 2^e, 0.5 <= f < 1, e <= 128
  e = e < -112 ? -112 : e;      // (s and 1 / s stay normal floats: 2^-114 .. 2^126; a tensor below 2^-112 is scaled by 2^126)
  return ldexpf(1.f, 14 - e);
}

// w [32,32,27] f32 -> four fragment images [27][2 cout halves][64 lanes][8] half floats: forward operator (h, l), dgrad operator
// (transposed and flipped: h, l); state = (bits of max |w|, s, 1 / s, -, -, 32 absolute row sums of the forward operator, 32 of
// the data-gradient operator).  Every workgroup finds the maximum of the 27 648 weights
// for itself (110 KB from L2; one workgroup doing everything took 34 us) and packs its share of the fragments.
__global__ __launch_bounds__(1024) void pack_weight_v3_split2_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp,
                                                                      float* __restrict__ state, int c_out, int c_in) {
  __shared__ uint32_t red[16];
  const int n = c_out * c_in * 27;
  uint32_t m = 0;
  for (int i = threadIdx.x; i < n; i += 1024) m = max(m, *reinterpret_cast<const uint32_t*>(w + i) & 0x7fffffffu);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = max(m, red[i]);
  const float s = f2_scale_of(m);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    state[0] = __builtin_bit_cast(float, m);
    state[1] = s;
    state[2] = 1.f / s;
  }
  // state[5 + r], state[37 + r] (r < 32): the absolute row sums of the forward / the data-gradient operator (sum over what an
  // output element contracts): |y| <= max |x| max_r state[5 + r] + max |bias| bounds an output before it exists -- the scale of
  // ITS split (sum pass below, which takes the maximum of the 32 sums itself).  A wave per row, lanes over the row's elements in
  // a fixed order + a butterfly: the same bits every time; the 64 rows are dealt out over the first 4 workgroups (one workgroup
  // doing all of them took 60 us one thread per row, 18 us one wave per four rows).
  {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (row < 64) {
      const int r = row & 31, back = row >> 5;
      float acc = 0.f;
      if (!back) {
        if (r < c_out) for (int i = lane; i < c_in * 27; i += 64) acc += fabsf(w[(size_t)r * c_in * 27 + i]);
      } else {
        if (r < c_in) for (int i = lane; i < c_out * 27; i += 64) acc += fabsf(w[((size_t)(i / 27) * c_in + r) * 27 + i % 27]);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
      if (lane == 0) state[5 + row] = acc;
    }
  }
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < 2 * F2_WFRAG; i += gridDim.x * 1024) {
    const int flip = i >= F2_WFRAG, e = flip ? i - F2_WFRAG : i;
    const int j = e & 7, lane = (e >> 3) & 63, half = (e >> 9) & 1, tap = e >> 10;
    const int row = 16 * half + (lane & 15);  // output channel of this operator
    const int k = 8 * (lane >> 4) + j;        // input channel of this operator
    float v = 0.f;
    if (!flip) {
      if (row < c_out && k < c_in) v = w[((size_t)row * c_in + k) * 27 + tap];
    } else {
      if (row < c_in && k < c_out) v = w[((size_t)k * c_in + row) * 27 + (26 - tap)];
    }
    const float f = v * s;
    const _Float16 h = (_Float16)f;
    const _Float16 l = (_Float16)(f - (float)h);
    wp[(size_t)(2 * flip) * F2_WFRAG + e] = __builtin_bit_cast(uint16_t, h);
    wp[(size_t)(2 * flip + 1) * F2_WFRAG + e] = __builtin_bit_cast(uint16_t, l);
  }
}

// parts [3][B][vox][32] f32 (x_l w_h, x_h w_l, x_h w_h) -> y [B][c = 32][vox] f32:
//   y = ((p0 + p1) + p2) / (s_x s_w) + bias,  then ReLU (relu) or zero where gate <= 0 (gate: NCDHW like y);
// the largest |y| goes to max_state (bits, atomicMax; the caller zeroes it) for the split of y.
// A wave owns tiles of 32 voxels x 32 channels: 16-byte reads along the channels, an LDS transpose, 16-byte writes along the voxels.
// P01_F16: the two small partial products are half-float images scaled by 2^-12 (conv3d_bf16_v3.hip OUTM 2): parts = p2 alone, p01 =
// [2][B][vox][32] half floats
// PLANES: the pass also writes y's OWN two-term split (out_h, out_l: NDHWC half-float images, the layout the partial products
// came in) for the next layer's launches, so that no split pass reads y again.  Its scale must be known before the first
// element: s_y = the power of two that brings a BOUND of |y| below 2^14 -- max |x| (sx[0]) times the operator's largest absolute
// row sum (*l1) plus max |bias| -- an upper bound by the triangle inequality, ~2^7 above the largest |y| on these layers; half
// floats have the range to spare (absolute error of an element <= 2^-32 of the largest).  out_state = (bits of max |y| by
// atomicMax -- zeroed by the caller --, s_y, 1 / s_y).  gate_h: the gating activation's h image (NDHWC): y where it is > 0.
template <bool P01_F16, bool PLANES>
__global__ __launch_bounds__(256) void sum3_ndhwc_to_ncdhw_kernel(const float* __restrict__ parts, size_t part_stride,
                                                                   const uint16_t* __restrict__ p01,
                                                                   const float* __restrict__ sx, const float* __restrict__ sw,
                                                                   const float* __restrict__ l1,
                                                                   const float* __restrict__ bias, const uint16_t* __restrict__ gate_h,
                                                                   float* __restrict__ y, uint16_t* __restrict__ out_h,
                                                                   uint16_t* __restrict__ out_l, uint32_t* __restrict__ max_state, int relu,
                                                                   long long vps, long long tiles_per_sample, long long total_tiles) {
  __shared__ float tile[4][32 * 33];
  __shared__ uint32_t wave_max[4];
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* t = tile[wave];
  const float unscale = sx[2] * sw[2];
  const int rq = lane & 7, rv = lane >> 3;      // read: 4 channels 4 rq .. of voxel 8 i + rv;  write: 4 voxels 4 rq .. of channel 8 i + rv
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + 4 * rq);
  float s_y = 1.f;
  if constexpr (PLANES) {
    float bmax = 0.f;
    if (bias) for (int i = 0; i < 32; ++i) bmax = fmaxf(bmax, fabsf(bias[i]));
    float l1max = 0.f;
    for (int i = 0; i < 32; ++i) l1max = fmaxf(l1max, l1[i]);
    const float bound = __builtin_bit_cast(float, reinterpret_cast<const uint32_t*>(sx)[0]) * l1max * 1.01f + bmax;
    s_y = f2_scale_of(__builtin_bit_cast(uint32_t, bound));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      reinterpret_cast<float*>(max_state)[1] = s_y;
      reinterpret_cast<float*>(max_state)[2] = 1.f / s_y;
    }
  }
  uint32_t m = 0;
  for (long long tl = (long long)blockIdx.x * 4 + wave; tl < total_tiles; tl += (long long)gridDim.x * 4) {
    const long long bi = tl / tiles_per_sample;
    const long long v0 = (tl - bi * tiles_per_sample) * 32;
    f32x4 a[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long v = v0 + 8 * i + rv;
      const bool ok = v < vps;
      const size_t off = ((size_t)(bi * vps + (ok ? v : 0)) * 32) + 4 * rq;
      if constexpr (P01_F16) {
        a[i][2] = ok ? *reinterpret_cast<const f32x4*>(parts + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          f16x4_t hv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          if (ok) hv = *reinterpret_cast<const f16x4_t*>(p01 + p * part_stride + off);
          a[i][p] = __builtin_convertvector(hv, f32x4) * 4096.f;
        }
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[i][p] = ok ? *reinterpret_cast<const f32x4*>(parts + p * part_stride + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long v = v0 + 8 * i + rv;
      const bool ok = v < vps;
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(__fmul_rn(__fadd_rn(__fadd_rn(a[i][0][j], a[i][1][j]), a[i][2][j]), unscale), b4[j]);
      if (relu) {      // (as torch: a NaN stays a NaN)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = o[j] < 0.f ? 0.f : o[j];
      }
      const size_t voff = ((size_t)(bi * vps + (ok ? v : 0)) * 32) + 4 * rq;
      if (gate_h) {
        f16x4_t gv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (ok) gv = *reinterpret_cast<const f16x4_t*>(gate_h + voff);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gv[j] > (_Float16)0.f ? o[j] : 0.f;
      }
      if (ok) {
        const u32x4 ob = __builtin_bit_cast(u32x4, o);
        m = max(max(m, ob[0] & 0x7fffffffu), max(max(ob[1] & 0x7fffffffu, ob[2] & 0x7fffffffu), ob[3] & 0x7fffffffu));
        if constexpr (PLANES) {
          const f32x4 f = o * s_y;
          const f16x4_t hh = __builtin_convertvector(f, f16x4_t);
          const f16x4_t ll = __builtin_convertvector(f - __builtin_convertvector(hh, f32x4), f16x4_t);
          *reinterpret_cast<f16x4_t*>(out_h + voff) = hh;
          *reinterpret_cast<f16x4_t*>(out_l + voff) = ll;
        }
      }
      if (y) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[(8 * i + rv) * 33 + 4 * rq + j] = o[j];
      }
    }
    if (!y) continue;      // (the consumer reads the operand images: no f32 tensor, no transpose)
    __builtin_amdgcn_wave_barrier();      // (a wave's LDS operations execute in order: its own writes are visible to its reads)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = 8 * i + rv;
      const long long v = v0 + 4 * rq;
      if (v < vps) {      // (vps % 4 == 0: whole quads)
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = t[(4 * rq + j) * 33 + ch];
        const size_t off = (size_t)(bi * 32 + ch) * vps + v;
        *reinterpret_cast<f32x4*>(y + off) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (max_state) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) wave_max[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
      if (m) atomicMax(max_state, m);
    }
  }
}

}  // namespace pv

using namespace pv;

extern "C" {

size_t pv_conv3d_split2_weight_elems(void) { return (size_t)4 * F2_WFRAG; }

int pv_conv3d_pack_weight_split2_f16(const float* w, uint16_t* wp, float* state, int32_t c_out, int32_t c_in, void* stream) {
  PV_REQUIRE(w && wp && state, PV_EINVAL, "pv_conv3d_pack_weight_split2_f16: null pointer");
  PV_REQUIRE(c_out > 0 && c_out <= 32 && c_in > 0 && c_in <= 32, PV_ESIZE,
             "pv_conv3d_pack_weight_split2_f16: channels (%d,%d) must be in 1..32", c_out, c_in);
  hipLaunchKernelGGL(pack_weight_v3_split2_kernel, dim3(27), dim3(1024), 0, as_stream(stream), w, wp, state, c_out, c_in);
  return check_launch("pv_conv3d_pack_weight_split2_f16");
}

int pv_conv3d_fwd_f16_f32out_covers(const pv_conv3d_dims* d) {
  if (!d || d->batch <= 0 || d->batch > 65535 || d->c_in <= 16 || d->c_in > 32 || d->c_out <= 0 || d->c_out > 32) return 0;
  if (d->pad_t < 0 || d->pad_t > 2 || d->pad_h < 0 || d->pad_h > 2 || d->pad_w < 0 || d->pad_w > 2) return 0;
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  if (to <= 0 || ho <= 0 || wo <= 0 || (size_t)d->t_in * d->h_in * d->w_in * 64 > 0x40000000ull) return 0;
  return v3_f32out_covers(d, to, ho, wo) ? 1 : 0;
}

int pv_conv3d_fwd_f16_f32out(const uint16_t* x, const uint16_t* wp, void* y, int32_t y_is_f16, const pv_conv3d_dims* d, void* stream) {
  PV_REQUIRE(d && x && wp && y, PV_EINVAL, "pv_conv3d_fwd_f16_f32out: null pointer");
  PV_REQUIRE(d->batch > 0 && d->batch <= 65535 && d->c_in > 16 && d->c_in <= 32 && d->c_out > 0 && d->c_out <= 32, PV_ESIZE,
             "pv_conv3d_fwd_f16_f32out: channels (%d -> %d): 17..32 in (64-byte voxels), 1..32 out", d->c_in, d->c_out);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_t <= 2 && d->pad_h >= 0 && d->pad_h <= 2 && d->pad_w >= 0 && d->pad_w <= 2, PV_EINVAL,
             "pv_conv3d_fwd_f16_f32out: padding must be 0..2");
  const int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  PV_REQUIRE(to > 0 && ho > 0 && wo > 0, PV_ESIZE, "pv_conv3d_fwd_f16_f32out: input smaller than the kernel");
  PV_REQUIRE((size_t)d->t_in * d->h_in * d->w_in * 64 <= 0x40000000ull, PV_ESIZE,
             "pv_conv3d_fwd_f16_f32out: one sample exceeds 1 GiB (buffer-addressing limit of this kernel)");
  const int rc = launch_conv3d_fwd_f16_f32out_v3(x, wp, y, y_is_f16 ? 1 : 0, d, to, ho, wo, as_stream(stream));
  PV_REQUIRE(rc != 1, PV_ESIZE, "pv_conv3d_fwd_f16_f32out: shape not covered (fewer than two output slices per time chunk, a sample of "
                                "y beyond 2 GiB, or unaligned tensors); the caller keeps pv_conv3d_general_fwd_f32");
  return rc;
}

int pv_sum3_ndhwc_to_ncdhw_f32(const float* parts, const uint16_t* p01_f16, const float* sx_state, const float* sw_state,
                               int32_t data_gradient, const float* bias, const uint16_t* gate_h, float* y, uint16_t* out_h,
                               uint16_t* out_l, float* max_state, int32_t relu, int32_t batch, int64_t vox_per_sample, void* stream) {
  PV_REQUIRE(parts && sx_state && sw_state && (y || out_h), PV_EINVAL, "pv_sum3_ndhwc_to_ncdhw_f32: null pointer");
  PV_REQUIRE((out_h != nullptr) == (out_l != nullptr) && (!out_h || max_state), PV_EINVAL,
             "pv_sum3_ndhwc_to_ncdhw_f32: out_h, out_l and max_state come together");
  PV_REQUIRE(batch > 0 && vox_per_sample > 0 && vox_per_sample % 4 == 0, PV_ESIZE,
             "pv_sum3_ndhwc_to_ncdhw_f32: voxels per sample must be a positive multiple of 4");
  PV_REQUIRE((((uintptr_t)parts | (uintptr_t)y | (uintptr_t)gate_h | (uintptr_t)bias | (uintptr_t)p01_f16 | (uintptr_t)out_h |
               (uintptr_t)out_l) & 15) == 0, PV_EINVAL,
             "pv_sum3_ndhwc_to_ncdhw_f32: 16-byte aligned tensors");
  const long long tps = (vox_per_sample + 31) / 32, total = tps * batch;
  const unsigned grid = (unsigned)std::min<long long>((total + 3) / 4, (long long)kNumCU * 8);
  const float* l1 = sw_state + (data_gradient ? 37 : 5);      // the operator's 32 absolute row sums
#define PV_LAUNCH_SUM3(F16P, PL)                                                                                              \
  hipLaunchKernelGGL((sum3_ndhwc_to_ncdhw_kernel<F16P, PL>), dim3(grid), dim3(256), 0, as_stream(stream), parts,               \
                     (size_t)batch * vox_per_sample * 32, p01_f16, sx_state, sw_state, l1, bias, gate_h, y, out_h, out_l,     \
                     reinterpret_cast<uint32_t*>(max_state), relu ? 1 : 0, (long long)vox_per_sample, tps, total)
  if (p01_f16) {
    if (out_h) PV_LAUNCH_SUM3(true, true); else PV_LAUNCH_SUM3(true, false);
  } else {
    if (out_h) PV_LAUNCH_SUM3(false, true); else PV_LAUNCH_SUM3(false, false);
  }
#undef PV_LAUNCH_SUM3
  return check_launch("pv_sum3_ndhwc_to_ncdhw_f32");
}

}  // extern "C"
