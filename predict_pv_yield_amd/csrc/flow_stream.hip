// HBM-bound streaming stages of the optical-flow advection path:
// u8 conversion, weighted mean of flow fields, per-channel normalisation and the cv.remap-exact
// bilinear warp.  All kernels are grid-stride, 16 B per lane where the layout allows it.
#include "pv_common.h"

namespace pv {

static thread_local char g_err[512];
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---------------------------------------------------------------------------------------------
// u8 conversion of 10-bit counts
// ---------------------------------------------------------------------------------------------
// mode 0: np.round(x / 4.0) (half to even), notebooks/13_...ipynb:112-119
// mode 1: ((x - 0) / 1023) * 255 truncated,  notebooks/optical_flow_1.ipynb:129-134 (f32 arithmetic)
__device__ __forceinline__ int u8_from_f32(float x, int mode, bool& bad) {
  float v;
  if (mode == PV_U8_ROUND_DIV4) {
    v = rintf(x * 0.25f);  // x/4.0 is an exact power-of-two scaling; rint = half-to-even
  } else {
    v = truncf(__fmul_rn(__fdiv_rn(x, 1023.0f), 255.0f));
  }
  if (!(v >= 0.0f && v <= 255.0f)) {  // also catches NaN
    bad = true;
    v = v > 255.0f ? 255.0f : 0.0f;
  }
  return (int)v;
}

__device__ __forceinline__ int u8_from_i16(int x, int mode, bool& bad) {
  if (mode == PV_U8_ROUND_DIV4) {
    // exact integer form of round_half_even(x / 4)
    int q = x >> 2, r = x & 3;
    int v = q + (r > 2 ? 1 : (r == 2 ? (q & 1) : 0));
    if (v < 0 || v > 255) {
      bad = true;
      v = v < 0 ? 0 : 255;
    }
    return v;
  }
  return u8_from_f32((float)x, mode, bad);
}

template <typename T>
__global__ __launch_bounds__(256) void u8_from_10bit_kernel(const T* __restrict__ src,
                                                             uint8_t* __restrict__ dst, size_t n,
                                                             int mode, int32_t* range_flag) {
  bool bad = false;
  size_t nvec = n / 8;
  size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid; i < nvec; i += stride) {
    uint32_t lo = 0, hi = 0;
    if constexpr (sizeof(T) == 2) {
      u32x4 raw = *reinterpret_cast<const u32x4*>(src + i * 8);  // 8 x int16 = 16 B
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int x = (int)(int16_t)((raw[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        uint32_t v = (uint32_t)u8_from_i16(x, mode, bad);
        if (j < 4) lo |= v << (8 * j); else hi |= v << (8 * (j - 4));
      }
    } else {
      f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8);
      f32x4 b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo |= (uint32_t)u8_from_f32(a[j], mode, bad) << (8 * j);
        hi |= (uint32_t)u8_from_f32(b[j], mode, bad) << (8 * j);
      }
    }
    u32x2 out = {lo, hi};
    *reinterpret_cast<u32x2*>(dst + i * 8) = out;
  }
  // tail
  for (size_t i = nvec * 8 + tid; i < n; i += stride) {
    if constexpr (sizeof(T) == 2) dst[i] = (uint8_t)u8_from_i16((int)src[i], mode, bad);
    else dst[i] = (uint8_t)u8_from_f32((float)src[i], mode, bad);
  }
  if (range_flag && __any(bad)) {
    if ((threadIdx.x & 63) == 0) atomicOr(range_flag, 1);
  }
}

// ---------------------------------------------------------------------------------------------
// weighted mean over the pair axis, float64 accumulation in index order (np.average semantics)
// ---------------------------------------------------------------------------------------------
struct WMeanWeights {
  double w[64];
  double inv_unused;
  double sum;
};

template <int VEC>
__global__ __launch_bounds__(256) void weighted_mean_kernel(const float* __restrict__ flows,
                                                             float* __restrict__ out,
                                                             int64_t n_groups, int n_per_group,
                                                             int64_t elems, WMeanWeights wts) {
  int64_t per_group = elems / VEC;
  int64_t total = n_groups * per_group;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t g = i / per_group;
    int64_t e = (i - g * per_group) * VEC;
    const float* base = flows + (g * n_per_group) * elems + e;
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
    for (int k = 0; k < n_per_group; ++k) {
      if constexpr (VEC == 4) {
        f32x4 x = *reinterpret_cast<const f32x4*>(base + (int64_t)k * elems);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = __dadd_rn(acc[v], __dmul_rn((double)x[v], wts.w[k]));
      } else {
        acc[0] = __dadd_rn(acc[0], __dmul_rn((double)base[(int64_t)k * elems], wts.w[k]));
      }
    }
    if constexpr (VEC == 4) {
      f32x4 r;
#pragma unroll
      for (int v = 0; v < 4; ++v) r[v] = (float)__ddiv_rn(acc[v], wts.sum);
      *reinterpret_cast<f32x4*>(out + g * elems + e) = r;
    } else {
      out[g * elems + e] = (float)__ddiv_rn(acc[0], wts.sum);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// per-channel normalisation (x - mean[c]) / std[c], true f32 division
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void normalise_kernel(const T* __restrict__ src,
                                                         float* __restrict__ dst, size_t n,
                                                         int64_t inner, int n_channels,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ std_) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int c = (int)((i / (size_t)inner) % (size_t)n_channels);
    float x = (float)src[i];
    dst[i] = __fdiv_rn(__fsub_rn(x, mean[c]), std_[c]);
  }
}

// vectorised variant: inner % 4 == 0 so the 4 lanes of a vector share a channel
template <typename T>
__global__ __launch_bounds__(256) void normalise_kernel_v4(const T* __restrict__ src,
                                                            float* __restrict__ dst, size_t n4,
                                                            int64_t inner4, int n_channels,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ std_) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    int c = (int)((i / (size_t)inner4) % (size_t)n_channels);
    float m = mean[c], s = std_[c];
    f32x4 x;
    if constexpr (sizeof(T) == 2) {
      u32x2 raw = *reinterpret_cast<const u32x2*>(src + i * 4);
      x[0] = (float)(int16_t)(raw[0] & 0xffffu);
      x[1] = (float)(int16_t)(raw[0] >> 16);
      x[2] = (float)(int16_t)(raw[1] & 0xffffu);
      x[3] = (float)(int16_t)(raw[1] >> 16);
    } else {
      x = *reinterpret_cast<const f32x4*>(src + i * 4);
    }
    f32x4 r;
#pragma unroll
    for (int v = 0; v < 4; ++v) r[v] = __fdiv_rn(__fsub_rn(x[v], m), s);
    *reinterpret_cast<f32x4*>(dst + i * 4) = r;
  }
}

// ---------------------------------------------------------------------------------------------
// cv.remap(INTER_LINEAR) restated: 1/32-px fixed-point coordinates, 32x32 bilinear weight table
// (SURVEY.md Appendix A.2).  One thread = VEC consecutive destination pixels of one image row,
// all n_steps extrapolation steps: the flow vector (8 B/px, the dominant compulsory read) is read
// once and reused for every step; the 4 source taps per pixel are gathers served by L1/L2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_x86(float v) {
  // cvRound(float) = cvtss2si: round-half-even; NaN / out of range -> INT_MIN
  if (!(fabsf(v) < 2147483648.0f)) return (int)0x80000000;
  return __float2int_rn(v);
}
__device__ __forceinline__ int sat_short(int v) {
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T>
__device__ __forceinline__ T remap_one(const T* __restrict__ img, int h, int w, float mx, float my,
                                       int border_mode, T border_value) {
  int sx = cv_round_x86(mx * 32.0f);
  int sy = cv_round_x86(my * 32.0f);
  int fxi = sx & 31, fyi = sy & 31;
  int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
  T p00, p01, p10, p11;
  bool inside = (unsigned)ix < (unsigned)(w - 1) && (unsigned)iy < (unsigned)(h - 1);
  if (inside) {
    const T* p = img + (int64_t)iy * w + ix;
    if constexpr (sizeof(T) == 4) {
      // the two taps of a row are adjacent: one 8-byte load each (4-byte aligned is all global_load_dwordx2 needs)
      struct __attribute__((packed, aligned(4))) Pair { T a, b; };
      const Pair r0 = *reinterpret_cast<const Pair*>(p);
      const Pair r1 = *reinterpret_cast<const Pair*>(p + w);
      p00 = r0.a; p01 = r0.b; p10 = r1.a; p11 = r1.b;
    } else {
      p00 = p[0]; p01 = p[1]; p10 = p[w]; p11 = p[w + 1];
    }
  } else if (border_mode == PV_BORDER_REPLICATE) {
    int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
    int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
    p00 = img[(int64_t)y0 * w + x0]; p01 = img[(int64_t)y0 * w + x1];
    p10 = img[(int64_t)y1 * w + x0]; p11 = img[(int64_t)y1 * w + x1];
  } else {
    if (ix >= w || ix + 1 < 0 || iy >= h || iy + 1 < 0) return border_value;
    bool x0in = (unsigned)ix < (unsigned)w, x1in = (unsigned)(ix + 1) < (unsigned)w;
    bool y0in = (unsigned)iy < (unsigned)h, y1in = (unsigned)(iy + 1) < (unsigned)h;
    p00 = (x0in && y0in) ? img[(int64_t)iy * w + ix] : border_value;
    p01 = (x1in && y0in) ? img[(int64_t)iy * w + ix + 1] : border_value;
    p10 = (x0in && y1in) ? img[(int64_t)(iy + 1) * w + ix] : border_value;
    p11 = (x1in && y1in) ? img[(int64_t)(iy + 1) * w + ix + 1] : border_value;
  }
  if constexpr (sizeof(T) == 4) {
    // f32 table: products of (1 - k/32, k/32) -- exact multiples of 1/1024
    float ax = (float)fxi * 0.03125f, ay = (float)fyi * 0.03125f;
    float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay);
    float w10 = (1.0f - ax) * ay, w11 = ax * ay;
    float r = __fmul_rn(p00, w00);
    r = __fadd_rn(r, __fmul_rn(p01, w01));
    r = __fadd_rn(r, __fmul_rn(p10, w10));
    r = __fadd_rn(r, __fmul_rn(p11, w11));
    return r;
  } else {
    // u8: 15-bit fixed point weights, exact
    int w00 = (32 - fxi) * (32 - fyi) * 32, w01 = fxi * (32 - fyi) * 32;
    int w10 = (32 - fxi) * fyi * 32, w11 = fxi * fyi * 32;
    if (w00 == 32768) w00 = 32767;  // saturate_cast<short> of the (0,0) table entry
    int acc = (int)p00 * w00 + (int)p01 * w01 + (int)p10 * w10 + (int)p11 * w11;
    int r = (acc + (1 << 14)) >> 15;
    return (T)(r < 0 ? 0 : (r > 255 ? 255 : r));
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void remap_kernel(const T* __restrict__ src, int64_t src_stride,
                                                     const float* __restrict__ flow, int64_t flow_stride,
                                                     T* __restrict__ dst, int64_t dst_image_stride,
                                                     int64_t dst_step_stride, int64_t n_images,
                                                     int n_steps, float step0, int h, int w,
                                                     int border_mode, T border_value) {
  int wq = w / VEC;
  int64_t per_image = (int64_t)h * wq;
  int64_t total = n_images * per_image;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t img_i = i / per_image;
    int rem = (int)(i - img_i * per_image);
    int y = rem / wq;
    int x0 = (rem - y * wq) * VEC;
    const T* img = src + img_i * src_stride;
    const float* fl = flow + img_i * flow_stride + ((int64_t)y * w + x0) * 2;
    float fxv[VEC], fyv[VEC];
    if constexpr (VEC == 4) {
      f32x4 a = *reinterpret_cast<const f32x4*>(fl);
      f32x4 b = *reinterpret_cast<const f32x4*>(fl + 4);
      fxv[0] = a[0]; fyv[0] = a[1]; fxv[1] = a[2]; fyv[1] = a[3];
      fxv[2] = b[0]; fyv[2] = b[1]; fxv[3] = b[2]; fyv[3] = b[3];
    } else {
      fxv[0] = fl[0]; fyv[0] = fl[1];
    }
    for (int s = 0; s < n_steps; ++s) {
      float k = step0 + (float)s;
      T outv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        // remap = -(flow * k); remap += arange   (f32, no fused multiply-add)
        float mx = __fadd_rn(-__fmul_rn(fxv[v], k), (float)(x0 + v));
        float my = __fadd_rn(-__fmul_rn(fyv[v], k), (float)y);
        outv[v] = remap_one<T>(img, h, w, mx, my, border_mode, border_value);
      }
      T* d = dst + img_i * dst_image_stride + (int64_t)s * dst_step_stride + (int64_t)y * w + x0;
      if constexpr (VEC == 4 && sizeof(T) == 4) {
        f32x4 o = {(float)outv[0], (float)outv[1], (float)outv[2], (float)outv[3]};
        *reinterpret_cast<f32x4*>(d) = o;
      } else if constexpr (VEC == 4 && sizeof(T) == 1) {
        uint32_t o = (uint32_t)outv[0] | ((uint32_t)outv[1] << 8) | ((uint32_t)outv[2] << 16) |
                     ((uint32_t)outv[3] << 24);
        *reinterpret_cast<uint32_t*>(d) = o;
      } else {
        d[0] = outv[0];
      }
    }
  }
}

// Small images (the 64 x 64 PV-site tiles): one workgroup = one image; the source image is staged in LDS once and all
// n_steps x 4 bilinear taps per pixel are LDS gathers instead of L1/L2 round trips.  Arithmetic = remap_one, unchanged.
template <typename T>
__global__ __launch_bounds__(256) void remap_lds_kernel(const T* __restrict__ src, int64_t src_stride,
                                                         const float* __restrict__ flow, int64_t flow_stride,
                                                         T* __restrict__ dst, int64_t dst_image_stride,
                                                         int64_t dst_step_stride, int n_steps, float step0, int h, int w,
                                                         int border_mode, T border_value, int split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T* img = reinterpret_cast<T*>(lds_raw);
  const int64_t img_i = blockIdx.x / split;   // `split` workgroups share an image (each stages all of it: any tap may be needed)
  const int part = blockIdx.x % split;
  const T* g = src + img_i * src_stride;
  const int n_px = h * w;
  constexpr int PER16 = 16 / sizeof(T);
  if (((uintptr_t)g % 16 == 0) && (n_px % PER16 == 0)) {
    for (int i = threadIdx.x; i < n_px / PER16; i += blockDim.x)
      reinterpret_cast<u32x4*>(img)[i] = reinterpret_cast<const u32x4*>(g)[i];
  } else {
    for (int i = threadIdx.x; i < n_px; i += blockDim.x) img[i] = g[i];
  }
  __syncthreads();
  const int wq = w / 4;
  const int q_per = (h * wq + split - 1) / split;
  const int q_end = (part + 1) * q_per < h * wq ? (part + 1) * q_per : h * wq;
  for (int i = part * q_per + threadIdx.x; i < q_end; i += blockDim.x) {
    const int y = i / wq, x0 = (i - y * wq) * 4;
    const float* fl = flow + img_i * flow_stride + ((int64_t)y * w + x0) * 2;
    const f32x4 a = *reinterpret_cast<const f32x4*>(fl);
    const f32x4 b = *reinterpret_cast<const f32x4*>(fl + 4);
    const float fxv[4] = {a[0], a[2], b[0], b[2]}, fyv[4] = {a[1], a[3], b[1], b[3]};
    for (int s = 0; s < n_steps; ++s) {
      const float k = step0 + (float)s;
      T outv[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float mx = __fadd_rn(-__fmul_rn(fxv[v], k), (float)(x0 + v));
        const float my = __fadd_rn(-__fmul_rn(fyv[v], k), (float)y);
        outv[v] = remap_one<T>(img, h, w, mx, my, border_mode, border_value);
      }
      T* d = dst + img_i * dst_image_stride + (int64_t)s * dst_step_stride + (int64_t)y * w + x0;
      if constexpr (sizeof(T) == 4) {
        const f32x4 o = {(float)outv[0], (float)outv[1], (float)outv[2], (float)outv[3]};
        *reinterpret_cast<f32x4*>(d) = o;
      } else {
        *reinterpret_cast<uint32_t*>(d) = (uint32_t)outv[0] | ((uint32_t)outv[1] << 8) | ((uint32_t)outv[2] << 16) |
                                          ((uint32_t)outv[3] << 24);
      }
    }
  }
}

template <typename T>
static int remap_launch(const T* src, int64_t src_stride, const float* flow, int64_t flow_stride,
                        T* dst, int64_t dst_image_stride, int64_t dst_step_stride, int64_t n_images,
                        int32_t n_steps, float step0, int32_t h, int32_t w, int border_mode,
                        T border_value, void* stream) {
  PV_REQUIRE(src && flow && dst, PV_EINVAL, "pv_remap_bilinear: null pointer");
  PV_REQUIRE(n_images >= 0 && n_steps >= 0 && h > 0 && w > 0, PV_EINVAL,
             "pv_remap_bilinear: bad sizes n_images=%lld n_steps=%d h=%d w=%d", (long long)n_images,
             n_steps, h, w);
  PV_REQUIRE(border_mode == PV_BORDER_CONSTANT || border_mode == PV_BORDER_REPLICATE, PV_EINVAL,
             "pv_remap_bilinear: border_mode %d not supported", border_mode);
  PV_REQUIRE(h <= 32767 && w <= 32767, PV_ESIZE, "pv_remap_bilinear: image larger than SHRT_MAX");
  if (n_images == 0 || n_steps == 0) return PV_OK;
  const size_t esz = sizeof(T);
  bool vec_ok = (w % 4 == 0) && (flow_stride % 4 == 0) && ((uintptr_t)flow % 16 == 0) &&
                ((dst_image_stride * esz) % (4 * esz) == 0) && ((dst_step_stride * esz) % (4 * esz) == 0) &&
                ((uintptr_t)dst % (4 * esz) == 0);
  hipStream_t st = as_stream(stream);
  const size_t img_bytes = (size_t)h * w * esz;
  if (vec_ok && img_bytes <= 48 * 1024 && n_images >= 128) {
    const int quads = h * (w / 4);
    int split = quads / 256;   // one pixel quad (x n_steps) per thread
    if (split < 1) split = 1;
    if (split > 8) split = 8;
    hipLaunchKernelGGL((remap_lds_kernel<T>), dim3((unsigned)(n_images * split)), dim3(256), img_bytes, st, src, src_stride,
                       flow, flow_stride, dst, dst_image_stride, dst_step_stride, n_steps, step0, h, w, border_mode,
                       border_value, split);
    return check_launch("pv_remap_bilinear");
  }
  if (vec_ok) {
    size_t work = (size_t)n_images * h * (w / 4);
    hipLaunchKernelGGL((remap_kernel<T, 4>), dim3(stream_grid(work, 256)), dim3(256), 0, st, src,
                       src_stride, flow, flow_stride, dst, dst_image_stride, dst_step_stride, n_images,
                       n_steps, step0, h, w, border_mode, border_value);
  } else {
    size_t work = (size_t)n_images * h * w;
    hipLaunchKernelGGL((remap_kernel<T, 1>), dim3(stream_grid(work, 256)), dim3(256), 0, st, src,
                       src_stride, flow, flow_stride, dst, dst_image_stride, dst_step_stride, n_images,
                       n_steps, step0, h, w, border_mode, border_value);
  }
  return check_launch("pv_remap_bilinear");
}

// ---------------------------------------------------------------------------------------------
// skimage.metrics.structural_similarity(im1, im2) with every option at its default -- the reference's only quality score
// for its optical-flow forecasts (notebooks/optical_flow_1.ipynb cells 31, 35, 38: `metrics.structural_similarity(
// ground_truth_image, remapped_image)`) and the objective of its parameter search (cells 38-42).  Definition (Wang et al.,
// IEEE TIP 2004, as scikit-image 0.18 evaluates it): images as float64, 7 x 7 uniform window, sample covariance
// (N / (N - 1), N = 49), C1 = (0.01 L)^2, C2 = (0.03 L)^2, the mean of the SSIM map over the pixels whose window lies inside
// the image (a border of 3 cropped).  One workgroup per image pair; a thread sums the 49 taps of its pixels in double and
// the map is reduced in a fixed order (lane order, wave order): the score does not depend on the launch.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ssim_mean_kernel(const T* __restrict__ a, const T* __restrict__ b, long long stride_a,
                                                        long long stride_b, int h, int w, double c1, double c2,
                                                        double* __restrict__ out) {
  constexpr int WIN = 7, PAD = 3;
  const T* ia = a + (long long)blockIdx.x * stride_a;
  const T* ib = b + (long long)blockIdx.x * stride_b;
  const int oh = h - 2 * PAD, ow = w - 2 * PAD;
  const double inv_n = 1.0 / (WIN * WIN), cov_norm = (double)(WIN * WIN) / (WIN * WIN - 1);
  double acc = 0.0;
  for (int i = threadIdx.x; i < oh * ow; i += 256) {
    const int y = i / ow, x = i - y * ow;
    double sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
    for (int dy = 0; dy < WIN; ++dy) {
      const T* ra = ia + (long long)(y + dy) * w + x;
      const T* rb = ib + (long long)(y + dy) * w + x;
#pragma unroll
      for (int dx = 0; dx < WIN; ++dx) {
        const double va = (double)ra[dx], vb = (double)rb[dx];
        sa += va, sb += vb, saa += va * va, sbb += vb * vb, sab += va * vb;
      }
    }
    const double ux = sa * inv_n, uy = sb * inv_n;
    const double vx = cov_norm * (saa * inv_n - ux * ux), vy = cov_norm * (sbb * inv_n - uy * uy), vxy = cov_norm * (sab * inv_n - ux * uy);
    acc += ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
  }
  __shared__ double part[256];
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += part[i];
    out[blockIdx.x] = s / ((double)oh * ow);
  }
}

template <typename T>
static int u8_launch(const T* src, uint8_t* dst, size_t n, int mode, int32_t* range_flag, void* stream) {
  if (n == 0) return PV_OK;  // empty input: nothing to do (an empty tensor has no address)
  PV_REQUIRE(src && dst, PV_EINVAL, "pv_u8_from_10bit: null pointer");
  PV_REQUIRE(mode == PV_U8_ROUND_DIV4 || mode == PV_U8_TRUNC_SCALE, PV_EINVAL,
             "pv_u8_from_10bit: bad mode %d", mode);
  PV_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), PV_EINVAL,
             "pv_u8_from_10bit: src must be 16-byte and dst 8-byte aligned");
  if (n == 0) return PV_OK;
  hipLaunchKernelGGL((u8_from_10bit_kernel<T>), dim3(stream_grid((n + 7) / 8, 256)), dim3(256), 0,
                     as_stream(stream), src, dst, n, mode, range_flag);
  return check_launch("pv_u8_from_10bit");
}

// Config-3 prologue in one pass (optical_flow.advect_future_frames): raw counts [B, T, C, H, W] (time-major, as the
// prepared batches hold them) -> (a) u8 frame stacks [B, C, T, H, W] for Farneback (same rounding as
// u8_from_10bit_kernel) and (b) the normalised f32 frames (x - mean_c) / std_c written straight into the model input
// [B, C, T + n_future, H, W] (slices 0..T-1).  Replaces a permute copy, the u8 pass, the normalise pass and a strided
// copy: the raw tensor is read once.  One thread = 8 consecutive pixels of one (b, t, c) frame; frame % 8 == 0.
// VPT vectors (of 8 pixels) per thread and trip, all loaded before the first is converted: with one 16-byte load in flight per
// thread the launch reached 0.51 of HBM on the 64 x 64 tiles (121 MB in 30 us: a CU had ~32 KB in flight, about what the
// memory latency needs at full rate and nothing to spare; profiles/r04/pmc_flow_prepare_stacks.json).
constexpr int PS_VPT = 2;
template <typename T>
__global__ __launch_bounds__(256) void prepare_stacks_kernel(const T* __restrict__ raw, uint8_t* __restrict__ u8,
                                                              float* __restrict__ out, long long n_vec, int t, int c,
                                                              int frame8, int t_out, int mode,
                                                              const float* __restrict__ mean, const float* __restrict__ std_,
                                                              int32_t* range_flag) {
  bool bad = false;
  // a workgroup takes 256 x VPT vectors of ONE frame: (b, t, c) of the frame are wave-uniform (scalar divisions once per trip;
  // the per-thread form -- five 64-bit divisions per 8 pixels -- made this streaming kernel instruction-bound: 0.42 of HBM)
  const int bpf = (frame8 + 256 * PS_VPT - 1) / (256 * PS_VPT);
  const long long n_frames = n_vec / frame8;
  for (long long blk = blockIdx.x; blk < n_frames * bpf; blk += gridDim.x) {
    const long long fr = __builtin_amdgcn_readfirstlane((int)(blk / bpf));     // (frames < 2^31: n_vec / frame8)
    const int px0 = (int)(blk - fr * bpf) * (256 * PS_VPT) + (int)threadIdx.x;
    long long f = fr;
    const int ci = (int)(f % c);
    f /= c;
    const int ti = (int)(f % t);
    const long long bi = f / t;
    const long long bc = bi * c + ci;
    const float m = mean[ci], sd = std_[ci];
    u32x4 r16[PS_VPT];
    f32x4 ra[PS_VPT], rb[PS_VPT];
#pragma unroll
    for (int v = 0; v < PS_VPT; ++v) {
      const int px = px0 + 256 * v;
      if (px < frame8) {
        const long long i = fr * frame8 + px;      // [B][T][C][frame8] (source order)
        if constexpr (sizeof(T) == 2) {
          r16[v] = *reinterpret_cast<const u32x4*>(raw + i * 8);
        } else {
          ra[v] = *reinterpret_cast<const f32x4*>(raw + i * 8);
          rb[v] = *reinterpret_cast<const f32x4*>(raw + i * 8 + 4);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < PS_VPT; ++v) {
      const int px = px0 + 256 * v;
      if (px >= frame8) continue;
      float x[8];
      uint32_t lo = 0, hi = 0;
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int xi = (int)(int16_t)((r16[v][j >> 1] >> ((j & 1) * 16)) & 0xffffu);
          x[j] = (float)xi;
          const uint32_t u = (uint32_t)u8_from_i16(xi, mode, bad);
          if (j < 4) lo |= u << (8 * j); else hi |= u << (8 * (j - 4));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[j] = ra[v][j]; x[j + 4] = rb[v][j];
          lo |= (uint32_t)u8_from_f32(ra[v][j], mode, bad) << (8 * j);
          hi |= (uint32_t)u8_from_f32(rb[v][j], mode, bad) << (8 * j);
        }
      }
      const u32x2 o8 = {lo, hi};
      *reinterpret_cast<u32x2*>(u8 + ((bc * t + ti) * (long long)frame8 + px) * 8) = o8;
      f32x4 r0, r1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r0[j] = __fdiv_rn(__fsub_rn(x[j], m), sd);
        r1[j] = __fdiv_rn(__fsub_rn(x[j + 4], m), sd);
      }
      float* dst = out + ((bc * t_out + ti) * (long long)frame8 + px) * 8;
      *reinterpret_cast<f32x4*>(dst) = r0;
      *reinterpret_cast<f32x4*>(dst + 4) = r1;
    }
  }
  if (range_flag && bad) atomicOr(range_flag, 1);
}

template <typename T>
static int prepare_stacks_launch(const T* raw, uint8_t* u8, float* out, int64_t batch, int32_t t, int32_t c, int64_t frame,
                                 int32_t t_out, int mode, const float* mean, const float* std_, int32_t* range_flag,
                                 void* stream) {
  PV_REQUIRE(raw && u8 && out && mean && std_, PV_EINVAL, "pv_prepare_stacks: null pointer");
  PV_REQUIRE(batch > 0 && t > 0 && c > 0 && frame > 0 && t_out >= t, PV_EINVAL, "pv_prepare_stacks: bad sizes");
  PV_REQUIRE(frame % 8 == 0 && frame / 8 <= 0x7fffffffLL, PV_ESIZE, "pv_prepare_stacks: frame size %lld must be a multiple of 8",
             (long long)frame);
  PV_REQUIRE(mode == PV_U8_ROUND_DIV4 || mode == PV_U8_TRUNC_SCALE, PV_EINVAL, "pv_prepare_stacks: bad mode");
  PV_REQUIRE(((uintptr_t)raw % 16 == 0) && ((uintptr_t)u8 % 8 == 0) && ((uintptr_t)out % 16 == 0), PV_EINVAL,
             "pv_prepare_stacks: unaligned operand");
  const long long n_vec = (long long)batch * t * c * (frame / 8);
  PV_REQUIRE((long long)batch * t * c <= 0x7fffffffLL, PV_ESIZE, "pv_prepare_stacks: too many frames");
  const long long n_blk = (long long)batch * t * c * ((frame / 8 + 256 * PS_VPT - 1) / (256 * PS_VPT));      // 256 x VPT vectors of one frame per workgroup trip
  hipLaunchKernelGGL((prepare_stacks_kernel<T>), dim3((unsigned)std::min<long long>(n_blk, 1 << 20)), dim3(256), 0,
                     as_stream(stream), raw, u8, out, n_vec, t, c, (int)(frame / 8), t_out, mode, mean, std_, range_flag);
  return check_launch("pv_prepare_stacks");
}

template <typename T>
static int normalise_launch(const T* src, float* dst, size_t n, int64_t inner, int32_t n_channels,
                            const float* mean, const float* std_, void* stream) {
  if (n == 0) return PV_OK;
  PV_REQUIRE(src && dst && mean && std_, PV_EINVAL, "pv_normalise: null pointer");
  PV_REQUIRE(inner > 0 && n_channels > 0, PV_EINVAL, "pv_normalise: bad inner/n_channels");
  if (n == 0) return PV_OK;
  hipStream_t st = as_stream(stream);
  bool vec = (inner % 4 == 0) && (n % 4 == 0) && ((uintptr_t)src % (4 * sizeof(T)) == 0) &&
             ((uintptr_t)dst % 16 == 0);
  if (vec) {
    hipLaunchKernelGGL((normalise_kernel_v4<T>), dim3(stream_grid(n / 4, 256)), dim3(256), 0, st, src,
                       dst, n / 4, inner / 4, n_channels, mean, std_);
  } else {
    hipLaunchKernelGGL((normalise_kernel<T>), dim3(stream_grid(n, 256)), dim3(256), 0, st, src, dst,
                       n, inner, n_channels, mean, std_);
  }
  return check_launch("pv_normalise");
}

}  // namespace pv

using namespace pv;

template <typename T>
static int ssim_launch(const T* a, int64_t stride_a, const T* b, int64_t stride_b, int64_t n_pairs, int32_t h, int32_t w,
                       double data_range, double* out, void* stream) {
  PV_REQUIRE(a && b && out, PV_EINVAL, "pv_ssim_mean: null pointer");
  PV_REQUIRE(n_pairs >= 0 && n_pairs <= 0x7fffffffLL, PV_EINVAL, "pv_ssim_mean: bad number of image pairs");
  PV_REQUIRE(h >= 7 && w >= 7, PV_ESIZE, "pv_ssim_mean: images must be at least 7 x 7 (the window), got %d x %d", h, w);
  PV_REQUIRE(data_range > 0, PV_EINVAL, "pv_ssim_mean: data_range must be positive");
  if (n_pairs == 0) return PV_OK;
  const double c1 = (0.01 * data_range) * (0.01 * data_range), c2 = (0.03 * data_range) * (0.03 * data_range);
  hipLaunchKernelGGL((ssim_mean_kernel<T>), dim3((unsigned)n_pairs), dim3(256), 0, as_stream(stream), a, b, (long long)stride_a,
                     (long long)stride_b, h, w, c1, c2, out);
  return check_launch("pv_ssim_mean");
}

extern "C" {

int pv_abi_version(void) { return PV_ABI_VERSION; }
const char* pv_last_error(void) { return pv::err_buf(); }

int pv_u8_from_10bit_i16(const int16_t* src, uint8_t* dst, size_t n, int mode, int32_t* range_flag,
                         void* stream) {
  return u8_launch<int16_t>(src, dst, n, mode, range_flag, stream);
}
int pv_u8_from_10bit_f32(const float* src, uint8_t* dst, size_t n, int mode, int32_t* range_flag,
                         void* stream) {
  return u8_launch<float>(src, dst, n, mode, range_flag, stream);
}

int pv_prepare_stacks_i16(const int16_t* raw, uint8_t* u8, float* out, int64_t batch, int32_t t, int32_t c, int64_t frame,
                          int32_t t_out, int mode, const float* mean, const float* std_, int32_t* range_flag, void* stream) {
  stage_mark("prepare_stacks (raw -> u8 stacks + normalised frames)", as_stream(stream));
  const int rc = prepare_stacks_launch<int16_t>(raw, u8, out, batch, t, c, frame, t_out, mode, mean, std_, range_flag, stream);
  stage_mark(nullptr, as_stream(stream));
  return rc;
}
int pv_prepare_stacks_f32(const float* raw, uint8_t* u8, float* out, int64_t batch, int32_t t, int32_t c, int64_t frame,
                          int32_t t_out, int mode, const float* mean, const float* std_, int32_t* range_flag, void* stream) {
  stage_mark("prepare_stacks (raw -> u8 stacks + normalised frames)", as_stream(stream));
  const int rc = prepare_stacks_launch<float>(raw, u8, out, batch, t, c, frame, t_out, mode, mean, std_, range_flag, stream);
  stage_mark(nullptr, as_stream(stream));
  return rc;
}

int pv_flow_weighted_mean_f32(const float* flows, const double* weights_host, float* out,
                              int64_t n_groups, int32_t n_per_group, int64_t elems, void* stream) {
  PV_REQUIRE(flows && out, PV_EINVAL, "pv_flow_weighted_mean_f32: null pointer");
  PV_REQUIRE(n_per_group >= 1 && n_per_group <= 64, PV_ESIZE,
             "pv_flow_weighted_mean_f32: n_per_group=%d outside 1..64", n_per_group);
  PV_REQUIRE(n_groups >= 0 && elems > 0, PV_EINVAL, "pv_flow_weighted_mean_f32: bad sizes");
  if (n_groups == 0) return PV_OK;
  WMeanWeights wts;
  double s = 0.0;
  for (int i = 0; i < 64; ++i) wts.w[i] = 0.0;
  for (int i = 0; i < n_per_group; ++i) {
    wts.w[i] = weights_host ? weights_host[i] : (double)(i + 1);
    s += wts.w[i];
  }
  PV_REQUIRE(s != 0.0, PV_EINVAL, "pv_flow_weighted_mean_f32: weights sum to zero");
  wts.sum = s;
  wts.inv_unused = 0.0;
  hipStream_t st = as_stream(stream);
  stage_mark("flow_weighted_mean", st);
  bool vec = (elems % 4 == 0) && ((uintptr_t)flows % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (vec) {
    hipLaunchKernelGGL((weighted_mean_kernel<4>), dim3(stream_grid((size_t)n_groups * (elems / 4), 256)),
                       dim3(256), 0, st, flows, out, n_groups, n_per_group, elems, wts);
  } else {
    hipLaunchKernelGGL((weighted_mean_kernel<1>), dim3(stream_grid((size_t)n_groups * elems, 256)),
                       dim3(256), 0, st, flows, out, n_groups, n_per_group, elems, wts);
  }
  stage_mark(nullptr, st);
  return check_launch("pv_flow_weighted_mean_f32");
}

int pv_remap_bilinear_f32(const float* src, int64_t src_stride, const float* flow, int64_t flow_stride,
                          float* dst, int64_t dst_image_stride, int64_t dst_step_stride,
                          int64_t n_images, int32_t n_steps, float step0, int32_t h, int32_t w,
                          int border_mode, float border_value, void* stream) {
  stage_mark("remap_bilinear", as_stream(stream));
  const int rc = remap_launch<float>(src, src_stride, flow, flow_stride, dst, dst_image_stride, dst_step_stride,
                                     n_images, n_steps, step0, h, w, border_mode, border_value, stream);
  stage_mark(nullptr, as_stream(stream));
  return rc;
}
int pv_remap_bilinear_u8(const uint8_t* src, int64_t src_stride, const float* flow, int64_t flow_stride,
                         uint8_t* dst, int64_t dst_image_stride, int64_t dst_step_stride,
                         int64_t n_images, int32_t n_steps, float step0, int32_t h, int32_t w,
                         int border_mode, uint8_t border_value, void* stream) {
  return remap_launch<uint8_t>(src, src_stride, flow, flow_stride, dst, dst_image_stride,
                               dst_step_stride, n_images, n_steps, step0, h, w, border_mode,
                               border_value, stream);
}

int pv_normalise_i16(const int16_t* src, float* dst, size_t n, int64_t inner, int32_t n_channels,
                     const float* mean, const float* std_, void* stream) {
  return normalise_launch<int16_t>(src, dst, n, inner, n_channels, mean, std_, stream);
}
int pv_normalise_f32(const float* src, float* dst, size_t n, int64_t inner, int32_t n_channels,
                     const float* mean, const float* std_, void* stream) {
  return normalise_launch<float>(src, dst, n, inner, n_channels, mean, std_, stream);
}

int pv_ssim_mean_u8(const uint8_t* im1, int64_t stride1, const uint8_t* im2, int64_t stride2, int64_t n_pairs, int32_t h,
                    int32_t w, double data_range, double* out, void* stream) {
  return ssim_launch<uint8_t>(im1, stride1, im2, stride2, n_pairs, h, w, data_range, out, stream);
}

int pv_ssim_mean_f32(const float* im1, int64_t stride1, const float* im2, int64_t stride2, int64_t n_pairs, int32_t h,
                     int32_t w, double data_range, double* out, void* stream) {
  return ssim_launch<float>(im1, stride1, im2, stride2, n_pairs, h, w, data_range, out, stream);
}

}  // extern "C"
