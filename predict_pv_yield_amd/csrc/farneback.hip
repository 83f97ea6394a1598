// Batched dense Farnebäck optical flow (OPTFLOW_FARNEBACK_GAUSSIAN) for gfx950.
// replaces: cv.calcOpticalFlowFarneback(prev, next, None, 0.5, 2, 40, 3, 5, 0.7, GAUSSIAN) called once per
// consecutive frame pair from a process pool (notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:
// 122-135, 175-240; same arguments in notebooks/optical_flow_1.ipynb:214-221).
// Algorithm: SURVEY.md Appendix A.1 (OpenCV modules/video/src/optflowgf.cpp), restated on the CPU in
// oracle/pv_oracle.c; this file follows the same operation order so the two agree to float rounding.
//
// All pairs of the batch advance through the pyramid together: every stage is ONE launch over
// [n_pairs] x level image, so a [B, 12, 11, 64, 64] stack (3872 pairs/32 samples) is ~25 launches.
// Stages per level:  prep (u8 -> f32, Gaussian smooth, resize)  ->  PolyExp (vertical, horizontal)
//                    -> UpdateMatrices -> iterations x { window blur V, window blur H + 2x2 solve,
//                    UpdateMatrices }  -> (next level) bilinear flow upsample x 1/pyr_scale.
#include <type_traits>
#include "pv_common.h"
#include <stdlib.h>

namespace pv {

struct FbTaps {
  float k[64];  // generic tap table (smooth kernel: full ksize taps; window: k[0..m])
  int n;
};
struct FbPoly {
  float g[8], xg[8], xxg[8];  // taps 0..n (poly_n <= 7); odd symmetry handled in the kernel
  double ig11, ig03, ig33, ig55;
  int n;
};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
__device__ __forceinline__ int clampi_d(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- prep: blurred + resized f32 image of one pyramid level ---------------------------------------
// blurred(y, x): separable Gaussian of the u8 image, rows first then columns, BORDER_REFLECT_101
__device__ __forceinline__ float row_filter(const uint8_t* __restrict__ s, int w, int x, const FbTaps& kt) {
  const int ks = kt.n, r = ks >> 1;
  float acc;
  if (ks <= 5) {
    acc = (float)s[x] * kt.k[r];
    for (int i = 1; i <= r; ++i)
      acc = __fadd_rn(acc, __fmul_rn((float)s[reflect101(x - i, w)] + (float)s[reflect101(x + i, w)], kt.k[r + i]));
  } else {
    acc = (float)s[reflect101(x - r, w)] * kt.k[0];
    for (int i = 1; i < ks; ++i) acc = __fadd_rn(acc, __fmul_rn((float)s[reflect101(x + i - r, w)], kt.k[i]));
  }
  return acc;
}
__device__ __forceinline__ float blurred_at(const uint8_t* __restrict__ img, int h, int w, int y, int x,
                                            const FbTaps& kt) {
  const int r = kt.n >> 1;
  float acc = __fmul_rn(row_filter(img + (size_t)y * w, w, x, kt), kt.k[r]);
  for (int i = 1; i <= r; ++i) {
    float a = row_filter(img + (size_t)reflect101(y - i, h) * w, w, x, kt);
    float b = row_filter(img + (size_t)reflect101(y + i, h) * w, w, x, kt);
    acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(a, b), kt.k[r + i]));
  }
  return acc;
}

// Image index -> source image.  Unchained: image im = (pair im / 2, prev | next).  CHAINED (chain_f = frames per group > 0:
// the pairs of a group are consecutive frames of one stack, next == prev + prev_stride): image im = frame (im % chain_f) of
// group (im / chain_f) -- a frame is the `next` of one pair and the `prev` of the following one, and everything computed per
// image (smoothing, resize, PolyExp) depends on the image alone, so it is computed ONCE per frame: T frames instead of
// 2 (T - 1) images per group (12 instead of 22 for the 12-frame stacks of the advection pipeline).
__device__ __forceinline__ const uint8_t* fb_image_of(long long im, const uint8_t* __restrict__ prev,
                                                      const uint8_t* __restrict__ next, long long prev_stride,
                                                      long long next_stride, long long pairs_per_group,
                                                      long long group_stride, int chain_f) {
  if (chain_f > 0) {
    const long long grp = im / chain_f, f = im - grp * chain_f;
    return prev + grp * group_stride + f * prev_stride;
  }
  const long long p = im >> 1;
  const long long grp = p / pairs_per_group, q = p - grp * pairs_per_group;
  return ((im & 1) ? next + q * next_stride : prev + q * prev_stride) + grp * group_stride;
}

// index of the R images of pair p: (first, second)
__device__ __forceinline__ void fb_r_images_of(long long p, long long pairs_per_group, int chain_f, long long* r0, long long* r1) {
  if (chain_f > 0) {
    const long long grp = p / pairs_per_group, q = p - grp * pairs_per_group;
    *r0 = grp * chain_f + q;
    *r1 = *r0 + 1;
  } else {
    *r0 = 2 * p;
    *r1 = 2 * p + 1;
  }
}

// images: see fb_image_of.  I: [n_img][lh][lw]
__global__ __launch_bounds__(256) void fb_prep_kernel(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ next,
                                                       long long prev_stride, long long next_stride,
                                                       long long pairs_per_group, long long group_stride,
                                                       float* __restrict__ I, long long n_img, int chain_f, int h, int w, int lh,
                                                       int lw, int mode /*0 copy, 1 area 2x2, 2 bilinear*/,
                                                       double inv_fx, double inv_fy, FbTaps kt) {
  const long long per_img = (long long)lh * lw;
  const long long total = n_img * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long im = i / per_img;
    int rem = (int)(i - im * per_img);
    int y = rem / lw, x = rem - y * lw;
    const uint8_t* img = fb_image_of(im, prev, next, prev_stride, next_stride, pairs_per_group, group_stride, chain_f);
    float v;
    if (mode == 0) {
      v = blurred_at(img, h, w, y, x, kt);
    } else if (mode == 1) {
      float a = __fadd_rn(blurred_at(img, h, w, 2 * y, 2 * x, kt), blurred_at(img, h, w, 2 * y, 2 * x + 1, kt));
      float b = __fadd_rn(blurred_at(img, h, w, 2 * y + 1, 2 * x, kt), blurred_at(img, h, w, 2 * y + 1, 2 * x + 1, kt));
      v = __fmul_rn(__fadd_rn(a, b), 0.25f);
    } else {
      float fx = (float)((x + 0.5) * inv_fx - 0.5);
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= w - 1) { fx = 0; sx = w - 1; }
      float fy = (float)((y + 0.5) * inv_fy - 0.5);
      int sy = (int)floorf(fy);
      fy -= sy;
      if (sy < 0) { fy = 0; sy = 0; }
      if (sy >= h - 1) { fy = 0; sy = h - 1; }
      int sy1 = clampi_d(sy + 1, 0, h - 1);
      float r0, r1;
      if (sx + 1 < w) {
        float a0 = 1.f - fx, a1 = fx;
        r0 = __fadd_rn(__fmul_rn(blurred_at(img, h, w, sy, sx, kt), a0), __fmul_rn(blurred_at(img, h, w, sy, sx + 1, kt), a1));
        r1 = __fadd_rn(__fmul_rn(blurred_at(img, h, w, sy1, sx, kt), a0), __fmul_rn(blurred_at(img, h, w, sy1, sx + 1, kt), a1));
      } else {
        r0 = blurred_at(img, h, w, sy, sx, kt);
        r1 = blurred_at(img, h, w, sy1, sx, kt);
      }
      v = __fadd_rn(__fmul_rn(r0, 1.f - fy), __fmul_rn(r1, fy));
    }
    I[i] = v;
  }
}

// ---- PolyExp ---------------------------------------------------------------------------------------
// vertical pass: T[img][y][x] = (t0, t1, t2), rows replicated at the border
__global__ __launch_bounds__(256) void fb_polyexp_v_kernel(const float* __restrict__ I, float* __restrict__ T,
                                                            long long n_img, int lh, int lw, FbPoly pk) {
  const long long per_img = (long long)lh * lw;
  const long long total = n_img * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long im = i / per_img;
    int rem = (int)(i - im * per_img);
    int y = rem / lw, x = rem - y * lw;
    const float* src = I + im * per_img;
    float t0 = __fmul_rn(src[(size_t)y * lw + x], pk.g[0]), t1 = 0.f, t2 = 0.f;
    for (int k = 1; k <= pk.n; ++k) {
      float s0 = src[(size_t)max(y - k, 0) * lw + x];
      float s1 = src[(size_t)min(y + k, lh - 1) * lw + x];
      float p = __fadd_rn(s0, s1);
      t0 = __fadd_rn(t0, __fmul_rn(pk.g[k], p));
      t1 = __fadd_rn(t1, __fmul_rn(pk.xg[k], __fsub_rn(s1, s0)));
      t2 = __fadd_rn(t2, __fmul_rn(pk.xxg[k], p));
    }
    float* dst = T + i * 3;
    dst[0] = t0; dst[1] = t1; dst[2] = t2;
  }
}

// horizontal pass (double accumulators, edge triples replicated) -> R[img][y][x][5]
__global__ __launch_bounds__(256) void fb_polyexp_h_kernel(const float* __restrict__ T, float* __restrict__ R,
                                                            long long n_img, int lh, int lw, FbPoly pk) {
  const long long per_img = (long long)lh * lw;
  const long long total = n_img * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long im = i / per_img;
    int rem = (int)(i - im * per_img);
    int y = rem / lw, x = rem - y * lw;
    const float* row = T + (im * per_img + (size_t)y * lw) * 3;
    const float* c = row + (size_t)x * 3;
    float g0 = pk.g[0];
    double b1 = __fmul_rn(c[0], g0), b2 = 0, b3 = __fmul_rn(c[1], g0), b4 = 0, b5 = __fmul_rn(c[2], g0), b6 = 0;
    for (int k = 1; k <= pk.n; ++k) {
      const float* rp = row + (size_t)min(x + k, lw - 1) * 3;
      const float* rm = row + (size_t)max(x - k, 0) * 3;
      double tg = (double)__fadd_rn(rp[0], rm[0]);
      g0 = pk.g[k];
      b1 = __dadd_rn(b1, __dmul_rn(tg, (double)g0));
      b4 = __dadd_rn(b4, __dmul_rn(tg, (double)pk.xxg[k]));
      b2 = __dadd_rn(b2, (double)__fmul_rn(__fsub_rn(rp[0], rm[0]), pk.xg[k]));
      b3 = __dadd_rn(b3, (double)__fmul_rn(__fadd_rn(rp[1], rm[1]), g0));
      b6 = __dadd_rn(b6, (double)__fmul_rn(__fsub_rn(rp[1], rm[1]), pk.xg[k]));
      b5 = __dadd_rn(b5, (double)__fmul_rn(__fadd_rn(rp[2], rm[2]), g0));
    }
    float* d = R + i * 5;
    d[1] = (float)__dmul_rn(b2, pk.ig11);
    d[0] = (float)__dmul_rn(b3, pk.ig11);
    d[3] = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b4, pk.ig33));
    d[2] = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b5, pk.ig33));
    d[4] = (float)__dmul_rn(b6, pk.ig55);
  }
}

// ---- prep + PolyExp fused for source images up to 64 x 64 (the PV-site tiles) ---------------------------------------
// One workgroup = one image of a pair: the u8 image, its separable Gaussian, the resized level image I and the vertical
// PolyExp triples T all live in LDS; only the 5 polynomial coefficients R go to memory (the three-kernel path writes and
// re-reads I and T: 130 bytes per level-pixel of extra traffic and two more launches).  Every expression is the one of
// fb_prep_kernel / fb_polyexp_v_kernel / fb_polyexp_h_kernel, evaluated in the same order: bit-identical results.
// 64 KB of LDS (the first PolyExp plane reuses the buffer of the blurred image, dead by then) and <= 64 registers: TWO
// 16-wave workgroups share a CU, so one's barrier-separated phases overlap the other's.  (Round 4 tried four pixels of a
// row per thread with 16-byte window reads and 8-wave workgroups: no faster -- 254 against 198 us at level 0 -- the phases
// are short and barrier-separated, and half the waves cover each other's LDS round trips half as well; and planes padded with
// replicated borders so that every tap is an immediate offset, taps unrolled over a template parameter: 178 against 156 us --
// at 64 registers the unrolled taps spill.)
constexpr int FB_PP_NT = 1024;
typedef float fb_pp_f2 __attribute__((ext_vector_type(2)));
template <bool F64_ACC>      // F64_ACC: the horizontal pass on double accumulators (the reference's); else f32 with fused multiply-adds
__global__ __launch_bounds__(FB_PP_NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void fb_prep_polyexp_tile_kernel(const uint8_t* __restrict__ prev,
                                                                    const uint8_t* __restrict__ next, long long prev_stride,
                                                                    long long next_stride, long long pairs_per_group,
                                                                    long long group_stride, float* __restrict__ R,
                                                                    long long n_img, int chain_f, int h, int w, int lh, int lw, int mode,
                                                                    double inv_fx, double inv_fy, FbTaps kt, FbPoly pk, int planar) {
  // planar != 0: pair-planar R (per image (c0, c1) float2 [lh][lw] | (c2, c3) float2 [lh][lw] | c4 float [lh][lw]: what
  // fb_level_u_kernel stages and gathers; lh * lw a multiple of 2), else R[img][lh][lw][5]
  __shared__ float bufA[64 * 64];        // source as float, later the blurred image, later PolyExp plane t0
  __shared__ float bufB[64 * 64];        // row-filtered image, later the level image I
  __shared__ float Tt12[2 * 64 * 64];    // vertical PolyExp planes t1, t2
  const int tid = threadIdx.x;
  const int ks = kt.n, r = ks >> 1;
  // row index of a flat pixel index: a shift for power-of-two widths (the usual 64 / 32), else a division
  const int sh_w = (w & (w - 1)) == 0 ? __builtin_ctz(w) : -1, sh_lw = (lw & (lw - 1)) == 0 ? __builtin_ctz(lw) : -1;
  auto row_w = [&](int i) { return sh_w >= 0 ? i >> sh_w : i / w; };
  auto row_lw = [&](int i) { return sh_lw >= 0 ? i >> sh_lw : i / lw; };
  for (long long im = blockIdx.x; im < n_img; im += gridDim.x) {
    const uint8_t* img = fb_image_of(im, prev, next, prev_stride, next_stride, pairs_per_group, group_stride, chain_f);
    const int npx = h * w, lpx = lh * lw;
    for (int i = tid; i < npx; i += FB_PP_NT) bufA[i] = (float)img[i];
    __syncthreads();
    // row filter (BORDER_REFLECT_101), same tap order as row_filter()
    for (int i = tid; i < npx; i += FB_PP_NT) {
      const int y = row_w(i), x = i - y * w;
      const float* srow = bufA + y * w;
      float acc;
      if (ks <= 5) {
        acc = srow[x] * kt.k[r];
        for (int t = 1; t <= r; ++t)
          acc = __fadd_rn(acc, __fmul_rn(srow[reflect101(x - t, w)] + srow[reflect101(x + t, w)], kt.k[r + t]));
      } else {
        acc = srow[reflect101(x - r, w)] * kt.k[0];
        for (int t = 1; t < ks; ++t) acc = __fadd_rn(acc, __fmul_rn(srow[reflect101(x + t - r, w)], kt.k[t]));
      }
      bufB[i] = acc;
    }
    __syncthreads();
    // column filter -> blurred image (over the source, which is no longer needed)
    for (int i = tid; i < npx; i += FB_PP_NT) {
      const int y = row_w(i), x = i - y * w;
      float acc = __fmul_rn(bufB[i], kt.k[r]);
      for (int t = 1; t <= r; ++t) {
        const float a = bufB[reflect101(y - t, h) * w + x], b = bufB[reflect101(y + t, h) * w + x];
        acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(a, b), kt.k[r + t]));
      }
      bufA[i] = acc;
    }
    __syncthreads();
    // resize to the level image I (into bufB)
    for (int i = tid; i < lpx; i += FB_PP_NT) {
      const int y = row_lw(i), x = i - y * lw;
      float v;
      if (mode == 0) {
        v = bufA[i];
      } else if (mode == 1) {
        const float a = __fadd_rn(bufA[(2 * y) * w + 2 * x], bufA[(2 * y) * w + 2 * x + 1]);
        const float b = __fadd_rn(bufA[(2 * y + 1) * w + 2 * x], bufA[(2 * y + 1) * w + 2 * x + 1]);
        v = __fmul_rn(__fadd_rn(a, b), 0.25f);
      } else {
        float fx = (float)((x + 0.5) * inv_fx - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= w - 1) { fx = 0; sx = w - 1; }
        float fy = (float)((y + 0.5) * inv_fy - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= h - 1) { fy = 0; sy = h - 1; }
        const int sy1 = clampi_d(sy + 1, 0, h - 1);
        float r0, r1;
        if (sx + 1 < w) {
          const float a0 = 1.f - fx, a1 = fx;
          r0 = __fadd_rn(__fmul_rn(bufA[sy * w + sx], a0), __fmul_rn(bufA[sy * w + sx + 1], a1));
          r1 = __fadd_rn(__fmul_rn(bufA[sy1 * w + sx], a0), __fmul_rn(bufA[sy1 * w + sx + 1], a1));
        } else {
          r0 = bufA[sy * w + sx];
          r1 = bufA[sy1 * w + sx];
        }
        v = __fadd_rn(__fmul_rn(r0, 1.f - fy), __fmul_rn(r1, fy));
      }
      bufB[i] = v;
    }
    __syncthreads();
    // PolyExp, vertical pass -> T (planar)
    for (int i = tid; i < lpx; i += FB_PP_NT) {
      const int y = row_lw(i), x = i - y * lw;
      float t0 = __fmul_rn(bufB[i], pk.g[0]), t1 = 0.f, t2 = 0.f;
      for (int k = 1; k <= pk.n; ++k) {
        const float s0 = bufB[max(y - k, 0) * lw + x];
        const float s1 = bufB[min(y + k, lh - 1) * lw + x];
        const float pp = __fadd_rn(s0, s1);
        t0 = __fadd_rn(t0, __fmul_rn(pk.g[k], pp));
        t1 = __fadd_rn(t1, __fmul_rn(pk.xg[k], __fsub_rn(s1, s0)));
        t2 = __fadd_rn(t2, __fmul_rn(pk.xxg[k], pp));
      }
      bufA[i] = t0, Tt12[i] = t1, Tt12[lpx + i] = t2;   // bufA (blurred image) was consumed by the resize above
    }
    __syncthreads();
    // PolyExp, horizontal pass -> R.  F64_ACC: the reference's double accumulators.  Otherwise f32 accumulators and fused
    // multiply-adds (one rounding per tap: ~3e-7 relative on R, ~1e-6 px on the flow, against a contract of 1e-3 px): the
    // double form is 45 % of this kernel's vector instructions (half rate, a conversion per operand)
    for (int i = tid; i < lpx; i += FB_PP_NT) {
      const int y = row_lw(i), x = i - y * lw;
      const float* t0r = bufA + y * lw;
      const float* t1r = Tt12 + y * lw;
      const float* t2r = Tt12 + lpx + y * lw;
      float g0 = pk.g[0];
      float d0, d1, d2, d3, d4;
      if constexpr (F64_ACC) {
        double b1 = __fmul_rn(t0r[x], g0), b2 = 0, b3 = __fmul_rn(t1r[x], g0), b4 = 0, b5 = __fmul_rn(t2r[x], g0), b6 = 0;
        for (int k = 1; k <= pk.n; ++k) {
          const int xp = min(x + k, lw - 1), xm = max(x - k, 0);
          const double tg = (double)__fadd_rn(t0r[xp], t0r[xm]);
          g0 = pk.g[k];
          b1 = __dadd_rn(b1, __dmul_rn(tg, (double)g0));
          b4 = __dadd_rn(b4, __dmul_rn(tg, (double)pk.xxg[k]));
          b2 = __dadd_rn(b2, (double)__fmul_rn(__fsub_rn(t0r[xp], t0r[xm]), pk.xg[k]));
          b3 = __dadd_rn(b3, (double)__fmul_rn(__fadd_rn(t1r[xp], t1r[xm]), g0));
          b6 = __dadd_rn(b6, (double)__fmul_rn(__fsub_rn(t1r[xp], t1r[xm]), pk.xg[k]));
          b5 = __dadd_rn(b5, (double)__fmul_rn(__fadd_rn(t2r[xp], t2r[xm]), g0));
        }
        d1 = (float)__dmul_rn(b2, pk.ig11), d0 = (float)__dmul_rn(b3, pk.ig11);
        d3 = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b4, pk.ig33));
        d2 = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b5, pk.ig33));
        d4 = (float)__dmul_rn(b6, pk.ig55);
      } else {
        float b1 = __fmul_rn(t0r[x], g0), b2 = 0.f, b3 = __fmul_rn(t1r[x], g0), b4 = 0.f, b5 = __fmul_rn(t2r[x], g0), b6 = 0.f;
        for (int k = 1; k <= pk.n; ++k) {
          const int xp = min(x + k, lw - 1), xm = max(x - k, 0);
          const float p0 = t0r[xp], m0 = t0r[xm], p1 = t1r[xp], m1 = t1r[xm], p2 = t2r[xp], m2 = t2r[xm];
          const float gk = pk.g[k], xgk = pk.xg[k];
          const float tg = __fadd_rn(p0, m0);
          b1 = __builtin_fmaf(tg, gk, b1);
          b4 = __builtin_fmaf(tg, pk.xxg[k], b4);
          b2 = __builtin_fmaf(__fsub_rn(p0, m0), xgk, b2);
          b3 = __builtin_fmaf(__fadd_rn(p1, m1), gk, b3);
          b6 = __builtin_fmaf(__fsub_rn(p1, m1), xgk, b6);
          b5 = __builtin_fmaf(__fadd_rn(p2, m2), gk, b5);
        }
        const float ig11 = (float)pk.ig11, ig03 = (float)pk.ig03, ig33 = (float)pk.ig33, ig55 = (float)pk.ig55;
        const float b1s = __fmul_rn(b1, ig03);
        d1 = __fmul_rn(b2, ig11), d0 = __fmul_rn(b3, ig11);
        d3 = __builtin_fmaf(b4, ig33, b1s);
        d2 = __builtin_fmaf(b5, ig33, b1s);
        d4 = __fmul_rn(b6, ig55);
      }
      if (planar) {
        float* d = R + im * lpx * 5;
        reinterpret_cast<fb_pp_f2*>(d)[i] = (fb_pp_f2){d0, d1};
        reinterpret_cast<fb_pp_f2*>(d + 2 * lpx)[i] = (fb_pp_f2){d2, d3};
        d[4 * lpx + i] = d4;
      } else {
        float* d = R + (im * lpx + i) * 5;
        d[0] = d0, d[1] = d1, d[2] = d2, d[3] = d3, d[4] = d4;
      }
    }
    __syncthreads();   // LDS is reused by the next image
  }
}

// ---- prep + PolyExp fused for FRAMES (source images larger than a tile: the 704 x 548 images of the notebooks) ----------
// One workgroup = one ty x tx tile of a level image.  What the tile needs of each stage lives in LDS with its halo: the
// source window as floats -> row-filtered -> blurred (the level's Gaussian, BORDER_REFLECT_101 in image coordinates) ->
// the level image I on the tile + poly_n rows / columns (coordinates clamped to the image: the PolyExp taps that replicate
// the border become plain offsets) -> the vertical PolyExp planes -> R.  I and T never go to memory and every index is 32-bit
// (the three-kernel path writes and re-reads 32 bytes per level pixel and divides 64-bit indices per pixel).  Expressions as in
// fb_prep_kernel / fb_polyexp_v_kernel / fb_polyexp_h_kernel (double accumulators in the horizontal pass): identical bits.
struct FbFrameTile {
  int ty, tx;            // tile of the level image
  int n_ty, n_tx;        // tiles per image
  int cap_sy, cap_sx;    // capacity (rows, columns) of the source window; the other windows fit inside it
};
constexpr int FB_FR_NT = 512;
__global__ __launch_bounds__(FB_FR_NT) void fb_prep_polyexp_frame_kernel(const uint8_t* __restrict__ prev,
                                                                          const uint8_t* __restrict__ next, long long prev_stride,
                                                                          long long next_stride, long long pairs_per_group,
                                                                          long long group_stride, float* __restrict__ R,
                                                                          long long n_img, int chain_f, int h, int w, int lh, int lw,
                                                                          int mode, double inv_fx, double inv_fy, FbTaps kt, FbPoly pk,
                                                                          FbFrameTile ft, int stage, float* __restrict__ Iimg) {
  // stage 0: everything (levels at the source's scale).  Coarse levels, whose tile + PolyExp halo would sample a source window
  // many times the tile: stage 1 = smoothing + resize of a tile WITHOUT halo -> the level image Iimg[n_img][lh][lw];
  // stage 2 = both PolyExp passes from Iimg (tile + halo).
  extern __shared__ float fr_lds[];
  const int tid = threadIdx.x;
  const int ks = kt.n, r = ks >> 1, n = stage == 1 ? 0 : pk.n;
  const int cap = ft.cap_sy * ft.cap_sx;
  float* bufA = fr_lds;                 // source window as floats, later the blurred window
  float* bufB = fr_lds + cap;           // row-filtered window, later the level image I (tile + halo)
  float* T0 = fr_lds + 2 * cap;         // vertical PolyExp planes [ty][tx + 2 n]
  const int IY = ft.ty + 2 * n, IX = ft.tx + 2 * n;
  float* T1 = T0 + ft.ty * IX;
  float* T2 = T1 + ft.ty * IX;
  // source row / column the level pixel (sampling mode 2) starts from, as in fb_prep_kernel
  auto src_lo = [&](int v, double inv, int lim) {
    float f = (float)((v + 0.5) * inv - 0.5);
    int sv = (int)floorf(f);
    if (sv < 0) sv = 0;
    if (sv >= lim - 1) sv = lim - 1;
    return sv;
  };
  const long long tiles_per_img = (long long)ft.n_ty * ft.n_tx;
  for (long long t = blockIdx.x; t < n_img * tiles_per_img; t += gridDim.x) {
    const long long im = t / tiles_per_img;
    const int tt = (int)(t - im * tiles_per_img);
    const int ty0 = (tt / ft.n_tx) * ft.ty, tx0 = (tt % ft.n_tx) * ft.tx;
    const uint8_t* img = fb_image_of(im, prev, next, prev_stride, next_stride, pairs_per_group, group_stride, chain_f);
    // level rows / columns the tile touches (clamped), the blurred rows / columns those sample, the rows / columns of the
    // source the two filter passes read (reflections of rows beyond the border fall inside the range)
    const int iy_lo = max(ty0 - n, 0), iy_hi = min(ty0 + ft.ty + n - 1, lh - 1);
    const int ix_lo = max(tx0 - n, 0), ix_hi = min(tx0 + ft.tx + n - 1, lw - 1);
    int by_lo, by_hi, bx_lo, bx_hi;
    if (mode == 0) {
      by_lo = iy_lo, by_hi = iy_hi, bx_lo = ix_lo, bx_hi = ix_hi;
    } else if (mode == 1) {
      by_lo = 2 * iy_lo, by_hi = 2 * iy_hi + 1, bx_lo = 2 * ix_lo, bx_hi = 2 * ix_hi + 1;
    } else {
      by_lo = src_lo(iy_lo, inv_fy, h), by_hi = min(src_lo(iy_hi, inv_fy, h) + 1, h - 1);
      bx_lo = src_lo(ix_lo, inv_fx, w), bx_hi = min(src_lo(ix_hi, inv_fx, w) + 1, w - 1);
    }
    const int ry_lo = max(by_lo - r, 0), ry_hi = min(by_hi + r, h - 1);
    const int sx_lo = max(bx_lo - r, 0), sx_hi = min(bx_hi + r, w - 1);
    const int SY = ry_hi - ry_lo + 1, SX = sx_hi - sx_lo + 1;       // source window
    const int BY = by_hi - by_lo + 1, BX = bx_hi - bx_lo + 1;       // blurred window
    if (stage == 2) {   // the level image exists: tile + halo, coordinates clamped to the image
      for (int i = tid; i < IY * IX; i += FB_FR_NT) {
        const int yy = i / IX, xx = i - yy * IX;
        const int y = min(max(ty0 - n + yy, 0), lh - 1), x = min(max(tx0 - n + xx, 0), lw - 1);
        bufB[i] = Iimg[(im * lh + y) * (long long)lw + x];
      }
      __syncthreads();
    } else {
    // 1. the source window as floats
    const unsigned inv_sx = 0xffffffffu / (unsigned)SX + 1;
    for (int i0 = tid; i0 < SY * SX; i0 += 4 * FB_FR_NT) {   // four byte loads in flight per thread
      uint8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * FB_FR_NT, SY * SX - 1);
        const int y = (int)__umulhi((unsigned)i, inv_sx), x = i - y * SX;
        v[u] = img[(size_t)(ry_lo + y) * w + sx_lo + x];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * FB_FR_NT < SY * SX) bufA[i0 + u * FB_FR_NT] = (float)v[u];
    }
    __syncthreads();
    // 2. row filter on the window's rows, for the blurred window's columns (tap order of row_filter()).  Tiles whose taps stay
    // inside the image (all but the frame's rim) with the 3- and 9-tap kernels of the reference's pyramid: no reflection, taps
    // unrolled; flat index -> (row, column) by a multiplication with the rounded-up reciprocal (exact below 2^16 rows x columns)
    const unsigned inv_bx = 0xffffffffu / (unsigned)BX + 1;
    const bool inner_x = bx_lo - r >= 0 && bx_hi + r <= w - 1, inner_y = by_lo - r >= 0 && by_hi + r <= h - 1;
    if (inner_x && ks == 3) {
      for (int i = tid; i < SY * BX; i += FB_FR_NT) {
        const int y = (int)__umulhi((unsigned)i, inv_bx), x = i - y * BX;
        const float* sp = bufA + y * SX + (bx_lo - sx_lo) + x;
        float acc = sp[0] * kt.k[1];
        acc = __fadd_rn(acc, __fmul_rn(sp[-1] + sp[1], kt.k[2]));
        bufB[i] = acc;
      }
    } else if (inner_x && ks == 9) {
      for (int i = tid; i < SY * BX; i += FB_FR_NT) {
        const int y = (int)__umulhi((unsigned)i, inv_bx), x = i - y * BX;
        const float* sp = bufA + y * SX + (bx_lo - sx_lo) + x - 4;
        float acc = sp[0] * kt.k[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) acc = __fadd_rn(acc, __fmul_rn(sp[q], kt.k[q]));
        bufB[i] = acc;
      }
    } else {
    for (int i = tid; i < SY * BX; i += FB_FR_NT) {
      const int y = i / BX, x = i - y * BX;
      const int ax = bx_lo + x;
      const float* srow = bufA + y * SX - sx_lo;
      float acc;
      if (ks <= 5) {
        acc = srow[ax] * kt.k[r];
        for (int q = 1; q <= r; ++q)
          acc = __fadd_rn(acc, __fmul_rn(srow[reflect101(ax - q, w)] + srow[reflect101(ax + q, w)], kt.k[r + q]));
      } else {
        acc = srow[reflect101(ax - r, w)] * kt.k[0];
        for (int q = 1; q < ks; ++q) acc = __fadd_rn(acc, __fmul_rn(srow[reflect101(ax + q - r, w)], kt.k[q]));
      }
      bufB[i] = acc;
    }
    }
    __syncthreads();
    // 3. column filter -> the blurred window (over the source window, dead by now)
    if (inner_y && ks == 3) {
      for (int i = tid; i < BY * BX; i += FB_FR_NT) {
        const float* cp = bufB + i + (by_lo - ry_lo) * BX;
        float acc = __fmul_rn(cp[0], kt.k[1]);
        acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(cp[-BX], cp[BX]), kt.k[2]));
        bufA[i] = acc;
      }
    } else if (inner_y && ks == 9) {
      for (int i = tid; i < BY * BX; i += FB_FR_NT) {
        const float* cp = bufB + i + (by_lo - ry_lo) * BX;
        float acc = __fmul_rn(cp[0], kt.k[4]);
#pragma unroll
        for (int q = 1; q <= 4; ++q) acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(cp[-q * BX], cp[q * BX]), kt.k[4 + q]));
        bufA[i] = acc;
      }
    } else {
    for (int i = tid; i < BY * BX; i += FB_FR_NT) {
      const int y = i / BX, x = i - y * BX;
      const int ay = by_lo + y;
      const float* col = bufB + x - ry_lo * BX;
      float acc = __fmul_rn(col[ay * BX], kt.k[r]);
      for (int q = 1; q <= r; ++q) {
        const float a = col[reflect101(ay - q, h) * BX], b = col[reflect101(ay + q, h) * BX];
        acc = __fadd_rn(acc, __fmul_rn(__fadd_rn(a, b), kt.k[r + q]));
      }
      bufA[i] = acc;
    }
    }
    __syncthreads();
    // 4. the level image on the tile + halo, coordinates clamped to the image (into the row-filtered window's buffer)
    const float* bl = bufA - by_lo * BX - bx_lo;    // blurred(y, x) = bl[y * BX + x]
    const unsigned inv_ix4 = 0xffffffffu / (unsigned)IX + 1;
    for (int i = tid; i < IY * IX; i += FB_FR_NT) {
      const int yy = (int)__umulhi((unsigned)i, inv_ix4), xx = i - yy * IX;
      const int y = min(max(ty0 - n + yy, 0), lh - 1), x = min(max(tx0 - n + xx, 0), lw - 1);
      float v;
      if (mode == 0) {
        v = bl[y * BX + x];
      } else if (mode == 1) {
        const float a = __fadd_rn(bl[(2 * y) * BX + 2 * x], bl[(2 * y) * BX + 2 * x + 1]);
        const float b = __fadd_rn(bl[(2 * y + 1) * BX + 2 * x], bl[(2 * y + 1) * BX + 2 * x + 1]);
        v = __fmul_rn(__fadd_rn(a, b), 0.25f);
      } else {
        float fx = (float)((x + 0.5) * inv_fx - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= w - 1) { fx = 0; sx = w - 1; }
        float fy = (float)((y + 0.5) * inv_fy - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= h - 1) { fy = 0; sy = h - 1; }
        const int sy1 = clampi_d(sy + 1, 0, h - 1);
        float r0, r1;
        if (sx + 1 < w) {
          const float a0 = 1.f - fx, a1 = fx;
          r0 = __fadd_rn(__fmul_rn(bl[sy * BX + sx], a0), __fmul_rn(bl[sy * BX + sx + 1], a1));
          r1 = __fadd_rn(__fmul_rn(bl[sy1 * BX + sx], a0), __fmul_rn(bl[sy1 * BX + sx + 1], a1));
        } else {
          r0 = bl[sy * BX + sx];
          r1 = bl[sy1 * BX + sx];
        }
        v = __fadd_rn(__fmul_rn(r0, 1.f - fy), __fmul_rn(r1, fy));
      }
      if (stage == 1) {
        if (ty0 + yy < lh && tx0 + xx < lw) Iimg[(im * lh + ty0 + yy) * (long long)lw + tx0 + xx] = v;
      } else {
        bufB[i] = v;
      }
    }
    __syncthreads();
    if (stage == 1) continue;
    }
    // 5. PolyExp, vertical pass on the tile's rows, every column of the halo (taps unrolled for the reference's poly_n = 5)
    const unsigned inv_ix = 0xffffffffu / (unsigned)IX + 1, inv_tx = 0xffffffffu / (unsigned)ft.tx + 1;
    auto vertical = [&](auto nn) __attribute__((always_inline)) {
      constexpr int NN = decltype(nn)::value;
      const int nt = NN ? NN : n;
      for (int i = tid; i < ft.ty * IX; i += FB_FR_NT) {
        const int y = (int)__umulhi((unsigned)i, inv_ix);
        const float* c = bufB + i + n * IX;
        float t0 = __fmul_rn(c[0], pk.g[0]), t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 1; k <= nt; ++k) {
          const float s0 = c[-k * IX];
          const float s1 = c[k * IX];
          const float pp = __fadd_rn(s0, s1);
          t0 = __fadd_rn(t0, __fmul_rn(pk.g[k], pp));
          t1 = __fadd_rn(t1, __fmul_rn(pk.xg[k], __fsub_rn(s1, s0)));
          t2 = __fadd_rn(t2, __fmul_rn(pk.xxg[k], pp));
        }
        (void)y;
        T0[i] = t0, T1[i] = t1, T2[i] = t2;
      }
    };
    if (n == 5) vertical(std::integral_constant<int, 5>{}); else vertical(std::integral_constant<int, 0>{});
    __syncthreads();
    // 6. PolyExp, horizontal pass (the reference's double accumulators) -> R[img][lh][lw][5]
    auto horizontal = [&](auto nn) __attribute__((always_inline)) {
      constexpr int NN = decltype(nn)::value;
      const int nt = NN ? NN : n;
      for (int i = tid; i < ft.ty * ft.tx; i += FB_FR_NT) {
        const int y = (int)__umulhi((unsigned)i, inv_tx), x = i - y * ft.tx;
        if (ty0 + y >= lh || tx0 + x >= lw) continue;
        const float* t0r = T0 + y * IX + x + n;
        const float* t1r = T1 + y * IX + x + n;
        const float* t2r = T2 + y * IX + x + n;
        float g0 = pk.g[0];
        double b1 = __fmul_rn(t0r[0], g0), b2 = 0, b3 = __fmul_rn(t1r[0], g0), b4 = 0, b5 = __fmul_rn(t2r[0], g0), b6 = 0;
#pragma unroll
        for (int k = 1; k <= nt; ++k) {
          const double tg = (double)__fadd_rn(t0r[k], t0r[-k]);
          g0 = pk.g[k];
          b1 = __dadd_rn(b1, __dmul_rn(tg, (double)g0));
          b4 = __dadd_rn(b4, __dmul_rn(tg, (double)pk.xxg[k]));
          b2 = __dadd_rn(b2, (double)__fmul_rn(__fsub_rn(t0r[k], t0r[-k]), pk.xg[k]));
          b3 = __dadd_rn(b3, (double)__fmul_rn(__fadd_rn(t1r[k], t1r[-k]), g0));
          b6 = __dadd_rn(b6, (double)__fmul_rn(__fsub_rn(t1r[k], t1r[-k]), pk.xg[k]));
          b5 = __dadd_rn(b5, (double)__fmul_rn(__fadd_rn(t2r[k], t2r[-k]), g0));
        }
        float* d = R + ((im * lh + ty0 + y) * (long long)lw + tx0 + x) * 5;
        d[1] = (float)__dmul_rn(b2, pk.ig11);
        d[0] = (float)__dmul_rn(b3, pk.ig11);
        d[3] = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b4, pk.ig33));
        d[2] = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b5, pk.ig33));
        d[4] = (float)__dmul_rn(b6, pk.ig55);
      }
    };
    if (n == 5) horizontal(std::integral_constant<int, 5>{}); else horizontal(std::integral_constant<int, 0>{});
    __syncthreads();   // LDS is reused by the next tile
  }
}

// ---- UpdateMatrices --------------------------------------------------------------------------------
// one pixel of FarnebackUpdateMatrices: R0 = this pixel's 5 coefficients, R1 = base of the second image's
// coefficient plane, (dx, dy) = current flow; out = (G11, G12, G22, h1, h2)
// five consecutive floats (one pixel's coefficients, 20-byte stride: 4-byte aligned only) as ONE 16-byte + one 4-byte load:
// lane-by-lane dword loads of such records touch every cache line five times (the frame kernel was bound by that, not by HBM)
typedef float fb_f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void fb_load5(const float* __restrict__ p, float (&v)[5]) {
  const fb_f4u a = *reinterpret_cast<const fb_f4u*>(p);
  v[0] = a[0], v[1] = a[1], v[2] = a[2], v[3] = a[3], v[4] = p[4];
}
__device__ __forceinline__ void fb_update_pixel(const float* __restrict__ R0p, const float* __restrict__ R1, float dx,
                                                float dy, int x, int y, int width, int height, float* m) {
  const size_t step1 = (size_t)width * 5;
  float R0[5];
  fb_load5(R0p, R0);
  float fx = __fadd_rn((float)x, dx), fy = __fadd_rn((float)y, dy);
  int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
  float r2, r3, r4, r5, r6;
  fx = __fsub_rn(fx, (float)x1);
  fy = __fsub_rn(fy, (float)y1);
  if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
    const float* ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
    float a00 = __fmul_rn(1.f - fx, 1.f - fy), a01 = __fmul_rn(fx, 1.f - fy);
    float a10 = __fmul_rn(1.f - fx, fy), a11 = __fmul_rn(fx, fy);
    float t00[5], t01[5], t10[5], t11[5];
    fb_load5(ptr, t00);
    fb_load5(ptr + 5, t01);
    fb_load5(ptr + step1, t10);
    fb_load5(ptr + step1 + 5, t11);
#define PV_BILIN(c) \
  __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a00, t00[c]), __fmul_rn(a01, t01[c])), __fmul_rn(a10, t10[c])), \
            __fmul_rn(a11, t11[c]))
    r2 = PV_BILIN(0);
    r3 = PV_BILIN(1);
    r4 = PV_BILIN(2);
    r5 = PV_BILIN(3);
    r6 = PV_BILIN(4);
#undef PV_BILIN
    r4 = __fmul_rn(__fadd_rn(R0[2], r4), 0.5f);
    r5 = __fmul_rn(__fadd_rn(R0[3], r5), 0.5f);
    r6 = __fmul_rn(__fadd_rn(R0[4], r6), 0.25f);
  } else {
    r2 = r3 = 0.f;
    r4 = R0[2];
    r5 = R0[3];
    r6 = __fmul_rn(R0[4], 0.5f);
  }
  r2 = __fmul_rn(__fsub_rn(R0[0], r2), 0.5f);
  r3 = __fmul_rn(__fsub_rn(R0[1], r3), 0.5f);
  r2 = __fadd_rn(r2, __fadd_rn(__fmul_rn(r4, dy), __fmul_rn(r6, dx)));
  r3 = __fadd_rn(r3, __fadd_rn(__fmul_rn(r6, dy), __fmul_rn(r5, dx)));
  if ((unsigned)(x - 5) >= (unsigned)(width - 10) || (unsigned)(y - 5) >= (unsigned)(height - 10)) {
    // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472} by distance from the edge, as selects (an indexed local array is a
    // memory load per lookup)
    auto border = [](int d) { return d < 2 ? 0.14f : (d < 5 ? 0.4472f : 1.f); };
    float scale = border(x);
    scale = __fmul_rn(scale, border(width - x - 1));
    scale = __fmul_rn(scale, border(y));
    scale = __fmul_rn(scale, border(height - y - 1));
    r2 = __fmul_rn(r2, scale); r3 = __fmul_rn(r3, scale); r4 = __fmul_rn(r4, scale);
    r5 = __fmul_rn(r5, scale); r6 = __fmul_rn(r6, scale);
  }
  m[0] = __fadd_rn(__fmul_rn(r4, r4), __fmul_rn(r6, r6));
  m[1] = __fmul_rn(__fadd_rn(r4, r5), r6);
  m[2] = __fadd_rn(__fmul_rn(r5, r5), __fmul_rn(r6, r6));
  m[3] = __fadd_rn(__fmul_rn(r4, r2), __fmul_rn(r6, r3));
  m[4] = __fadd_rn(__fmul_rn(r6, r2), __fmul_rn(r5, r3));
}

// one pixel of cv::resize(prevFlow -> (dw, dh), INTER_LINEAR) * (1 / pyr_scale): the expressions of fb_flow_upsample_kernel
typedef float fb_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fb_f32x2_t fb_upsampled_flow(const float* __restrict__ src, size_t p, int sh, int sw, int x, int y,
                                                        double inv_fx, double inv_fy, float mul) {
  float fx = (float)((x + 0.5) * inv_fx - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  if (sx < 0) { fx = 0; sx = 0; }
  if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
  float fy = (float)((y + 0.5) * inv_fy - 0.5);
  int sy = (int)floorf(fy);
  fy -= sy;
  if (sy < 0) { fy = 0; sy = 0; }
  if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
  const int sy1 = clampi_d(sy + 1, 0, sh - 1);
  const fb_f32x2_t* s0 = reinterpret_cast<const fb_f32x2_t*>(src) + (p * sh + sy) * sw;
  const fb_f32x2_t* s1 = reinterpret_cast<const fb_f32x2_t*>(src) + (p * sh + sy1) * sw;
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const fb_f32x2_t t00 = s0[sx], t10 = s1[sx];
  fb_f32x2_t r0 = t00, r1 = t10;
  if (sx + 1 < sw) {
    const fb_f32x2_t t01 = s0[sx + 1], t11 = s1[sx + 1];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      r0[c] = __fadd_rn(__fmul_rn(t00[c], a0), __fmul_rn(t01[c], a1));
      r1[c] = __fadd_rn(__fmul_rn(t10[c], a0), __fmul_rn(t11[c], a1));
    }
  }
  fb_f32x2_t o;
#pragma unroll
  for (int c = 0; c < 2; ++c) o[c] = __fmul_rn(__fadd_rn(__fmul_rn(r0[c], b0), __fmul_rn(r1[c], b1)), mul);
  return o;
}

// the same without control flow (the level kernel evaluates it between other work: a branch would split the live ranges around
// it): the right-hand taps are read at a clamped column and enter with weight 0 where the resize has none -- t00 * 1 + t01 * 0
// is t00 itself, so the values are those of fb_upsampled_flow
__device__ __forceinline__ fb_f32x2_t fb_upsampled_flow_nb(const float* __restrict__ src, size_t p, int sh, int sw, int x, int y,
                                                           double inv_fx, double inv_fy, float mul) {
  float fx = (float)((x + 0.5) * inv_fx - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  fx = (sx < 0 || sx >= sw - 1) ? 0.f : fx;
  sx = sx < 0 ? 0 : (sx >= sw - 1 ? sw - 1 : sx);
  float fy = (float)((y + 0.5) * inv_fy - 0.5);
  int sy = (int)floorf(fy);
  fy -= sy;
  fy = (sy < 0 || sy >= sh - 1) ? 0.f : fy;
  sy = sy < 0 ? 0 : (sy >= sh - 1 ? sh - 1 : sy);
  const int sy1 = min(sy + 1, sh - 1), sx1 = min(sx + 1, sw - 1);
  const fb_f32x2_t* s0 = reinterpret_cast<const fb_f32x2_t*>(src) + (p * sh + sy) * sw;
  const fb_f32x2_t* s1 = reinterpret_cast<const fb_f32x2_t*>(src) + (p * sh + sy1) * sw;
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const fb_f32x2_t t00 = s0[sx], t10 = s1[sx], t01 = s0[sx1], t11 = s1[sx1];
  const fb_f32x2_t r0 = t00 * a0 + t01 * a1, r1 = t10 * a0 + t11 * a1;
  return (r0 * b0 + r1 * b1) * mul;
}

// R: [n_img][lh][lw][5], the two images of pair p per fb_r_images_of; flow: [n_pairs][lh][lw][2];
// FLOW_SRC: 0 = the flow field is read; 1 = it is the previous (coarser) level's flow resized on the fly -- the resized
// field has no other reader before the blur + solve overwrites it, so the 16 bytes per pixel of writing and re-reading
// it (and the launch) are saved; 2 = zero (the coarsest level starts from no motion: no memset, no read).
// M: [n_pairs][lh][lw][5] (planar == 0) or [n_pairs][5][lh][lw] (planar != 0, what the fused tile kernel reads)
struct FbUpsample { int sh, sw; double inv_fx, inv_fy; float mul; };
template <int FLOW_SRC>
__global__ __launch_bounds__(256) void fb_update_matrices_kernel(const float* __restrict__ R, const float* __restrict__ flow,
                                                                  float* __restrict__ M, long long n_pairs, int height,
                                                                  int width, int planar, long long pairs_per_group,
                                                                  int chain_f, FbUpsample up) {
  const long long per_img = (long long)height * width;
  const long long total = n_pairs * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long p = i / per_img;
    int rem = (int)(i - p * per_img);
    int y = rem / width, x = rem - y * width;
    long long i0, i1;
    fb_r_images_of(p, pairs_per_group, chain_f, &i0, &i1);
    const float* R0 = R + (i0 * per_img + rem) * 5;
    const float* R1 = R + i1 * per_img * 5;
    float fdx = 0.f, fdy = 0.f;
    if constexpr (FLOW_SRC == 0) {
      fdx = flow[i * 2], fdy = flow[i * 2 + 1];
    } else if constexpr (FLOW_SRC == 1) {
      const fb_f32x2_t f = fb_upsampled_flow(flow, (size_t)p, up.sh, up.sw, x, y, up.inv_fx, up.inv_fy, up.mul);
      fdx = f[0], fdy = f[1];
    }
    float m[5];
    fb_update_pixel(R0, R1, fdx, fdy, x, y, width, height, m);
    if (planar) {
#pragma unroll
      for (int c = 0; c < 5; ++c) M[(p * 5 + c) * per_img + rem] = m[c];
    } else {
      float* d = M + i * 5;
      *reinterpret_cast<fb_f4u*>(d) = (fb_f4u){m[0], m[1], m[2], m[3]};
      d[4] = m[4];
    }
  }
}

// ---- window blur on the matrix cores ----------------------------------------------------------------------------------
// For images up to 64 x 64 the separable, border-replicated window blur is two small matrix products per channel,
//   U = X Gh^T (along x),  Out = Gv U (along y),   G[y][y'] = sum of the taps k with clamp(y + k) == y'
// (a banded 64 x 64 matrix that already contains the border replication, built per level by fb_window_matrix_kernel).
//
// The products run on the bf16 matrix cores at f32 accuracy: every f32 operand is split, by truncation, into three bf16
// terms x = h + m + l (8 + 8 + 8 mantissa bits; the two subtractions are exact), and a product keeps the six partial
// products down to 2^-16 (hh, hm, mh, hl, lh, mm) -- what is dropped is <= 2^-23 relative, the rounding of one f32
// operation.  v_mfma_f32_32x32x16_bf16 contracts 16 elements in 8 passes where the exact-f32 v_mfma_f32_32x32x2_f32 needs
// 8 instructions of 16 passes: six bf16 products cost 3/8 of the matrix-pipe time of one f32 product (the first version
// of this kernel ran on the f32 cores: 314 us per level-0 iteration at B = 32, 41 % of that pipe's peak).  Accumulation
// is f32 in both; only the summation order differs from the tap loop (parity bar: 1e-3 px, measured ~1e-6).
//
// One wave owns a (image pair, 32-column strip): the horizontal pass gives it U[all rows][its 32 columns] in accumulator
// registers, which ARE the B operand of the vertical pass (the contraction index simply follows the accumulator's row
// order, Gv is gathered accordingly once per launch) -- no exchange of U through LDS.  NB = 2 (images up to 64 x 64): two
// waves per pair, two pairs per workgroup; NB = 1 (up to 32 x 32, the coarse level): one wave per pair, four per workgroup.
// The channel image is staged zero-padded in LDS (double buffered, next channel's loads in flight under the MFMAs).
typedef float fb_v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void fb_window_matrix_kernel(float* __restrict__ G, int n, FbTaps kt, int mosaic) {
  // G[64][64]; rows / columns >= n stay zero.  mosaic (n <= 32): the n x n matrix twice on the diagonal, at 0 and at 32 --
  // the blur of a 64 x 64 image made of 2 x 2 independent tiles (fb_level_u_kernel<.., MOSAIC = true>)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * 64; i += gridDim.x * blockDim.x) {
    int y = i >> 6, yp = i & 63;
    const bool same_block = !mosaic || (y >> 5) == (yp >> 5);
    if (mosaic) y &= 31, yp &= 31;
    float s = 0.f;
    if (same_block && y < n && yp < n) {
      for (int k = -kt.n; k <= kt.n; ++k) {
        int yy = y + k;
        yy = yy < 0 ? 0 : (yy > n - 1 ? n - 1 : yy);
        if (yy == yp) s += kt.k[k < 0 ? -k : k];
      }
    }
    G[i] = s;
  }
}

__device__ __forceinline__ int fb_acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

struct FbSplit3 { bf16x8 h, m, l; };

// x[0..7] -> three bf16x8 with x = h + m + l up to 2^-24 |x| (truncation split; element i in bf16 slot i)
__device__ __forceinline__ FbSplit3 fb_split3(const float (&x)[8]) {
  u32x4 hw, mw, lw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t a = __builtin_bit_cast(uint32_t, x[2 * j]), b = __builtin_bit_cast(uint32_t, x[2 * j + 1]);
    hw[j] = __builtin_amdgcn_perm(b, a, 0x07060302u);                 // (b & 0xffff0000) | (a >> 16)
    const float ra = x[2 * j] - __builtin_bit_cast(float, a & 0xffff0000u);
    const float rb = x[2 * j + 1] - __builtin_bit_cast(float, b & 0xffff0000u);
    const uint32_t a1 = __builtin_bit_cast(uint32_t, ra), b1 = __builtin_bit_cast(uint32_t, rb);
    mw[j] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
    const float sa = ra - __builtin_bit_cast(float, a1 & 0xffff0000u);
    const float sb = rb - __builtin_bit_cast(float, b1 & 0xffff0000u);
    lw[j] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sb), __builtin_bit_cast(uint32_t, sa), 0x07060302u);
  }
  FbSplit3 o;
  o.h = __builtin_bit_cast(bf16x8, hw);
  o.m = __builtin_bit_cast(bf16x8, mw);
  o.l = __builtin_bit_cast(bf16x8, lw);
  return o;
}

// acc += A B with both operands split: the six partial products above 2^-24, smallest first
__device__ __forceinline__ fb_v16f fb_mfma3(const FbSplit3& a, const FbSplit3& b, fb_v16f acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
  return acc;
}

template <int NB>   // tile edge = 32 NB; NB waves per image pair, 4 / NB pairs per workgroup
// (NB == 1: 80 accumulator registers and 47 KB of LDS -- two workgroups per CU)
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void fb_tile_mfma_kernel(const float* __restrict__ Min, const float* __restrict__ Gv,
                                                            const float* __restrict__ Gh, float* __restrict__ flow,
                                                            int height, int width, long long n_pairs) {
  constexpr int T = 32 * NB, XS = T + 8, PLANE = T * XS, PPW = 4 / NB, KS = 2 * NB;   // KS contraction steps of 16
  constexpr int NE = T * T / (64 * NB);      // staged elements per lane and channel
  // NB > 1: the channel image is split ONCE, while it is staged: three bf16 planes (h, m, l) per buffer -- both strips of a
  // pair read the same image as their A operand, so splitting at the read would do the work twice (level-0 iteration
  // 238 -> 218 us).  NB == 1: one wave per pair reads its image once, so it stays f32 in LDS (a third of the LDS bytes) and is
  // split at the read (planes 1, 2 unused: 2 bf16 = 1 float per element of plane 0 ... the f32 image takes planes 0 and 1).
  constexpr bool PRESPLIT = NB > 1;
  __shared__ __attribute__((aligned(16))) uint16_t Xs[PPW][2][PRESPLIT ? 3 : 2][PLANE];
  // Gv split operands, lane-major: the same for every wave (they depend on the output row = lane, not on the strip)
  __shared__ u32x4 GvS[NB * KS * 3][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int pw = wave / NB, strip = wave % NB;
  const long long per_img = (long long)height * width;

  // ---- constant operands, split once: Gh^T as B operand of the horizontal pass (k = x' in natural order), Gv as A operand of
  // the vertical pass (k-slot i of lane-half h in step (blk, s) = accumulator row fb_acc_row(8 s + i, h) of row block blk)
  FbSplit3 gh[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gh[(32 * strip + col) * 64 + 16 * ks + 8 * half + i];
    gh[ks] = fb_split3(t);
  }
  for (int j = wave; j < NB * KS; j += 4) {      // (mbo, ks) pairs dealt to the four waves
    const int mbo = j / KS, ks = j - mbo * KS;
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gv[(32 * mbo + col) * 64 + 32 * (ks >> 1) + fb_acc_row(8 * (ks & 1) + i, half)];
    const FbSplit3 g = fb_split3(t);
    GvS[j * 3 + 0][lane] = __builtin_bit_cast(u32x4, g.h);
    GvS[j * 3 + 1][lane] = __builtin_bit_cast(u32x4, g.m);
    GvS[j * 3 + 2][lane] = __builtin_bit_cast(u32x4, g.l);
  }
  __syncthreads();

  const bool vec = (width & 3) == 0 && ((uintptr_t)Min & 15) == 0;
  const long long groups = (n_pairs + PPW - 1) / PPW;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long p = grp * PPW + pw;
    const bool p_ok = p < n_pairs;
    fb_v16f res[5][NB];
    float stg[NE];   // one channel image in flight: the loads of channel c + 1 are issued before the MFMAs of channel c
    // staging element e of this lane: vec: quad q = strip*64 + lane + 64 NB (e / 4), column 4 (q % (T/4)) + e % 4;
    // scalar: index i = strip*64 + lane + 64 NB e
    auto load_channel = [&](int c) {
      const float* src = Min + ((p_ok ? p : 0) * 5 + c) * per_img;
      if (vec) {
#pragma unroll
        for (int e = 0; e < NE / 4; ++e) {
          const int q = strip * 64 + lane + 64 * NB * e;      // quad of 4 consecutive columns
          const int y = q / (T / 4), x = (q - y * (T / 4)) * 4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (p_ok && y < height && x < width) v = *reinterpret_cast<const f32x4*>(src + (long long)y * width + x);
          stg[4 * e] = v[0]; stg[4 * e + 1] = v[1]; stg[4 * e + 2] = v[2]; stg[4 * e + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int i = strip * 64 + lane + 64 * NB * e;
          const int y = i / T, x = i - y * T;
          stg[e] = (p_ok && y < height && x < width) ? src[(long long)y * width + x] : 0.f;
        }
      }
    };
    load_channel(0);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      // ---- channel c (zero padded to the tile), split into its three bf16 planes, into this pair's LDS image; the next
      // channel's loads start right away -----------------------------------------------------------------------------------
      uint16_t* Xc = &Xs[pw][c & 1][0][0];
      float* Xf = reinterpret_cast<float*>(Xc);     // !PRESPLIT: f32 image, row stride XS floats
      if constexpr (!PRESPLIT) {
        if (vec) {
#pragma unroll
          for (int e = 0; e < NE / 4; ++e) {
            const int q = strip * 64 + lane + 64 * NB * e;
            const int y = q / (T / 4), x = (q - y * (T / 4)) * 4;
            *reinterpret_cast<f32x4*>(Xf + y * XS + x) = (f32x4){stg[4 * e], stg[4 * e + 1], stg[4 * e + 2], stg[4 * e + 3]};
          }
        } else {
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int i = strip * 64 + lane + 64 * NB * e;
            const int y = i / T;
            Xf[y * XS + (i - y * T)] = stg[e];
          }
        }
      } else
#pragma unroll
      for (int e8 = 0; e8 < NE / 8; ++e8) {
        const float t[8] = {stg[8 * e8], stg[8 * e8 + 1], stg[8 * e8 + 2], stg[8 * e8 + 3],
                            stg[8 * e8 + 4], stg[8 * e8 + 5], stg[8 * e8 + 6], stg[8 * e8 + 7]};
        const FbSplit3 sp = fb_split3(t);
        const u32x4 hw = __builtin_bit_cast(u32x4, sp.h), mw = __builtin_bit_cast(u32x4, sp.m), lw = __builtin_bit_cast(u32x4, sp.l);
        if (vec) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {        // two quads of 4 consecutive columns
            const int q = strip * 64 + lane + 64 * NB * (2 * e8 + u);
            const int y = q / (T / 4), x = (q - y * (T / 4)) * 4;
            *reinterpret_cast<u32x2*>(Xc + y * XS + x) = (u32x2){hw[2 * u], hw[2 * u + 1]};
            *reinterpret_cast<u32x2*>(Xc + PLANE + y * XS + x) = (u32x2){mw[2 * u], mw[2 * u + 1]};
            *reinterpret_cast<u32x2*>(Xc + 2 * PLANE + y * XS + x) = (u32x2){lw[2 * u], lw[2 * u + 1]};
          }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = strip * 64 + lane + 64 * NB * (8 * e8 + u);
            const int y = i / T, x = i - y * T;
            const int sh = (u & 1) * 16;
            Xc[y * XS + x] = (uint16_t)(hw[u >> 1] >> sh);
            Xc[PLANE + y * XS + x] = (uint16_t)(mw[u >> 1] >> sh);
            Xc[2 * PLANE + y * XS + x] = (uint16_t)(lw[u >> 1] >> sh);
          }
        }
      }
      if (NB > 1) __syncthreads();   // the image is staged by both waves of the pair (the buffer of channel c - 1 may still be read)
      else __builtin_amdgcn_wave_barrier();
      if (c < 4) load_channel(c + 1);
      // ---- horizontal: U[y'][x] = sum_x' X[y'][x'] Gh[x][x'], all row blocks, this wave's 32 columns ----------------------
      fb_v16f u[NB];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) u[mb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          FbSplit3 a;
          if constexpr (PRESPLIT) {
            const uint16_t* xa = Xc + (32 * mb + col) * XS + 16 * ks + 8 * half;
            a.h = *reinterpret_cast<const bf16x8*>(xa);
            a.m = *reinterpret_cast<const bf16x8*>(xa + PLANE);
            a.l = *reinterpret_cast<const bf16x8*>(xa + 2 * PLANE);
          } else {
            const float* xa = Xf + (32 * mb + col) * XS + 16 * ks + 8 * half;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xa), x1 = *reinterpret_cast<const f32x4*>(xa + 4);
            const float t[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            a = fb_split3(t);
          }
          u[mb] = fb_mfma3(a, gh[ks], u[mb]);
        }
      }
      // ---- vertical: Out[y][x] = sum_y' Gv[y][y'] U[y'][x]: the accumulators of the horizontal pass are the B operand -------
#pragma unroll
      for (int mbo = 0; mbo < NB; ++mbo)
#pragma unroll
        for (int r = 0; r < 16; ++r) res[c][mbo][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = u[ks >> 1][8 * (ks & 1) + i];
        const FbSplit3 b = fb_split3(t);
#pragma unroll
        for (int mbo = 0; mbo < NB; ++mbo) {
          FbSplit3 g;
          g.h = __builtin_bit_cast(bf16x8, GvS[(mbo * KS + ks) * 3 + 0][lane]);
          g.m = __builtin_bit_cast(bf16x8, GvS[(mbo * KS + ks) * 3 + 1][lane]);
          g.l = __builtin_bit_cast(bf16x8, GvS[(mbo * KS + ks) * 3 + 2][lane]);
          res[c][mbo] = fb_mfma3(g, b, res[c][mbo]);
        }
      }
    }
    if (NB > 1) __syncthreads();   // both image buffers free before the next pair's channels 0 / 1 are staged
    // ---- 2x2 solve; accumulator register r = row y, lane = column x: coalesced flow rows ---------------------------------
    if (p_ok) {
#pragma unroll
      for (int mbo = 0; mbo < NB; ++mbo)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int y = 32 * mbo + fb_acc_row(r, half), x = 32 * strip + col;
          if (y < height && x < width) {
            double g11 = res[0][mbo][r], g12 = res[1][mbo][r], g22 = res[2][mbo][r], h1 = res[3][mbo][r], h2 = res[4][mbo][r];
            double det = __dadd_rn(__dsub_rn(__dmul_rn(g11, g22), __dmul_rn(g12, g12)), 1e-3);
            double idet = __ddiv_rn(1.0, det);
            const float fxv = (float)__dmul_rn(__dsub_rn(__dmul_rn(g11, h2), __dmul_rn(g12, h1)), idet);
            const float fyv = (float)__dmul_rn(__dsub_rn(__dmul_rn(g22, h1), __dmul_rn(g12, h2)), idet);
            float* fl = flow + (p * per_img + (long long)y * width + x) * 2;
            fl[0] = fxv;
            fl[1] = fyv;
          }
        }
    }
  }
}

// ---- 64 x 64 tiles, second form: one wave per QUADRANT of the output, two workgroups per CU ---------------------------
// fb_tile_mfma_kernel<2> gives a wave a 64-row x 32-column strip: 5 channels x 2 row blocks of results = 160 accumulator
// registers, which pins the kernel at one wave per SIMD -- and its counters read: matrix pipe busy 23 %, vector ALU 38 %,
// parked 38 % of the wave's cycles (nothing covers a wave's splits, LDS round trips and barriers).  Here a wave owns ONE
// 32 x 32 quadrant (strip, output row block mbo): the vertical pass needs U of both row blocks of its strip, so the
// horizontal pass is computed by both waves of a strip (48 + 24 instead of 48 + 48 MFMA triples per channel and wave: half as
// many again matrix instructions per pair), but the results are 5 x 16 registers and the workgroup (= one pair, 79 KB of
// LDS) fits a CU TWICE: two waves per SIMD, each running while the other splits operands or waits.  Staging is shared by the
// four waves (16 values per lane and channel, split once into the three bf16 planes), double buffered over the channels.
// Same arithmetic as fb_tile_mfma_kernel<2> per output element (the products and their order inside fb_mfma3; the
// contraction order of both passes), so the flows are bit-identical to it.
__global__ __launch_bounds__(256, 2) void fb_tile_mfma_q_kernel(const float* __restrict__ Min, const float* __restrict__ Gv,
                                                                const float* __restrict__ Gh, float* __restrict__ flow,
                                                                int height, int width, long long n_pairs) {
  constexpr int T = 64, XS = T + 8, PLANE = T * XS, KS = 4;
  constexpr int NE = T * T / 256;            // staged elements per lane and channel
  __shared__ __attribute__((aligned(16))) uint16_t Xs[2][3][PLANE];
  __shared__ u32x4 GvS[2 * KS * 3][64];      // [mbo][ks][plane], lane-major
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int strip = wave & 1, mbo = wave >> 1;
  const long long per_img = (long long)height * width;

  FbSplit3 gh[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gh[(32 * strip + col) * 64 + 16 * ks + 8 * half + i];
    gh[ks] = fb_split3(t);
  }
  for (int j = wave; j < 2 * KS; j += 4) {      // (mbo, ks) pairs dealt to the four waves
    const int mb = j / KS, ks = j - mb * KS;
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gv[(32 * mb + col) * 64 + 32 * (ks >> 1) + fb_acc_row(8 * (ks & 1) + i, half)];
    const FbSplit3 g = fb_split3(t);
    GvS[j * 3 + 0][lane] = __builtin_bit_cast(u32x4, g.h);
    GvS[j * 3 + 1][lane] = __builtin_bit_cast(u32x4, g.m);
    GvS[j * 3 + 2][lane] = __builtin_bit_cast(u32x4, g.l);
  }
  __syncthreads();

  for (long long p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    fb_v16f res[5];
    float stg[NE];
    // staging element e of this thread: quad q = tid + 256 (e / 4) of 4 consecutive columns, column 4 (q % 16) + e % 4
    auto load_channel = [&](int c) {
      const float* src = Min + (p * 5 + c) * per_img;
#pragma unroll
      for (int e = 0; e < NE / 4; ++e) {
        const int q = tid + 256 * e;
        const int y = q >> 4, x = (q & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y < height && x < width) v = *reinterpret_cast<const f32x4*>(src + (long long)y * width + x);
        stg[4 * e] = v[0]; stg[4 * e + 1] = v[1]; stg[4 * e + 2] = v[2]; stg[4 * e + 3] = v[3];
      }
    };
    load_channel(0);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      uint16_t* Xc = &Xs[c & 1][0][0];
#pragma unroll
      for (int e8 = 0; e8 < NE / 8; ++e8) {
        const float t[8] = {stg[8 * e8], stg[8 * e8 + 1], stg[8 * e8 + 2], stg[8 * e8 + 3],
                            stg[8 * e8 + 4], stg[8 * e8 + 5], stg[8 * e8 + 6], stg[8 * e8 + 7]};
        const FbSplit3 sp = fb_split3(t);
        const u32x4 hw = __builtin_bit_cast(u32x4, sp.h), mw = __builtin_bit_cast(u32x4, sp.m), lw = __builtin_bit_cast(u32x4, sp.l);
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {        // two quads of 4 consecutive columns
          const int q = tid + 256 * (2 * e8 + u2);
          const int y = q >> 4, x = (q & 15) * 4;
          *reinterpret_cast<u32x2*>(Xc + y * XS + x) = (u32x2){hw[2 * u2], hw[2 * u2 + 1]};
          *reinterpret_cast<u32x2*>(Xc + PLANE + y * XS + x) = (u32x2){mw[2 * u2], mw[2 * u2 + 1]};
          *reinterpret_cast<u32x2*>(Xc + 2 * PLANE + y * XS + x) = (u32x2){lw[2 * u2], lw[2 * u2 + 1]};
        }
      }
      __syncthreads();   // the image is staged by all four waves (the buffer of channel c - 1 may still be read)
      if (c < 4) load_channel(c + 1);
      // ---- horizontal: U[y'][x] = sum_x' X[y'][x'] Gh[x][x'], both row blocks, this wave's 32 columns ----------------------
      fb_v16f u[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) u[mb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          FbSplit3 a;
          const uint16_t* xa = Xc + (32 * mb + col) * XS + 16 * ks + 8 * half;
          a.h = *reinterpret_cast<const bf16x8*>(xa);
          a.m = *reinterpret_cast<const bf16x8*>(xa + PLANE);
          a.l = *reinterpret_cast<const bf16x8*>(xa + 2 * PLANE);
          u[mb] = fb_mfma3(a, gh[ks], u[mb]);
        }
      }
      // ---- vertical, this wave's output row block: Out[y][x] = sum_y' Gv[y][y'] U[y'][x] -----------------------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) res[c][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = u[ks >> 1][8 * (ks & 1) + i];
        const FbSplit3 b = fb_split3(t);
        FbSplit3 g;
        g.h = __builtin_bit_cast(bf16x8, GvS[(mbo * KS + ks) * 3 + 0][lane]);
        g.m = __builtin_bit_cast(bf16x8, GvS[(mbo * KS + ks) * 3 + 1][lane]);
        g.l = __builtin_bit_cast(bf16x8, GvS[(mbo * KS + ks) * 3 + 2][lane]);
        res[c] = fb_mfma3(g, b, res[c]);
      }
    }
    __syncthreads();   // both image buffers free before the next pair's channels 0 / 1 are staged
    // ---- 2x2 solve; accumulator register r = row y, lane = column x: coalesced flow rows ---------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int y = 32 * mbo + fb_acc_row(r, half), x = 32 * strip + col;
      if (y < height && x < width) {
        double g11 = res[0][r], g12 = res[1][r], g22 = res[2][r], h1 = res[3][r], h2 = res[4][r];
        double det = __dadd_rn(__dsub_rn(__dmul_rn(g11, g22), __dmul_rn(g12, g12)), 1e-3);
        double idet = __ddiv_rn(1.0, det);
        const float fxv = (float)__dmul_rn(__dsub_rn(__dmul_rn(g11, h2), __dmul_rn(g12, h1)), idet);
        const float fyv = (float)__dmul_rn(__dsub_rn(__dmul_rn(g22, h1), __dmul_rn(g12, h2)), idet);
        float* fl = flow + (p * per_img + (long long)y * width + x) * 2;
        fl[0] = fxv;
        fl[1] = fyv;
      }
    }
  }
}

typedef int fb_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char* fb_lds_ptr;
// ---- building blocks of the level kernel (fb_level_u_kernel) and of the matrix-core PolyExp (round 4) ------------------------
// What the counters and stamps of round 3's fused iteration said (profiles/r03): its R1 gathers hit the LDS four ways (20-byte
// records, a wave's four rows on the same banks), and on the vector ALU the three-way bf16 split (5.5 instructions per
// element) and the f64 solve weighed as much as UpdateMatrices itself.  Hence:
//   * R is PAIR-PLANAR in memory (per image: (c0, c1) as float2 [h][w], (c2, c3) as float2 [h][w], c4 as float [h][w]; written
//     so by the PolyExp kernels) and in LDS (the same three planes with a fixed row stride of 64 pixels whatever the level's
//     size; MOSAIC: the four 32 x 32 tiles side by side in the same planes).  A lane owns one COLUMN and walks 8 rows, so the
//     64 lanes of a gather read consecutive pixels of a row (displaced by the flow): no bank conflicts for a smooth flow
//     field, every neighbour is a constant offset from one address, and a channel pair arrives as the register pair the
//     packed f32 instructions take (12 LDS reads per pixel instead of 20, no register shuffling).
//   * matrix products run on the F16 matrix cores with TWO-term operands: x = h + l, h = rne_f16(x), l = rne_f16(x - h)
//     carries 22 significant bits, so three products (hh, hl, lh) reach 2^-22 where the bf16 split needs six -- half the
//     matrix work and 2.5 instead of 5.5 vector instructions per element split.  f16 has a narrow range, so every operand
//     is scaled by an exact power of two first: the window matrices by 2^15 (taps 4e-3 .. 0.5 -> 130 .. 16384), M by a
//     per-pair factor s = 2^(15 - e) from the pair's largest |M| (one extra barrier per stage) so that |M s| < 2^15;
//     residuals land in f16's subnormals at worst, which v_mfma_f32_32x32x16_f16 honours (tools/probes/mfma_f16_denorm.hip).
//     The blurred sums carry the factor k = 2^15 s into the solve, which is homogeneous but for the regulariser:
//     flow = num k^2 / (det k^2 + 1e-3 k^2).
//   * an image is handed over TRANSPOSED (X^T[x][y'], a lane's 8 rows are 16 contiguous bytes per plane: two ds_write_b128
//     per channel), so the products are V^T = X^T Gv^T, Out^T = Gh V^T -- the first product's accumulators are the second
//     one's operand without an exchange through LDS.
//   * the 2 x 2 solve in f32 with error-free products (Kahan's ad - bc with fma) and one refined reciprocal per pixel, where
//     the f64 form cost 3 200 cycles per pair at half rate.
typedef _Float16 fb_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fb_f16x2 __attribute__((ext_vector_type(2)));
typedef float fb_f2 __attribute__((ext_vector_type(2)));
struct FbSplit2 { fb_f16x8 h, l; };
constexpr float FB_G_SCALE = 32768.f, FB_G_UNSCALE = 1.f / 32768.f;

// x[0..7] * scale (an exact power of two) -> h + l with h = rne_f16, l = rne_f16(x - h) (x - h is exact in f32)
__device__ __forceinline__ FbSplit2 fb_split2(const float (&x)[8], float scale) {
  u32x4 hw, lw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const fb_f2 v = (fb_f2){x[2 * j], x[2 * j + 1]} * scale;
    const fb_f16x2 h = __builtin_convertvector(v, fb_f16x2);
    const fb_f2 r = v - __builtin_convertvector(h, fb_f2);
    const fb_f16x2 l = __builtin_convertvector(r, fb_f16x2);
    hw[j] = __builtin_bit_cast(uint32_t, h);
    lw[j] = __builtin_bit_cast(uint32_t, l);
  }
  FbSplit2 o;
  o.h = __builtin_bit_cast(fb_f16x8, hw);
  o.l = __builtin_bit_cast(fb_f16x8, lw);
  return o;
}
// acc += A B on two-term operands: the three partial products above 2^-22, smallest first
__device__ __forceinline__ fb_v16f fb_mfma2(const FbSplit2& a, const FbSplit2& b, fb_v16f acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc, 0, 0, 0);
  return acc;
}
// ad - bc with one rounding error of the result (Kahan): w = bc rounded, e = w - bc exactly, f = ad - w rounded once
__device__ __forceinline__ float fb_det2(float a, float d, float b, float c) {
  const float w = __fmul_rn(b, c);
  const float e = __builtin_fmaf(-b, c, w);
  const float f = __builtin_fmaf(a, d, -w);
  return __fadd_rn(f, e);
}
// FarnebackUpdateMatrices of one pixel in two halves -- the expressions and the rounding order of fb_update_pixel, channel
// pairs on the packed f32 instructions.  G: the bilinear blend of the four R1 neighbours (r2..r6 of the reference before R0
// enters); F: everything after it.
struct FbBlend { fb_f2 b01, b23; float b4; };      // b4 = NaN: the displaced point lies outside the image
struct FbTapsP { fb_f2 u00, u01, u10, u11, v00, v01, v10, v11; float w00, w01, w10, w11, fx, fy; bool inside; };
struct FbImgs { long long a, b, c, d; };      // first coefficient images of a unit's pairs (one pair: a)
__device__ __forceinline__ void fb_update_pixel_finish2(const fb_f2 R01, const fb_f2 R23, const float R4, const FbBlend& g, float dx,
                                                         float dy, float scale, float (&m)[5]) {
  const bool inside = g.b4 == g.b4;
  fb_f2 r45 = (R23 + g.b23) * 0.5f;
  float r6 = __fmul_rn(__fadd_rn(R4, g.b4), 0.25f);
  fb_f2 r23 = inside ? g.b01 : (fb_f2){0.f, 0.f};
  r45 = inside ? r45 : R23;
  r6 = inside ? r6 : __fmul_rn(R4, 0.5f);
  r23 = (R01 - r23) * 0.5f;
  // r2 += r4 dy + r6 dx;  r3 += r6 dy + r5 dx
  r23 = r23 + ((fb_f2){r45[0], r6} * dy + (fb_f2){r6, r45[1]} * dx);
  r23 = r23 * scale;
  r45 = r45 * scale;
  r6 = __fmul_rn(r6, scale);
  const fb_f2 sq = r45 * r45;
  const float r66 = __fmul_rn(r6, r6);
  m[0] = __fadd_rn(sq[0], r66);
  m[2] = __fadd_rn(sq[1], r66);
  m[1] = __fmul_rn(__fadd_rn(r45[0], r45[1]), r6);
  const fb_f2 h = (fb_f2){r45[0], r6} * r23[0] + (fb_f2){r6, r45[1]} * r23[1];      // (r4 r2 + r6 r3, r6 r2 + r5 r3)
  m[3] = h[0];
  m[4] = h[1];
}

#ifdef PV_DIAG_STAMPS
__device__ unsigned long long fb_iter_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif
// ---- one launch per pyramid LEVEL: every iteration of every pair, eight UNIFORM waves ------------------------------------------
// (The round's first form -- four multiplying + eight producing waves, 768 threads -- showed in its stamps that the two kinds
// of waves take turns inside a unit anyway (iteration it + 1 needs iteration it's flow: a stage is a serial chain), and in its
// counters that at three waves per SIMD (168 registers) a unit's R0 and the flow between two iterations do not fit in
// registers: R0 was read again from memory for every iteration (0.9 of that launch's 1.9 GB) and the flow went through memory
// and back.  This kernel evaluates the same expressions in the same order: its flows were bit-identical to that form's, which
// was removed at the end of the round; profiles/r04/split_wave_kernel_* are its measurements.)
// A 512-thread workgroup (two waves per SIMD, 256 registers) does every phase with ALL eight waves:
//   P  UpdateMatrices: wave w owns rows 8 w .. 8 w + 7, a lane one column (G: 12 LDS reads + the bilinear blend per pixel, F: the
//      rest, per pixel right behind it); the unit's R0 stays in 40 registers through all its iterations, R1 in LDS;
//   M  the window blur: wave (strip, mbo, g) multiplies for block (strip, mbo) the channels of group g (g = 0: 0, 1, 2;
//      g = 1: 3, 4) -- the two waves of a SIMD cover each other's LDS reads and operand splits.  Channels are handed over two
//      at a time through a double-buffered LDS image (rounds (0, 3), (1, 4), (2, -)); both window operands live in registers;
//   S  the two groups of a block swap half of their blurred channels through LDS mailboxes and solve half of the block's pixels
//      each; the flow goes to an LDS image (row-major, rows of 528 bytes) from where every lane takes its 8 pixels for the next
//      iteration: the flow between two iterations never leaves the CU.  The unit's last iteration stores it to memory instead.
// During the unit's last M phase the eight waves bring the next unit's R1 through the registers R0 no longer needs, then its
// R0, and its source flow into the free flow image.  Per stage: six barriers, no memory traffic but the flow store of the last
// iteration and 160 KB per UNIT of R.
template <int FLOW_INIT, bool MOSAIC>
__global__ __launch_bounds__(512) void fb_level_u_kernel(const float* __restrict__ R, const float* __restrict__ flow_prev, float* flow,
                                                         const float* __restrict__ Gv, const float* __restrict__ Gh, int height,
                                                         int width, long long n_pairs, long long pairs_per_group, int chain_f,
                                                         FbUpsample up, int iterations) {
  constexpr int T = 64, XS = T + 8, PLANE = T * XS, KS = 4;
  constexpr int FROW = 528;      // bytes per row of the flow image in LDS (64 float2 + 16: 16-byte rows on rotating banks)
  // R1 of the unit in hand: (c0, c1) float2 [64][64] | (c2, c3) float2 [64][64] | c4 float [64][64]
  __shared__ __attribute__((aligned(16))) float R1s[5 * T * T];
  // X^T of two channels per round, (h, l) f16 planes: four images X00 X01 | X10 X11 = [round parity][group 0's channel, group
  // 1's].  Once the products have read them the same bytes are the flow image (the first 33 KB) and, behind it, the mailboxes
  // through which the two groups of a block swap blurred channels for the solve (X00, still read in round 2, lies below them)
  constexpr int XIMG = 2 * PLANE * 2;      // bytes of one channel's X^T image (h and l planes)
  constexpr int MAIL0 = 64 * FROW;         // the mailboxes start behind the flow image: 4 blocks x 10 slots x 1 KB
  __shared__ __attribute__((aligned(16))) unsigned char XsB[MAIL0 + 4 * 10 * 1024];
  static_assert(MAIL0 >= XIMG && MAIL0 + 4 * 10 * 1024 >= 4 * XIMG, "LDS image plan");
  __shared__ __attribute__((aligned(16))) float pmax[16];                 // largest |M| per (wave, lane half)
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Everything derived from the lane number is re-derived at every phase boundary from an opaque copy (rederive below):
  // otherwise each phase's addresses and per-lane constants -- some fifty registers -- stay alive through all the others
  int lane_src = threadIdx.x & 63;
  int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
  const int lpx = height * width;
  constexpr int NP = MOSAIC ? 4 : 1;
  constexpr uint32_t NOT_THERE = 0x80000000u;      // buffer offset outside every descriptor below: reads zeros
  const long long n_units = (n_pairs + NP - 1) / NP;
  const long long p_lo = n_units * blockIdx.x / gridDim.x, p_hi = n_units * (blockIdx.x + 1) / gridDim.x;
  const int n_it = iterations;
  // multiplying role: block (strip, mbo) of the 64 x 64 image (MOSAIC: tile (ty, tx) = pair 4 u + 2 strip + mbo), group g
  const int sm = pw & 3, strip = sm & 1, mbo = sm >> 1, grp = pw >> 2;
  // producing role: rows 8 pw .. + 7, column = lane (MOSAIC: tile row pw >> 2, tile column lane >> 5)
  const int m_ty = pw >> 2;
  int m_tx = lane >> 5;
  int xl = MOSAIC ? (lane & 31) : lane;                       // column inside the pair's image
  const int yl0 = MOSAIC ? 8 * (pw & 3) : 8 * pw;             // first row inside the pair's image
  int lds_org = MOSAIC ? (32 * m_ty) * 64 + 32 * m_tx : 0;            // the tile's origin in an LDS plane (pixels)
  bool col_ok = xl < width;

  // window operands, both in registers: Gv^T as the B operand of the first product (n = y, this lane's row of the strip; k =
  // y' in natural order); Gh as the A operand of the second (k-slot i of lane half h in step ks = accumulator row
  // fb_acc_row(8 (ks & 1) + i, h) of row block ks >> 1 of the first product)
  FbSplit2 gv[KS], gh[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gv[(32 * strip + col) * 64 + 16 * ks + 8 * half + i];
    gv[ks] = fb_split2(t, FB_G_SCALE);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gh[(32 * mbo + col) * 64 + 32 * (ks >> 1) + fb_acc_row(8 * (ks & 1) + i, half)];
    gh[ks] = fb_split2(t, FB_G_SCALE);
  }
  // exponent e of a unit's (MOSAIC: of tile (ty, tx)'s) largest |M| = f 2^e, 0.5 <= f < 1, clamped so that every power of
  // two formed from it is a normal f32; M 2^(15 - e) then lies below 2^15
  auto unit_exp = [&](int ty, int tx) __attribute__((always_inline)) -> int {
    const f32x4* pm4 = reinterpret_cast<const f32x4*>(pmax);
    float mx;
    if constexpr (MOSAIC) {
      const f32x4 a = pm4[2 * ty], b = pm4[2 * ty + 1];
      mx = tx ? fmaxf(fmaxf(a[1], a[3]), fmaxf(b[1], b[3])) : fmaxf(fmaxf(a[0], a[2]), fmaxf(b[0], b[2]));
    } else {
      const f32x4 a = pm4[0], b = pm4[1], c = pm4[2], d = pm4[3];
      const f32x4 m4 = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
      mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    }
    const int e = __builtin_amdgcn_frexp_expf(mx);
    return e < -25 ? -25 : (e > 100 ? 100 : e);
  };

  // ---- the unit -> coefficient image bookkeeping --------------------------------------------------------------------------------
  long long i0_run, i1_unused;
  fb_r_images_of(p_lo * NP, pairs_per_group, chain_f, &i0_run, &i1_unused);
  long long q_run = chain_f > 0 ? (p_lo * NP) % pairs_per_group : 0;
  const long long img_step = chain_f > 0 ? 1 : 2, img_jump = chain_f > 0 ? chain_f - pairs_per_group : 0;
  auto next_image = [&]() -> long long {
    const long long r = i0_run;
    q_run += 1;
    const long long wrap = (long long)((int)(chain_f > 0) & (int)(q_run == pairs_per_group));      // the next stack's first frame
    i0_run += img_step + wrap * img_jump;
    q_run -= wrap * q_run;
    return r;
  };
  auto take_unit = [&](FbImgs& i0) {
    i0.a = next_image();
    if constexpr (MOSAIC) i0.b = next_image(), i0.c = next_image(), i0.d = next_image();
  };
  // R1 of `unit` (second coefficient images = i0 + 1) through registers: 80 chunks of 1 KB (chunks 0..31 = plane (c0, c1), two
  // rows of 64 float2 each; 32..63 = plane (c2, c3); 64..79 = plane c4, four rows of 64 floats each), wave pw takes chunks pw,
  // pw + 8, ...  Lane j brings 16 bytes: in the pair planes row (j >> 5) of the chunk, pixels 2 (j & 31), + 1; in the c4 plane
  // row (j >> 4), pixels 4 (j & 15) .. + 3.  A unit beyond the range, a missing pair, rows or columns beyond the image read zeros
  int dp_x = MOSAIC ? 2 * ((lane & 31) & 15) : 2 * (lane & 31), dp_tx = (lane & 31) >> 4;
  int ds_x = MOSAIC ? 4 * ((lane & 15) & 7) : 4 * (lane & 15), ds_tx = (lane & 15) >> 3;
  uint32_t dma_lane_p = (uint32_t)((lane >> 5) * width + dp_x) * 8, dma_lane_s = (uint32_t)((lane >> 4) * width + ds_x) * 4;
  auto r1_fetch = [&](const FbImgs& i0, long long unit, u32x4 (&buf)[10]) __attribute__((always_inline)) {
    const bool unit_ok = unit < p_hi;
#pragma unroll
    for (int n = 0; n < 10; ++n) {
      const int k = pw + 8 * n;
      const bool pair_plane = n < 8;                              // compile-time per n
      const int kk = pair_plane ? (k & 31) : (k - 64);            // chunk inside its plane
      const int rows = pair_plane ? 2 : 4;                        // LDS rows per chunk
      const int plane_off = pair_plane ? (n < 4 ? 0 : lpx * 8) : lpx * 16;      // bytes from the image's start
      const int px_bytes = pair_plane ? 8 : 4;
      const int d_x = pair_plane ? dp_x : ds_x;
      const uint32_t lane_off = pair_plane ? dma_lane_p : dma_lane_s;
      if constexpr (MOSAIC) {
        const int r0w = rows * kk, ty = r0w >> 5, yb = r0w & 31;
        const int d_tx = pair_plane ? dp_tx : ds_tx;
        const long long pl = unit * 4 + 2 * ty + d_tx;
        const long long img_l = i0.a + ty * (i0.c - i0.a) + 1, img_r = i0.b + ty * (i0.d - i0.b) + 1;
        const long long img = d_tx ? img_r : img_l;
        const bool ok = (int)unit_ok & (int)(yb < height) & (int)(pl < n_pairs) & (int)(d_x < width);
        uint32_t there = (uint32_t)(img * lpx * 20) + lane_off;
        asm volatile("" : "+v"(there));      // (computed on every path: the select below must stay a select)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R), 0, 0x7fffffff, 0x00020000);
        buf[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? there : NOT_THERE, (uint32_t)(plane_off + yb * width * px_bytes), 0));
      } else {
        const int yb = rows * kk;
        const bool ok = (int)unit_ok & (int)(yb < height) & (int)(d_x < width);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R + (i0.a + 1) * lpx * 5), 0, lpx * 20, 0x00020000);
        buf[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? lane_off : NOT_THERE, (uint32_t)(plane_off + yb * width * px_bytes), 0));
      }
    }
  };
  auto r1_commit = [&](const u32x4 (&buf)[10]) __attribute__((always_inline)) {
    u32x4* dst = reinterpret_cast<u32x4*>(R1s) + lane;
#pragma unroll
    for (int n = 0; n < 10; ++n) dst[(pw + 8 * n) * 64] = buf[n];
  };

  // ---- producing role ----------------------------------------------------------------------------------------------------
  // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472} by distance from the edge; the column's two factors once per lane
  auto border = [](int d) { return d < 2 ? 0.14f : (d < 5 ? 0.4472f : 1.f); };
  float scale_x = __fmul_rn(border(xl), border(width - xl - 1));
  float mA[5][8];                // [channel][pixel]: a channel's 8 values are what a hand-over writes
  fb_f2 fl[8];                   // the flow the next UpdateMatrices starts from
  fb_f2 r01[8], r23[8];          // R0 of this lane's 8 pixels, resident through the unit's iterations
  float r4[8];
  float pm;                      // running largest |M| of the unit being evaluated
  auto lane_pair = [&](long long unit, bool& ok) -> long long {
    const long long pl = MOSAIC ? unit * 4 + 2 * m_ty + m_tx : unit;
    ok = (int)(unit < p_hi) & (int)(pl < n_pairs);
    return ok ? pl : 0;
  };
  // the flow a unit's first iteration starts from -> fl.  FLOW_INIT == 1: cv::resize(INTER_LINEAR) of the coarser level's flow,
  // times 1 / pyr_scale, evaluated on the fly (fb_upsampled_flow_nb's expressions).  The source coordinates are the same for
  // every unit: the column's (tap, weight) once per lane, the 8 rows' once per wave -- in scalar registers
  uint32_t up_v0 = 0, up_v1 = 0;      // byte offsets of the column's two taps inside a source row
  float up_fx = 0.f;
  uint32_t up_rows[8];                // byte offsets of the two source rows, (row1 << 16) | row0 (a source image is <= 32 KB)
  float up_fy[8];
  if constexpr (FLOW_INIT == 1) {
    const int x = min(xl, width - 1);
    float fx = (float)((x + 0.5) * up.inv_fx - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    fx = (sx < 0 || sx >= up.sw - 1) ? 0.f : fx;
    sx = sx < 0 ? 0 : (sx >= up.sw - 1 ? up.sw - 1 : sx);
    up_v0 = (uint32_t)sx * 8, up_v1 = (uint32_t)min(sx + 1, up.sw - 1) * 8, up_fx = fx;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int y = min(__builtin_amdgcn_readfirstlane(yl0 + i), height - 1);
      float fy = (float)((y + 0.5) * up.inv_fy - 0.5);
      int sy = (int)floorf(fy);
      fy -= sy;
      fy = (sy < 0 || sy >= up.sh - 1) ? 0.f : fy;
      sy = sy < 0 ? 0 : (sy >= up.sh - 1 ? up.sh - 1 : sy);
      up_rows[i] = __builtin_amdgcn_readfirstlane((uint32_t)(sy * up.sw * 8) | ((uint32_t)(min(sy + 1, up.sh - 1) * up.sw * 8) << 16));
      up_fy[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fy)));
    }
  }
  auto load_flow_init = [&](long long unit) __attribute__((always_inline)) {
    bool pair_ok;
    const long long pl = lane_pair(unit, pair_ok);
    if constexpr (FLOW_INIT == 1) {
      const int src_bytes = up.sh * up.sw * 8;
      // (MOSAIC: the lanes of a wave read two pairs' images, 31-bit offsets into the whole array -- the launcher's condition)
      const __amdgpu_buffer_rsrc_t rs =
          MOSAIC ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow_prev), 0, 0x7fffffff, 0x00020000)
                 : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow_prev + pl * (src_bytes / 4)), 0, src_bytes, 0x00020000);
      const uint32_t img = MOSAIC ? (uint32_t)pl * (uint32_t)src_bytes : 0u;
      const uint32_t v0 = img + up_v0, v1 = img + up_v1;
      const float a0 = 1.f - up_fx, a1 = up_fx;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t row0 = up_rows[i] & 0xffffu, row1 = up_rows[i] >> 16;
        const float b0 = 1.f - up_fy[i], b1 = up_fy[i];
        const fb_f2 t00 = __builtin_bit_cast(fb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, v0, row0, 0));
        const fb_f2 t10 = __builtin_bit_cast(fb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, v0, row1, 0));
        const fb_f2 t01 = __builtin_bit_cast(fb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, v1, row0, 0));
        const fb_f2 t11 = __builtin_bit_cast(fb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, v1, row1, 0));
        const fb_f2 r0 = t00 * a0 + t01 * a1, r1 = t10 * a0 + t11 * a1;
        fl[i] = (r0 * b0 + r1 * b1) * up.mul;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float z = 0.f;      // (opaque: with a known zero flow every gather address and weight becomes a per-lane invariant)
        asm volatile("" : "+v"(z));
        fl[i] = (fb_f2){z, z};
      }
    }
  };
  // The same from LDS: during a unit's last iteration the flow image's bytes are free, and the NEXT unit's source flow (its
  // pairs' images are contiguous: NP x sh x sw x 8 bytes, 32 KB at most) is brought there through registers -- requested
  // after B2, stored after B3 -- so that the next unit starts from LDS taps instead of waiting for memory
  const int src_bytes = FLOW_INIT == 1 ? up.sh * up.sw * 8 : 0;
  const int src_passes = (NP * src_bytes + 8191) >> 13;      // 512 lanes x 16 bytes per pass
  auto src_fetch = [&](long long unit, u32x4 (&b)[4]) __attribute__((always_inline)) {
    if constexpr (FLOW_INIT == 1) {
      const long long first = unit * NP;
      long long left = unit < p_hi ? (n_pairs - first) * src_bytes : 0;
      left = left < (long long)NP * src_bytes ? left : (long long)NP * src_bytes;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow_prev + first * (src_bytes / 4)), 0, (int)left, 0x00020000);
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (n < src_passes) b[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(n * 512 + tid) * 16, 0, 0));
    }
  };
  auto src_commit = [&](const u32x4 (&b)[4]) __attribute__((always_inline)) {
    if constexpr (FLOW_INIT == 1) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (n < src_passes) *reinterpret_cast<u32x4*>(XsB + (n * 512 + tid) * 16) = b[n];
    }
  };
  auto flow_init_from_lds = [&]() __attribute__((always_inline)) {
    if constexpr (FLOW_INIT == 1) {
      const unsigned char* img = XsB + (MOSAIC ? (2 * m_ty + m_tx) * src_bytes : 0);
      const float a0 = 1.f - up_fx, a1 = up_fx;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t row0 = up_rows[i] & 0xffffu, row1 = up_rows[i] >> 16;
        const float b0 = 1.f - up_fy[i], b1 = up_fy[i];
        const fb_f2 t00 = *reinterpret_cast<const fb_f2*>(img + row0 + up_v0), t10 = *reinterpret_cast<const fb_f2*>(img + row1 + up_v0);
        const fb_f2 t01 = *reinterpret_cast<const fb_f2*>(img + row0 + up_v1), t11 = *reinterpret_cast<const fb_f2*>(img + row1 + up_v1);
        const fb_f2 r0 = t00 * a0 + t01 * a1, r1 = t10 * a0 + t11 * a1;
        fl[i] = (r0 * b0 + r1 * b1) * up.mul;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float z = 0.f;
        asm volatile("" : "+v"(z));
        fl[i] = (fb_f2){z, z};
      }
    }
  };
  // G: the four R1 neighbours of pixel i (R1 is in LDS, the flow in fl): reads, then the bilinear blend -> bl[i]
  auto gather = [&](int i, bool lane_ok, FbTapsP& t) __attribute__((always_inline)) {
    const int y = yl0 + i;
    const float fx0 = __fadd_rn((float)xl, fl[i][0]), fy0 = __fadd_rn((float)y, fl[i][1]);
    const int x1 = (int)floorf(fx0), y1 = (int)floorf(fy0);
    t.fx = __fsub_rn(fx0, (float)x1), t.fy = __fsub_rn(fy0, (float)y1);
    t.inside = (int)lane_ok & (int)(y < height) & (int)((unsigned)x1 < (unsigned)(width - 1)) & (int)((unsigned)y1 < (unsigned)(height - 1));
    int idx = lds_org + y1 * 64 + x1;
    asm volatile("" : "+v"(idx));
    idx = t.inside ? idx : 0;
    const fb_f2* p0 = reinterpret_cast<const fb_f2*>(R1s) + idx;
    const fb_f2* p1 = reinterpret_cast<const fb_f2*>(R1s + 2 * T * T) + idx;
    const float* p2 = R1s + 4 * T * T + idx;
    t.u00 = p0[0], t.u01 = p0[1], t.u10 = p0[64], t.u11 = p0[65];
    t.v00 = p1[0], t.v01 = p1[1], t.v10 = p1[64], t.v11 = p1[65];
    t.w00 = p2[0], t.w01 = p2[1], t.w10 = p2[64], t.w11 = p2[65];
  };
  auto blend = [&](const FbTapsP& t, FbBlend& b) __attribute__((always_inline)) {
    const float fx = t.fx, fy = t.fy;
    const float a00 = __fmul_rn(1.f - fx, 1.f - fy), a01 = __fmul_rn(fx, 1.f - fy);
    const float a10 = __fmul_rn(1.f - fx, fy), a11 = __fmul_rn(fx, fy);
    b.b01 = ((t.u00 * a00 + t.u01 * a01) + t.u10 * a10) + t.u11 * a11;
    b.b23 = ((t.v00 * a00 + t.v01 * a01) + t.v10 * a10) + t.v11 * a11;
    const float b4 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a00, t.w00), __fmul_rn(a01, t.w01)), __fmul_rn(a10, t.w10)), __fmul_rn(a11, t.w11));
    b.b4 = t.inside ? b4 : __builtin_nanf("");
  };
  // F: the rest of UpdateMatrices for pixel i -> mA; pm collects the unit's largest |M|
  auto finish1 = [&](int i, const FbBlend& b) __attribute__((always_inline)) {
    const int y = yl0 + i;
    const float scale = __fmul_rn(__fmul_rn(scale_x, border(y)), border(height - y - 1));
    float m[5];
    fb_update_pixel_finish2(r01[i], r23[i], r4[i], b, fl[i][0], fl[i][1], scale, m);
#pragma unroll
    for (int c = 0; c < 5; ++c) mA[c][i] = m[c];
    pm = fmaxf(pm, fmaxf(fmaxf(fabsf(m[0]), fabsf(m[1])), fmaxf(fmaxf(fabsf(m[2]), fabsf(m[3])), fabsf(m[4]))));
  };
  // UpdateMatrices of the lane's 8 pixels: the next pixel's reads are issued before this pixel's arithmetic
  auto update_matrices = [&](bool lane_ok) __attribute__((always_inline)) {
    FbTapsP ta, tb;
    FbBlend b;
    pm = 0.f;
    gather(0, lane_ok, ta);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      __builtin_amdgcn_sched_barrier(0);
      gather(i + 1, lane_ok, tb);
      blend(ta, b);
      finish1(i, b);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < 8) gather(i + 2, lane_ok, ta);
      blend(tb, b);
      finish1(i + 1, b);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // R0 of this lane's 8 pixels of `unit` (first coefficient images i0) -> r01 / r23 / r4
  auto load_r0 = [&](long long unit, const FbImgs& i0) __attribute__((always_inline)) {
    bool pair_ok;
    (void)lane_pair(unit, pair_ok);
    __amdgpu_buffer_rsrc_t rs;
    uint32_t base;
    if constexpr (MOSAIC) {
      const long long img_l = i0.a + m_ty * (i0.c - i0.a), img_r = i0.b + m_ty * (i0.d - i0.b);      // (no select of addresses)
      rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R), 0, 0x7fffffff, 0x00020000);
      base = (uint32_t)((m_tx ? img_r : img_l) * lpx * 20);
    } else {
      rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R + i0.a * lpx * 5), 0, lpx * 20, 0x00020000);
      base = 0;
    }
    const bool ok = (int)pair_ok & (int)col_ok;
    uint32_t t8 = base + (uint32_t)xl * 8, t4 = base + (uint32_t)xl * 4;
    asm volatile("" : "+v"(t8), "+v"(t4));      // (computed on every path: the selects below must stay selects)
    const uint32_t v8 = ok ? t8 : NOT_THERE, v4 = ok ? t4 : NOT_THERE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int y = yl0 + i;
      const bool row_ok = y < height;
      r01[i] = __builtin_bit_cast(fb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, row_ok ? v8 : NOT_THERE, (uint32_t)(y * width) * 8, 0));
      r23[i] = __builtin_bit_cast(fb_f2, __builtin_amdgcn_raw_buffer_load_b64(rs, row_ok ? v8 : NOT_THERE, (uint32_t)(lpx + y * width) * 8, 0));
      r4[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, row_ok ? v4 : NOT_THERE, (uint32_t)(4 * lpx + y * width) * 4, 0));
    }
  };
  // the unit's largest |M| per lane half -> pmax (MOSAIC: a half is a tile column)
  auto publish_max = [&]() __attribute__((always_inline)) {
    float v = pm;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    pmax[2 * pw + half] = v;      // every lane of the half holds the maximum: 32 identical stores, no branch
  };
  // channel c of mA, scaled and split, as rows y' = 8 pw .. + 7 of row x of X^T: 16 contiguous bytes per plane
  auto write_channel = [&](int c, float s, uint16_t* X) __attribute__((always_inline)) {
    const FbSplit2 sp = fb_split2(mA[c], s);
    uint16_t* Xc = X + lane * XS + 8 * pw;
    *reinterpret_cast<u32x4*>(Xc) = __builtin_bit_cast(u32x4, sp.h);
    *reinterpret_cast<u32x4*>(Xc + PLANE) = __builtin_bit_cast(u32x4, sp.l);
  };
  // ---- multiplying role: Out^T block (mbo, strip) of one channel = Gh (X^T Gv^T) --------------------------------------------
  auto product = [&](const uint16_t* Xc) __attribute__((always_inline)) -> fb_v16f {
    fb_v16f u[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) u[mb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        FbSplit2 a;
        const uint16_t* xa = Xc + (32 * mb + col) * XS + 16 * ks + 8 * half;
        a.h = *reinterpret_cast<const fb_f16x8*>(xa);
        a.l = *reinterpret_cast<const fb_f16x8*>(xa + PLANE);
        u[mb] = fb_mfma2(a, gv[ks], u[mb]);
      }
    }
    fb_v16f res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = u[ks >> 1][8 * (ks & 1) + i];
      const FbSplit2 b = fb_split2(t, FB_G_UNSCALE);
      res = fb_mfma2(gh[ks], b, res);
    }
    return res;
  };
  uint16_t* const X00 = reinterpret_cast<uint16_t*>(XsB);
  uint16_t* const X01 = reinterpret_cast<uint16_t*>(XsB + XIMG);
  uint16_t* const X10 = reinterpret_cast<uint16_t*>(XsB + 2 * XIMG);
  uint16_t* const X11 = reinterpret_cast<uint16_t*>(XsB + 3 * XIMG);
  // mailbox of block sm: slots 0..3 = group 1's channels (3, 4) x accumulator quads (0, 1) for group 0; slots 4..9 = group 0's
  // channels (0, 1, 2) x quads (2, 3) for group 1; a slot = 64 lanes x 16 bytes
  f32x4* mailbox = reinterpret_cast<f32x4*>(XsB + MAIL0) + sm * 640 + lane;
  unsigned char* const flow_img = XsB;
  auto rederive = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane_src));
    lane = lane_src, tid = pw * 64 + lane, col = lane & 31, half = lane >> 5;
    m_tx = lane >> 5, xl = MOSAIC ? (lane & 31) : lane;
    lds_org = MOSAIC ? (32 * m_ty) * 64 + 32 * m_tx : 0;
    col_ok = xl < width;
    dp_x = MOSAIC ? 2 * ((lane & 31) & 15) : 2 * (lane & 31), dp_tx = (lane & 31) >> 4;
    ds_x = MOSAIC ? 4 * ((lane & 15) & 7) : 4 * (lane & 15), ds_tx = (lane & 15) >> 3;
    dma_lane_p = (uint32_t)((lane >> 5) * width + dp_x) * 8, dma_lane_s = (uint32_t)((lane >> 4) * width + ds_x) * 4;
    scale_x = __fmul_rn(border(xl), border(width - xl - 1));
    mailbox = reinterpret_cast<f32x4*>(XsB + MAIL0) + sm * 640 + lane;
  };

#ifdef PV_DIAG_STAMPS
  unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, s0, s1;
#ifdef FBU_BARRIER_DETAIL      // slots 0..5 = the waits at Bm, B0..B4, slot 6 = everything else
#define FBU_PHASE(slot) do { PV_STAMP(s1); dg[6] += s1 - s0; s0 = s1; } while (0)
#define FBU_BAR(n) do { PV_STAMP(s1); dg[n] += s1 - s0; s0 = s1; } while (0)
#else
#define FBU_PHASE(slot) do { PV_STAMP(s1); dg[slot] += s1 - s0; s0 = s1; } while (0)
#define FBU_BAR(n) FBU_PHASE(6)
#endif
#else
#define FBU_PHASE(slot) do { } while (0)
#define FBU_BAR(n) do { } while (0)
#endif

  // one iteration of unit u; LAST: the unit's last (the next unit is prepared under it, the flow goes to memory)
  FbImgs i0c, i0n;
  auto stage = [&](long long u, bool lane_ok, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    PV_STAMP(s0);
    rederive();
    lane_ok = (int)lane_ok & (int)col_ok;      // (the column test with this phase's lane number)
    // ---- P -----------------------------------------------------------------------------------------------------------------
    update_matrices(lane_ok);
    FBU_PHASE(0);
    publish_max();
    FBU_PHASE(1);
    __syncthreads();      // Bm: the unit's largest |M| is published; every wave is through with this iteration's gathers
    FBU_BAR(0);
    rederive();
    u32x4 r1buf[10];
    if constexpr (LAST) r1_fetch(i0n, u + 1, r1buf);      // (into the registers R0 has left)
    const float s = ldexpf(1.f, 15 - unit_exp(m_ty, m_tx));
    write_channel(0, s, X00), write_channel(3, s, X01);
    FBU_PHASE(2);
    __syncthreads();      // B0
    FBU_BAR(1);
    // ---- M -----------------------------------------------------------------------------------------------------------------
    write_channel(1, s, X10), write_channel(4, s, X11);
    fb_v16f res[3];
    res[0] = product(grp ? X01 : X00);      // group 0: channel 0; group 1: channel 3
    FBU_PHASE(3);
    __syncthreads();      // B1
    FBU_BAR(2);
    if constexpr (LAST) r1_commit(r1buf);      // (the compiler's own wait for the fetch; the next gathers are behind B4)
    write_channel(2, s, X00);
    res[1] = product(grp ? X11 : X10);      // group 0: channel 1; group 1: channel 4
    FBU_PHASE(3);
    __syncthreads();      // B2
    FBU_BAR(3);
    u32x4 srcbuf[4];
    if constexpr (LAST) src_fetch(u + 1, srcbuf);
    // each group sends the other the half of its blurred channels it will not solve itself: accumulator quads 0, 1 (columns
    // 8 g4 + 4 half .. + 3 for g4 = 0, 1) are solved by group 0, quads 2, 3 by group 1
    if (grp) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) mailbox[(c * 2 + q) * 64] = (f32x4){res[c][4 * q], res[c][4 * q + 1], res[c][4 * q + 2], res[c][4 * q + 3]};
    } else {
      res[2] = product(X00);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 2; q < 4; ++q) mailbox[(4 + c * 2 + q - 2) * 64] = (f32x4){res[c][4 * q], res[c][4 * q + 1], res[c][4 * q + 2], res[c][4 * q + 3]};
    }
    FBU_PHASE(3);
    __syncthreads();      // B3: the mailboxes are filled; nobody reads an X image any more
    FBU_BAR(4);
    rederive();
    if constexpr (LAST) {
      src_commit(srcbuf);
      load_r0(u + 1, i0n);   // into the registers R0 of this unit has left; wanted after the next unit's first gathers
    }
    // ---- S: 2 x 2 solve on sums that carry the factor kk = 2^15 s = 2^(30 - e); lane = row y, registers = columns x ------------
    {
      const int e = unit_exp(strip, mbo);
      const float lam = ldexpf(1e-3f, 2 * (30 - e));
      const int y = (MOSAIC ? 0 : 32 * strip) + col;
      const long long pr = MOSAIC ? u * 4 + 2 * strip + mbo : u;
      float* frow = flow + (pr * lpx + (long long)y * width) * 2;
      unsigned char* fimg = flow_img + (32 * strip + col) * FROW + (32 * mbo + 4 * half) * 8;
      // quad g4 of the block from the five blurred channels' registers q5[channel][j]
      auto solve_quad = [&](int g4, const float (&q5)[5][4]) {
        const int x0 = (MOSAIC ? 0 : 32 * mbo) + 8 * g4 + 4 * half;
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g11 = q5[0][j], g12 = q5[1][j], g22 = q5[2][j], h1 = q5[3][j], h2 = q5[4][j];
          // det >= lam > 0 (a sum of squares' determinant plus the regulariser), well inside the normal range: one hardware
          // reciprocal (1 ulp), one Newton step (0.5 ulp), then each quotient corrected once by its residual
          const float det = __fadd_rn(fb_det2(g11, g22, g12, g12), lam);
          float rc = __builtin_amdgcn_rcpf(det);
          rc = __builtin_fmaf(__builtin_fmaf(-det, rc, 1.f), rc, rc);
          const float nx = fb_det2(g11, h2, g12, h1), ny = fb_det2(g22, h1, g12, h2);
          const float qx = __fmul_rn(nx, rc), qy = __fmul_rn(ny, rc);
          o[2 * j] = __builtin_fmaf(__builtin_fmaf(-det, qx, nx), rc, qx);
          o[2 * j + 1] = __builtin_fmaf(__builtin_fmaf(-det, qy, ny), rc, qy);
        }
        if constexpr (LAST) {
          if (y < height && x0 < width && pr < n_pairs) {      // rows of whole 4-pixel quads (the launcher's condition)
            *reinterpret_cast<f32x4*>(frow + x0 * 2) = (f32x4){o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(frow + x0 * 2 + 4) = (f32x4){o[4], o[5], o[6], o[7]};
          }
        } else {
          *reinterpret_cast<f32x4*>(fimg + g4 * 64) = (f32x4){o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f32x4*>(fimg + g4 * 64 + 16) = (f32x4){o[4], o[5], o[6], o[7]};
        }
      };
      if (grp) {
#pragma unroll
        for (int q = 2; q < 4; ++q) {
          float q5[5][4];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const f32x4 v = mailbox[(4 + c * 2 + q - 2) * 64];
            q5[c][0] = v[0], q5[c][1] = v[1], q5[c][2] = v[2], q5[c][3] = v[3];
          }
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) q5[3 + c][j] = res[c][4 * q + j];
          solve_quad(q, q5);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float q5[5][4];
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) q5[c][j] = res[c][4 * q + j];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x4 v = mailbox[(c * 2 + q) * 64];
            q5[3 + c][0] = v[0], q5[3 + c][1] = v[1], q5[3 + c][2] = v[2], q5[3 + c][3] = v[3];
          }
          solve_quad(q, q5);
        }
      }
    }
    FBU_PHASE(4);
    __syncthreads();      // B4: the flow image (not LAST) / the next unit's R1 and source flow (LAST) are complete
    FBU_BAR(5);
    rederive();
    if constexpr (LAST) {
      flow_init_from_lds();
    } else {
      const unsigned char* fsrc = flow_img + (8 * pw) * FROW + lane * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const fb_f2 v = *reinterpret_cast<const fb_f2*>(fsrc + i * FROW);
        const bool px_ok = (int)lane_ok & (int)(yl0 + i < height);
        fl[i] = px_ok ? v : (fb_f2){0.f, 0.f};
      }
    }
    FBU_PHASE(5);
#ifdef PV_DIAG_STAMPS
    dg[7] += 1;
#endif
  };

  take_unit(i0c);
  take_unit(i0n);
  {
    u32x4 r1buf[10];
    r1_fetch(i0c, p_lo, r1buf);
    load_flow_init(p_lo);
    load_r0(p_lo, i0c);
    r1_commit(r1buf);
    __syncthreads();      // R1 of the first unit is in place
  }
  for (long long u = p_lo; u < p_hi; ++u) {
    bool ok;
    (void)lane_pair(u, ok);
    for (int it = 0; it + 1 < n_it; ++it) stage(u, ok, std::false_type{});
    stage(u, ok, std::true_type{});
    i0c = i0n;
    take_unit(i0n);
  }
#undef FBU_PHASE
#undef FBU_BAR
#ifdef PV_DIAG_STAMPS
  if (lane == 0 && blockIdx.x * 8 + pw < PV_DIAG_WAVES)
    for (int i = 0; i < PV_DIAG_SLOTS; ++i) fb_iter_diag[(size_t)(blockIdx.x * 8 + pw) * PV_DIAG_SLOTS + i] = dg[i];
#endif
}

// ---- prep + PolyExp with the two PolyExp passes on the F16 matrix cores (round 4) -------------------------------------------
// fb_prep_polyexp_tile_kernel spends 23 000 cycles per 64 x 64 image on the two separable 2n+1-tap passes (index arithmetic
// and LDS reads of a per-pixel tap loop).  Both passes are products with banded 64 x 64 matrices that already contain the
// border replication (fb_polyexp_matrix_kernel): vertically t_j = V_j I for V_g, V_xg, V_xxg, horizontally
//   b1 = t0 H_g^T  b2 = t0 H_xg^T  b4 = t0 H_xxg^T  b3 = t1 H_g^T  b6 = t1 H_xg^T  b5 = t2 H_g^T,
// nine products per image, evaluated exactly like the window blur of fb_level_u_kernel: the image handed over transposed as two
// half-float planes (x s = h + l, 22 bits), first product U_j = X^T V_j^T, its accumulators re-split as the B operand of the
// second, three matrix instructions per two-term pair.  Every scale is a power of two fixed by the taps (the image is 8-bit:
// |I| <= 255, |t_j| <= 255 max-row-sum(V_j)): no reductions.  Eight waves: wave (strip, mbo, g) owns output block (strip, mbo);
// group 0 takes t0 (b1, b2, b4), group 1 takes t1 and t2 (b3, b6, b5) and hands b3 and b5 over through LDS, so that group 0
// stores the (c0, c1) and (c2, c3) pairs whole and group 1 the c4 plane.  The 3-tap Gaussian and the resize before it stay on
// the vector ALU (fb_prep_polyexp_tile_kernel's expressions).  MOSAIC (levels up to 32 x 32): four images per tile, block-
// diagonal matrices.  Accumulation order differs from the tap loops: results agree to ~1e-6 relative, not bit for bit.
#ifdef PV_DIAG_STAMPS
__device__ unsigned long long fb_polyexp_diag[PV_DIAG_WAVES * PV_DIAG_SLOTS];
#endif
struct FbPolyScales {
  float sV[3], sH[3], fU[3];      // scales of V_g, V_xg, V_xxg / H_g, H_xg, H_xxg; re-split factors of U_0..U_2
  float c11_b3, c11_b2, c03_b1, c33_b5, c33_b4, c55_b6;      // ig.. with the products' scales divided out
};

__global__ __launch_bounds__(256) void fb_polyexp_matrix_kernel(float* __restrict__ P6, int lh, int lw, FbPoly pk, int mosaic) {
  // P6[3 v + j][64][64]: v = 0 vertical (size lh), 1 horizontal (size lw); j = 0: g, 1: x g (odd), 2: x^2 g.  Row y, column y':
  // the weight of input y' in output y, border replicated.  mosaic (sizes <= 32): the matrix twice on the diagonal (0 and 32)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 6 * 64 * 64; i += gridDim.x * blockDim.x) {
    const int which = i >> 12, j = which % 3, n = which < 3 ? lh : lw;
    int y = (i >> 6) & 63, yp = i & 63;
    const bool same_block = !mosaic || (y >> 5) == (yp >> 5);
    if (mosaic) y &= 31, yp &= 31;
    float s = 0.f;
    if (same_block && y < n && yp < n) {
      if (j == 0 && y == yp) s = pk.g[0];
      for (int k = 1; k <= pk.n; ++k) {
        const float c = j == 0 ? pk.g[k] : (j == 1 ? pk.xg[k] : pk.xxg[k]);
        if (min(y + k, n - 1) == yp) s += c;
        if (max(y - k, 0) == yp) s += j == 1 ? -c : c;
      }
    }
    P6[i] = s;
  }
}

template <bool MOSAIC>
__global__ __launch_bounds__(512) void fb_prep_polyexp_mfma_kernel(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ next,
                                                                   long long prev_stride, long long next_stride,
                                                                   long long pairs_per_group, long long group_stride,
                                                                   float* __restrict__ R, long long n_img, int chain_f, int h, int w,
                                                                   int lh, int lw, int mode, double inv_fx, double inv_fy, FbTaps kt,
                                                                   const float* __restrict__ P6, FbPolyScales sc) {
  constexpr int XS = 64 + 8, PLANE = 64 * XS, KS = 4, NI = MOSAIC ? 4 : 1;
  __shared__ float bufA[64 * 64];        // source as float, later the blurred image
  __shared__ __attribute__((aligned(16))) uint16_t Xs[2][PLANE];      // the level image(s), transposed, (h, l) half-float planes
  __shared__ __attribute__((aligned(16))) f32x4 mail[4][2][4][64];    // group 1 -> group 0: b3, b5 of block sm
  const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (everything derived from the lane number is re-derived per phase from an opaque copy, as in fb_level_u_kernel: the phases'
  // addresses would otherwise stay alive through each other -- some sixty registers beside 128 of operands)
  int lane_src = threadIdx.x & 63;
  int lane = lane_src, tid = pw * 64 + lane, col = lane & 31, half = lane >> 5;
  const int sm = pw & 3, strip = sm & 1, mbo = sm >> 1, grp = pw >> 2;
  const int npx = h * w, lpx = lh * lw;

  // operands (128 registers): group 0: V_g | H_g, H_xg, H_xxg; group 1: V_xg, V_xxg | H_g, H_xg.  opX is the group's fourth
  // matrix: H_xxg (an A operand) for group 0, V_xxg (a B operand) for group 1
  FbSplit2 gvA[KS], opX[KS], ghA[KS], ghB[KS];
  {
    const float* va = P6 + (grp ? 1 : 0) * 4096;
    const float sva = grp ? sc.sV[1] : sc.sV[0];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = va[(32 * strip + col) * 64 + 16 * ks + 8 * half + i];
      gvA[ks] = fb_split2(t, sva);
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = P6[3 * 4096 + (32 * mbo + col) * 64 + 32 * (ks >> 1) + fb_acc_row(8 * (ks & 1) + i, half)];
      ghA[ks] = fb_split2(t, sc.sH[0]);
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = P6[4 * 4096 + (32 * mbo + col) * 64 + 32 * (ks >> 1) + fb_acc_row(8 * (ks & 1) + i, half)];
      ghB[ks] = fb_split2(t, sc.sH[1]);
      // (one address expression for both groups: the group picks the matrix, the row and the column order by arithmetic)
      const int xrow = grp ? 32 * strip + col : 32 * mbo + col;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int xcol = grp ? 16 * ks + 8 * half + i : 32 * (ks >> 1) + fb_acc_row(8 * (ks & 1) + i, half);
        t[i] = P6[(grp ? 2 : 5) * 4096 + xrow * 64 + xcol];
      }
      opX[ks] = fb_split2(t, grp ? sc.sV[2] : sc.sH[2]);
    }
  }
  auto first = [&](const FbSplit2 (&gv)[KS], float f, FbSplit2 (&b)[KS]) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {      // row block mb of U becomes k-steps 2 mb, 2 mb + 1 of the second product: one block alive
      fb_v16f u;
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        FbSplit2 a;
        const uint16_t* xa = &Xs[0][0] + (32 * mb + col) * XS + 16 * ks + 8 * half;
        a.h = *reinterpret_cast<const fb_f16x8*>(xa);
        a.l = *reinterpret_cast<const fb_f16x8*>(xa + PLANE);
        u = fb_mfma2(a, gv[ks], u);
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = u[8 * k2 + i];
        b[2 * mb + k2] = fb_split2(t, f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto second = [&](const FbSplit2 (&gh)[KS], const FbSplit2 (&b)[KS]) __attribute__((always_inline)) -> fb_v16f {
    fb_v16f res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.f;
#pragma unroll
    // U as the A operand, the window matrix as B (the same register contents: the two operand layouts mirror each other): the
    // result is the block itself rather than its transpose -- lane = COLUMN x, registers = rows -- so that the 32 lanes of a
    // half wave store 256 contiguous bytes of a row
    for (int ks = 0; ks < KS; ++ks) res = fb_mfma2(b[ks], gh[ks], res);
    return res;
  };

  const long long n_units = (n_img + NI - 1) / NI;
  // The Gaussian before the resize is the 3-tap one here (the launcher's condition: sigma 0 / 0.5 at pyr_scale 0.5), the resize a
  // copy (mode 0) or the exact 2 x 2 mean (mode 1): a thread owns ONE column and 8 rows of the source -- the row filter of its
  // 10 rows straight from the source in LDS (3 reads per row, consecutive lanes), the column filter and the resize from
  // registers (mode 1: the right-hand neighbour by a lane shuffle), one barrier per image.  Same expressions and rounding
  // order as fb_prep_polyexp_tile_kernel's filters.
  const float k0 = kt.k[1], k1 = kt.k[2];      // taps (k1, k0, k1) of the 3-tap kernel: kt.k[rr], kt.k[rr + 1] with rr = 1
  int xm = lane == 0 ? 1 : lane - 1, xp = lane == w - 1 ? w - 2 : lane + 1;      // BORDER_REFLECT_101
  auto rederive = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane_src));
    lane = lane_src, tid = pw * 64 + lane, col = lane & 31, half = lane >> 5;
    xm = lane == 0 ? 1 : lane - 1, xp = lane == w - 1 ? w - 2 : lane + 1;
  };
  // the source image of (unit, q) as two 32-bit words per thread (8 pixels: row tid / (w / 8)... flat index 8 tid), prefetched
  auto fetch_src = [&](long long unit, int q) __attribute__((always_inline)) -> u32x2 {
    const long long im = unit * NI + q;
    const bool ok = (int)(unit < n_units) & (int)(im < n_img) & (int)(8 * tid < npx);
    const uint8_t* img = fb_image_of(ok ? im : 0, prev, next, prev_stride, next_stride, pairs_per_group, group_stride, chain_f);
    u32x2 v = {0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x2*>(img + 8 * tid);
    return v;
  };
  u32x2 src_next = fetch_src(blockIdx.x, 0);
#ifdef PV_DIAG_STAMPS
  unsigned long long dg[PV_DIAG_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, s0, s1;
#define FBP_PHASE(slot) do { PV_STAMP(s1); dg[slot] += s1 - s0; s0 = s1; } while (0)
  PV_STAMP(s0);
#else
#define FBP_PHASE(slot) do { } while (0)
#endif
  for (long long unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    // ---- the level image(s): u8 -> 3-tap Gaussian -> resize -> half-float planes, transposed ---------------------------------
#pragma unroll 1
    for (int q = 0; q < NI; ++q) {
      const bool im_ok = unit * NI + q < n_img;
      rederive();
      {      // the source as floats in LDS, rows of w
        const u32x2 sv = src_next;
        if (8 * tid < npx) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = (float)((sv[e >> 2] >> (8 * (e & 3))) & 0xffu);
          *reinterpret_cast<f32x4*>(bufA + 8 * tid) = (f32x4){f[0], f[1], f[2], f[3]};
          *reinterpret_cast<f32x4*>(bufA + 8 * tid + 4) = (f32x4){f[4], f[5], f[6], f[7]};
        }
      }
      // the next source image's words are on their way while this one is filtered and multiplied
      src_next = q + 1 < NI ? fetch_src(unit, q + 1) : fetch_src(unit + gridDim.x, 0);
      FBP_PHASE(0);
      __syncthreads();
      FBP_PHASE(1);
      float bl[8];      // the blurred image at (rows 8 pw .. + 7, column lane)
      {
        float rf[10];      // row-filtered rows 8 pw - 1 .. 8 pw + 8 (reflected at the image's edge)
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          int y = 8 * pw - 1 + i;
          y = y < 0 ? 1 : (y > h - 1 ? 2 * h - 2 - y : y);
          y = y > h - 1 ? h - 1 : (y < 0 ? 0 : y);      // (rows of waves beyond the image: any valid row)
          const float* srow = bufA + y * w;
          const float c = lane < w ? srow[lane] : 0.f, l = lane < w ? srow[xm] : 0.f, r = lane < w ? srow[xp] : 0.f;
          rf[i] = __fadd_rn(c * k0, __fmul_rn(l + r, k1));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bl[i] = __fadd_rn(__fmul_rn(rf[i + 1], k0), __fmul_rn(__fadd_rn(rf[i], rf[i + 2]), k1));
      }
      if (mode == 0) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (int)im_ok & (int)(lane < lw) & (int)(8 * pw + i < lh) ? bl[i] : 0.f;
        const FbSplit2 sp = fb_split2(v, 64.f);      // |I| <= 255: below 2^14
        if (!MOSAIC || (pw < 4 && lane < 32)) {      // (MOSAIC: a source of up to 32 x 32 fills tile (q >> 1, q & 1))
          const int ty = MOSAIC ? q >> 1 : 0, tx = MOSAIC ? q & 1 : 0;
          uint16_t* Xc = &Xs[0][0] + (32 * tx + lane) * XS + 32 * ty + 8 * pw;
          *reinterpret_cast<u32x4*>(Xc) = __builtin_bit_cast(u32x4, sp.h);
          *reinterpret_cast<u32x4*>(Xc + PLANE) = __builtin_bit_cast(u32x4, sp.l);
        }
      } else {
        // mode 1: level pixel (y', x') = mean of the 2 x 2 block at (2 y', 2 x'): the even lanes combine their column with the
        // next lane's; this wave's rows 8 pw .. + 7 give level rows 4 pw .. + 3
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = __fadd_rn(bl[2 * i], __shfl_down(bl[2 * i], 1, 64));
          const float b = __fadd_rn(bl[2 * i + 1], __shfl_down(bl[2 * i + 1], 1, 64));
          const int xo = lane >> 1, yo = 4 * pw + i;
          v[i] = (int)im_ok & (int)(xo < lw) & (int)(yo < lh) ? __fmul_rn(__fadd_rn(a, b), 0.25f) : 0.f;
        }
        const FbSplit2 sp = fb_split2(v, 64.f);
        if ((lane & 1) == 0) {
          const int ty = MOSAIC ? q >> 1 : 0, tx = MOSAIC ? q & 1 : 0;
          uint16_t* Xc = &Xs[0][0] + (32 * tx + (lane >> 1)) * XS + 32 * ty + 4 * pw;
          const u32x4 hw = __builtin_bit_cast(u32x4, sp.h), lw4 = __builtin_bit_cast(u32x4, sp.l);
          *reinterpret_cast<u32x2*>(Xc) = (u32x2){hw[0], hw[1]};
          *reinterpret_cast<u32x2*>(Xc + PLANE) = (u32x2){lw4[0], lw4[1]};
        }
      }
      FBP_PHASE(2);
      __syncthreads();      // X complete (last image of the unit) / bufA free for the next image
      FBP_PHASE(1);
    }
    // ---- the nine products ---------------------------------------------------------------------------------------------------
    rederive();
    FbSplit2 b[KS];
    fb_v16f r0, r1, r2;      // group 0: b1, b2, b4; group 1: b3, b6, b5
    if (grp == 0) {
      first(gvA, sc.fU[0], b);
      r0 = second(ghA, b), r1 = second(ghB, b), r2 = second(opX, b);
    } else {
      first(gvA, sc.fU[1], b);
      r0 = second(ghA, b);      // b3: handed over at once
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) mail[sm][0][qd][lane] = (f32x4){r0[4 * qd], r0[4 * qd + 1], r0[4 * qd + 2], r0[4 * qd + 3]};
      __builtin_amdgcn_sched_barrier(0);
      r1 = second(ghB, b);      // b6: this group's own plane
      first(opX, sc.fU[2], b);
      r2 = second(ghA, b);      // b5
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) mail[sm][1][qd][lane] = (f32x4){r2[4 * qd], r2[4 * qd + 1], r2[4 * qd + 2], r2[4 * qd + 3]};
    }
    FBP_PHASE(3);
    __syncthreads();      // the mailboxes are filled; nobody reads X any more
    FBP_PHASE(1);
    rederive();
    // ---- R = the polynomial coefficients; lane = column x of the block, registers = its rows ---------------------------------------
    {
      const long long im = MOSAIC ? unit * 4 + 2 * strip + mbo : unit;
      const int x = (MOSAIC ? 0 : 32 * mbo) + col;
      float* d = R + im * lpx * 5;
      if (im < n_img && x < lw) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          f32x4 b3 = {0.f, 0.f, 0.f, 0.f}, b5 = {0.f, 0.f, 0.f, 0.f};
          if (grp == 0) b3 = mail[sm][0][qd][lane], b5 = mail[sm][1][qd][lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * qd + e;
            const int y = (MOSAIC ? 0 : 32 * strip) + fb_acc_row(r, half);
            if (y < lh) {
              if (grp == 0) {
                const float b1s = __fmul_rn(r0[r], sc.c03_b1);
                *reinterpret_cast<fb_f2*>(d + ((size_t)y * lw + x) * 2) = (fb_f2){__fmul_rn(b3[e], sc.c11_b3), __fmul_rn(r1[r], sc.c11_b2)};
                *reinterpret_cast<fb_f2*>(d + 2 * lpx + ((size_t)y * lw + x) * 2) =
                    (fb_f2){__builtin_fmaf(b5[e], sc.c33_b5, b1s), __builtin_fmaf(r2[r], sc.c33_b4, b1s)};
              } else {
                d[4 * lpx + (size_t)y * lw + x] = __fmul_rn(r1[r], sc.c55_b6);
              }
            }
          }
        }
      }
    }
    FBP_PHASE(4);
    __syncthreads();      // the mailboxes and X are reused by the next unit
    FBP_PHASE(1);
#ifdef PV_DIAG_STAMPS
    dg[7] += 1;
#endif
  }
#undef FBP_PHASE
#ifdef PV_DIAG_STAMPS
  if (lane == 0 && blockIdx.x * 8 + pw < PV_DIAG_WAVES)
    for (int i = 0; i < PV_DIAG_SLOTS; ++i) fb_polyexp_diag[(size_t)(blockIdx.x * 8 + pw) * PV_DIAG_SLOTS + i] = dg[i];
#endif
}

// ---- Gaussian window blur of the 5-channel M, vertical then horizontal (+ 2x2 solve) --------------
__global__ __launch_bounds__(256) void fb_blur_v_kernel(const float* __restrict__ M, float* __restrict__ V,
                                                         long long n_pairs, int height, int width, FbTaps kt) {
  const long long row_elems = (long long)width * 5;
  const long long per_img = (long long)height * row_elems;
  const long long total = n_pairs * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int m = kt.n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long p = i / per_img;
    long long rem = i - p * per_img;
    int y = (int)(rem / row_elems);
    int xe = (int)(rem - (long long)y * row_elems);
    const float* src = M + p * per_img + xe;
    float s0 = __fmul_rn(src[(size_t)y * row_elems], kt.k[0]);
    for (int k = 1; k <= m; ++k) {
      float a = src[(size_t)min(y + k, height - 1) * row_elems];
      float b = src[(size_t)max(y - k, 0) * row_elems];
      s0 = __fadd_rn(s0, __fmul_rn(__fadd_rn(a, b), kt.k[k]));
    }
    V[i] = s0;
  }
}

__global__ __launch_bounds__(256) void fb_blur_h_solve_kernel(const float* __restrict__ V, float* __restrict__ flow,
                                                               long long n_pairs, int height, int width, FbTaps kt) {
  const long long per_img = (long long)height * width;
  const long long total = n_pairs * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int m = kt.n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long p = i / per_img;
    int rem = (int)(i - p * per_img);
    int y = rem / width, x = rem - y * width;
    const float* row = V + (p * per_img + (size_t)y * width) * 5;
    float h5[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) h5[c] = __fmul_rn(row[(size_t)x * 5 + c], kt.k[0]);
    for (int k = 1; k <= m; ++k) {
      const float* a = row + (size_t)max(x - k, 0) * 5;
      const float* b = row + (size_t)min(x + k, width - 1) * 5;
      float kk = kt.k[k];
#pragma unroll
      for (int c = 0; c < 5; ++c) h5[c] = __fadd_rn(h5[c], __fmul_rn(kk, __fadd_rn(a[c], b[c])));
    }
    double g11 = h5[0], g12 = h5[1], g22 = h5[2], h1 = h5[3], h2 = h5[4];
    double det = __dadd_rn(__dsub_rn(__dmul_rn(g11, g22), __dmul_rn(g12, g12)), 1e-3);
    double idet = __ddiv_rn(1.0, det);
    float* fl = flow + i * 2;
    fl[0] = (float)__dmul_rn(__dsub_rn(__dmul_rn(g11, h2), __dmul_rn(g12, h1)), idet);
    fl[1] = (float)__dmul_rn(__dsub_rn(__dmul_rn(g22, h1), __dmul_rn(g12, h2)), idet);
  }
}

// ---- the same two passes for FRAMES (levels larger than one 64 x 64 tile: the 704 x 548 images of the notebooks) ------
// Each thread keeps a RUN + 2 MW window of its column in registers and slides the MW-tap window down it: RUN + 2 MW loads
// for RUN outputs instead of 2 MW + 1 per output, every one of them coalesced across the wave.  The vertical pass writes
// its result TRANSPOSED and planar (Vt[pair][channel][x][y], rows padded to hp) so that the horizontal pass is the same
// walk: a lane per image row, the window sliding along x.  Operation order as in the two kernels above: identical bits.
template <int MW, int RUN>
__global__ __launch_bounds__(256) void fb_blur_v_run_kernel(const float* __restrict__ M, float* __restrict__ Vt, int height,
                                                             int width, int hp, int n_runs, int n_xb, long long n_strips,
                                                             FbTaps kt) {
  static_assert(RUN == 32, "the transposing store below moves 32-row segments, two per wave instruction");
  __shared__ float ot[RUN][257];        // the block's RUN x 256 outputs, transposed on the way out (odd row stride: no conflicts)
  __shared__ unsigned seg_off[256];     // where each element column's RUN-float segment starts in Vt (elements, per pair)
  // workgroups go round-robin over the 8 XCDs: a strip (one pair, 256 element columns, every run of rows) stays on ONE XCD,
  // its runs back to back, so the 2 MW rows two neighbouring runs share are hits in that XCD's L2
  const long long jx = blockIdx.x >> 3;
  const long long strip = (jx / n_runs) * 8 + (blockIdx.x & 7);
  if (strip >= n_strips) return;
  const int run = (int)(jx % n_runs);
  const int xb = (int)(strip % n_xb);
  const long long p = strip / n_xb;
  const int row_elems = width * 5;
  const int xe = min(xb * 256 + (int)threadIdx.x, row_elems - 1);   // (lanes beyond the row repeat its last column; not stored)
  const int y0 = run * RUN;
  const float* src = M + p * (long long)height * row_elems + xe;
  float wv[RUN + 2 * MW];
#pragma unroll
  for (int j = 0; j < RUN + 2 * MW; ++j) {
    const int y = min(max(y0 - MW + j, 0), height - 1);
    wv[j] = src[(size_t)y * row_elems];
  }
  const int x = xe / 5, c = xe - x * 5;
  seg_off[threadIdx.x] = (unsigned)((c * width + x) * hp + y0);   // hp is a multiple of RUN: 128-byte segments
#pragma unroll
  for (int r = 0; r < RUN; ++r) {
    float s0 = __fmul_rn(wv[r + MW], kt.k[0]);
#pragma unroll
    for (int k = 1; k <= MW; ++k) s0 = __fadd_rn(s0, __fmul_rn(__fadd_rn(wv[r + MW + k], wv[r + MW - k]), kt.k[k]));
    ot[r][threadIdx.x] = s0;
  }
  __syncthreads();
  float* dst = Vt + p * 5 * (long long)width * hp;
  const int n_cols = min(256, row_elems - xb * 256);
  const int yy = threadIdx.x & 31, s8 = threadIdx.x >> 5;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {        // a half wave per segment: 128 contiguous bytes
    const int sg = i * 8 + s8;
    if (sg < n_cols) dst[seg_off[sg] + yy] = ot[yy][sg];
  }
}

template <int MW, int RUN>
__global__ __launch_bounds__(64) void fb_blur_h_solve_run_kernel(const float* __restrict__ Vt, float* __restrict__ flow,
                                                                  int height, int width, int hp, int n_runs, int n_yb,
                                                                  long long n_strips, FbTaps kt) {
  __shared__ __attribute__((aligned(16))) float hs[5][64][RUN];   // the five blurred channels of the lane's RUN pixels (each lane reads back its own words)
  const long long jx = blockIdx.x >> 3;                      // (a strip = one pair, 64 rows, every run of columns: one XCD)
  const long long strip = (jx / n_runs) * 8 + (blockIdx.x & 7);
  if (strip >= n_strips) return;
  const int run = (int)(jx % n_runs);
  const int yb = (int)(strip % n_yb);
  const long long p = strip / n_yb;
  const int y = yb * 64 + threadIdx.x;
  if (y >= height) return;
  const int x0 = run * RUN;
  const float* base = Vt + p * 5 * (long long)width * hp + y;
#pragma unroll 1
  for (int c = 0; c < 5; ++c) {   // one channel's window in registers at a time
    const float* src = base + (size_t)c * width * hp;
    float wv[RUN + 2 * MW];
#pragma unroll
    for (int j = 0; j < RUN + 2 * MW; ++j) {
      const int x = min(max(x0 - MW + j, 0), width - 1);
      wv[j] = src[(unsigned)(x * hp)];
    }
    float o[RUN];
#pragma unroll
    for (int r = 0; r < RUN; ++r) {
      float s0 = __fmul_rn(wv[r + MW], kt.k[0]);
#pragma unroll
      for (int k = 1; k <= MW; ++k) s0 = __fadd_rn(s0, __fmul_rn(__fadd_rn(wv[r + MW - k], wv[r + MW + k]), kt.k[k]));
      o[r] = s0;
    }
#pragma unroll
    for (int r = 0; r < RUN; r += 4) *reinterpret_cast<float4*>(&hs[c][threadIdx.x][r]) = make_float4(o[r], o[r + 1], o[r + 2], o[r + 3]);
  }
  float* fl = flow + ((p * height + y) * (long long)width + x0) * 2;
#pragma unroll
  for (int r = 0; r < RUN; ++r) {
    if (x0 + r < width) {
      const double g11 = hs[0][threadIdx.x][r], g12 = hs[1][threadIdx.x][r], g22 = hs[2][threadIdx.x][r],
                   h1 = hs[3][threadIdx.x][r], h2 = hs[4][threadIdx.x][r];
      const double det = __dadd_rn(__dsub_rn(__dmul_rn(g11, g22), __dmul_rn(g12, g12)), 1e-3);
      const double idet = __ddiv_rn(1.0, det);
      *reinterpret_cast<float2*>(fl + 2 * r) = make_float2((float)__dmul_rn(__dsub_rn(__dmul_rn(g11, h2), __dmul_rn(g12, h1)), idet),
                                                           (float)__dmul_rn(__dsub_rn(__dmul_rn(g22, h1), __dmul_rn(g12, h2)), idet));
    }
  }
}
constexpr int FB_RUN_V = 32, FB_RUN_H = 8; static_assert(FB_RUN_V <= 32 && FB_RUN_V % 4 == 0, "fb_layout pads the transposed rows to 32");   // outputs per thread of the two frame passes

// ---- flow upsample: cv::resize(prevFlow -> (lw, lh), INTER_LINEAR) * (1 / pyr_scale) --------------
// IDX = int when the whole output fits 31 bits (the per-element 64-bit division otherwise dominates the kernel); both
// flow components of a tap travel as one 8-byte load / store.
template <typename IDX>
__global__ __launch_bounds__(256) void fb_flow_upsample_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                long long n_pairs, int sh, int sw, int dh, int dw,
                                                                double inv_fx, double inv_fy, float mul) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const IDX per_img = (IDX)dh * dw;
  const IDX total = (IDX)n_pairs * per_img;
  const IDX stride = (IDX)gridDim.x * blockDim.x;
  for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const IDX p = i / per_img;
    const int rem = (int)(i - p * per_img);
    const int y = rem / dw, x = rem - y * dw;
    float fx = (float)((x + 0.5) * inv_fx - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)((y + 0.5) * inv_fy - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    const int sy1 = clampi_d(sy + 1, 0, sh - 1);
    const f32x2_t* s0 = reinterpret_cast<const f32x2_t*>(src) + ((size_t)p * sh + sy) * sw;
    const f32x2_t* s1 = reinterpret_cast<const f32x2_t*>(src) + ((size_t)p * sh + sy1) * sw;
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const f32x2_t t00 = s0[sx], t10 = s1[sx];
    f32x2_t r0 = t00, r1 = t10;
    if (sx + 1 < sw) {
      const f32x2_t t01 = s0[sx + 1], t11 = s1[sx + 1];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        r0[c] = __fadd_rn(__fmul_rn(t00[c], a0), __fmul_rn(t01[c], a1));
        r1[c] = __fadd_rn(__fmul_rn(t10[c], a0), __fmul_rn(t11[c], a1));
      }
    }
    f32x2_t o;
#pragma unroll
    for (int c = 0; c < 2; ++c) o[c] = __fmul_rn(__fadd_rn(__fmul_rn(r0[c], b0), __fmul_rn(r1[c], b1)), mul);
    reinterpret_cast<f32x2_t*>(dst)[i] = o;
  }
}

// ---- host-side tables (same arithmetic as oracle/pv_oracle.c) -------------------------------------
static int host_cv_round(double v) { return (int)nearbyint(v); }

static void host_smooth_taps(int n, double sigma, FbTaps* t) {
  static const float tab3[] = {0.25f, 0.5f, 0.25f};
  static const float tab5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
  static const float tab7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
  static const float tab1[] = {1.f};
  const float* fixed = nullptr;
  if ((n & 1) && n <= 7 && sigma <= 0) fixed = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    double v = fixed ? (double)fixed[i] : exp(scale2X * x * x);
    t->k[i] = (float)v;
    sum += t->k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) t->k[i] = (float)(t->k[i] * sum);
  t->n = n;
}

static void host_window_taps(int winsize, FbTaps* t) {
  int m = winsize / 2;
  double sigma = m * 0.3, s = 1;
  t->k[0] = (float)s;
  for (int i = 1; i <= m; i++) {
    float v = (float)exp(-i * i / (2 * sigma * sigma));
    t->k[i] = v;
    s += v * 2;
  }
  s = 1. / s;
  for (int i = 0; i <= m; i++) t->k[i] = (float)(t->k[i] * s);
  t->n = m;
}

static void host_poly_tables(int n, double sigma, FbPoly* pk) {
  float gb[32], xgb[32], xxgb[32];
  float *g = gb + n, *xg = xgb + n, *xxg = xxgb + n;
  if (sigma < 1.1920929e-07) sigma = n * 0.3;
  double s = 0.;
  for (int x = -n; x <= n; x++) {
    g[x] = (float)exp(-x * x / (2 * sigma * sigma));
    s += g[x];
  }
  s = 1. / s;
  for (int x = -n; x <= n; x++) {
    g[x] = (float)(g[x] * s);
    xg[x] = (float)(x * g[x]);
    xxg[x] = (float)(x * x * g[x]);
  }
  double G[6][6] = {{0}};
  for (int y = -n; y <= n; y++)
    for (int x = -n; x <= n; x++) {
      G[0][0] += g[y] * g[x];
      G[1][1] += g[y] * g[x] * x * x;
      G[3][3] += g[y] * g[x] * x * x * x * x;
      G[5][5] += g[y] * g[x] * x * x * y * y;
    }
  G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
  G[4][4] = G[3][3];
  G[3][4] = G[4][3] = G[5][5];
  double A[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 12; ++j) A[i][j] = j < 6 ? G[i][j] : (j - 6 == i ? 1.0 : 0.0);
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 12; ++j) std::swap(A[c][j], A[p][j]);
    double d = 1.0 / A[c][c];
    for (int j = 0; j < 12; ++j) A[c][j] *= d;
    for (int r = 0; r < 6; ++r)
      if (r != c) {
        double f = A[r][c];
        if (f != 0.0)
          for (int j = 0; j < 12; ++j) A[r][j] -= f * A[c][j];
      }
  }
  pk->ig11 = A[1][7];
  pk->ig03 = A[0][9];
  pk->ig33 = A[3][9];
  pk->ig55 = A[5][11];
  pk->n = n;
  for (int k = 0; k <= n; ++k) {
    pk->g[k] = g[k];
    pk->xg[k] = xg[k];
    pk->xxg[k] = xxg[k];
  }
}

static int fb_num_levels(int h, int w, double pyr_scale, int levels) {
  int k;
  double scale;
  for (k = 0, scale = 1; k < levels; k++) {
    scale *= pyr_scale;
    if (w * scale < 32 || h * scale < 32) break;
  }
  return k;
}

// power-of-two scales of fb_prep_polyexp_mfma_kernel's operands, from the taps alone (the image is 8-bit)
static void host_polyexp_scales(const FbPoly& pk, int lh, int lw, FbPolyScales* sc) {
  auto pow2_below_2_13 = [](double bound) -> double {      // s with bound * s < 2^13 (a binade of margin for rounding)
    if (!(bound > 0)) return 1.0;
    int e;
    frexp(bound, &e);
    return ldexp(1.0, 13 - e);
  };
  double tU[3], sH[3];
  for (int j = 0; j < 3; ++j) {
    for (int dir = 0; dir < 2; ++dir) {
      const int n = dir ? lw : lh;
      double mx = 0, rs = 0;
      for (int y = 0; y < n; ++y) {
        double row[64] = {0};
        if (j == 0) row[y] += pk.g[0];
        for (int k = 1; k <= pk.n; ++k) {
          const double c = j == 0 ? pk.g[k] : (j == 1 ? pk.xg[k] : pk.xxg[k]);
          row[std::min(y + k, n - 1)] += c;
          row[std::max(y - k, 0)] += j == 1 ? -c : c;
        }
        double ssum = 0;
        for (int yp = 0; yp < n; ++yp) ssum += fabs(row[yp]), mx = std::max(mx, fabs(row[yp]));
        rs = std::max(rs, ssum);
      }
      if (dir == 0) {
        const double sV = pow2_below_2_13(mx);
        tU[j] = pow2_below_2_13(255.0 * rs);
        sc->sV[j] = (float)sV;
        sc->fU[j] = (float)(tU[j] / (64.0 * sV));
      } else {
        sH[j] = pow2_below_2_13(mx);
        sc->sH[j] = (float)sH[j];
      }
    }
  }
  // b1 = (H_g, U_0), b2 = (H_xg, U_0), b4 = (H_xxg, U_0), b3 = (H_g, U_1), b6 = (H_xg, U_1), b5 = (H_g, U_2)
  sc->c03_b1 = (float)(pk.ig03 / (sH[0] * tU[0]));
  sc->c11_b2 = (float)(pk.ig11 / (sH[1] * tU[0]));
  sc->c33_b4 = (float)(pk.ig33 / (sH[2] * tU[0]));
  sc->c11_b3 = (float)(pk.ig11 / (sH[0] * tU[1]));
  sc->c55_b6 = (float)(pk.ig55 / (sH[1] * tU[1]));
  sc->c33_b5 = (float)(pk.ig33 / (sH[0] * tU[2]));
}

struct FbLayout {
  size_t off_I, off_T, off_R, off_M, off_V, off_flowA, off_flowB, off_G, total;
};
static FbLayout fb_layout(long long n_pairs, int h, int w) {
  const size_t px = (size_t)h * w;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  FbLayout L;
  size_t o = 0;
  L.off_I = o; o = align(o + (size_t)n_pairs * 2 * px * 4);
  L.off_T = o; o = align(o + (size_t)n_pairs * 2 * px * 12);
  L.off_R = o; o = align(o + (size_t)n_pairs * 2 * px * 20);
  L.off_M = o; o = align(o + (size_t)n_pairs * px * 20);
  L.off_V = o; o = align(o + (size_t)n_pairs * (size_t)w * (size_t)((h + 31) / 32 * 32) * 20);   // frames: transposed, rows padded to a whole run
  L.off_flowA = o; o = align(o + (size_t)n_pairs * px * 8);
  L.off_flowB = o; o = align(o + (size_t)n_pairs * px * 8);
  L.off_G = o; o = align(o + 8 * 64 * 64 * sizeof(float));   // window matrices (2) and PolyExp matrices (6) of the current pyramid level
  L.total = o;
  return L;
}

static int fb_check_params(const pv_farneback_params* p, int h, int w) {
  PV_REQUIRE(p, PV_EINVAL, "pv_farneback: null params");
  PV_REQUIRE(p->flags == PV_OPTFLOW_FARNEBACK_GAUSSIAN, PV_EINVAL,
             "pv_farneback: only flags=OPTFLOW_FARNEBACK_GAUSSIAN (256) is built (got %d)", p->flags);
  PV_REQUIRE(p->poly_n == 5 || p->poly_n == 7, PV_EINVAL, "pv_farneback: poly_n must be 5 or 7");
  PV_REQUIRE(p->pyr_scale > 0 && p->pyr_scale < 1, PV_EINVAL, "pv_farneback: pyr_scale must be in (0,1)");
  PV_REQUIRE(p->winsize >= 2 && p->winsize / 2 <= 63, PV_ESIZE, "pv_farneback: winsize must be 2..127");
  PV_REQUIRE(p->iterations >= 1 && p->levels >= 0, PV_EINVAL, "pv_farneback: bad iterations/levels");
  PV_REQUIRE(h >= 2 && w >= 2, PV_ESIZE, "pv_farneback: image too small");
  return PV_OK;
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_farneback_workspace_bytes(int64_t n_pairs, int32_t h, int32_t w, const pv_farneback_params* params, size_t* bytes) {
  PV_REQUIRE(bytes && n_pairs >= 0, PV_EINVAL, "pv_farneback_workspace_bytes: bad arguments");
  int rc = fb_check_params(params, h, w);
  if (rc) return rc;
  *bytes = fb_layout(n_pairs, h, w).total;
  return PV_OK;
}

int pv_farneback_batch_u8(const uint8_t* prev, const uint8_t* next, int64_t prev_stride, int64_t next_stride,
                          int64_t pairs_per_group, int64_t group_stride, float* flow0, int64_t n_pairs, int32_t h, int32_t w, const pv_farneback_params* p,
                          void* workspace, size_t workspace_bytes, void* stream) {
  int rc = fb_check_params(p, h, w);
  if (rc) return rc;
  PV_REQUIRE(prev && next && flow0 && workspace, PV_EINVAL, "pv_farneback_batch_u8: null pointer");
  PV_REQUIRE(n_pairs >= 0, PV_EINVAL, "pv_farneback_batch_u8: negative n_pairs");
  if (pairs_per_group <= 0) { pairs_per_group = n_pairs > 0 ? n_pairs : 1; group_stride = 0; }
  if (n_pairs == 0) return PV_OK;
  FbLayout L = fb_layout(n_pairs, h, w);
  PV_REQUIRE(workspace_bytes >= L.total, PV_ESIZE, "pv_farneback_batch_u8: workspace %zu < %zu bytes", workspace_bytes,
             L.total);
  hipStream_t st = as_stream(stream);
  char* ws = (char*)workspace;
  float* I = (float*)(ws + L.off_I);
  float* T = (float*)(ws + L.off_T);
  float* R = (float*)(ws + L.off_R);
  float* M = (float*)(ws + L.off_M);
  float* V = (float*)(ws + L.off_V);
  float* flow_buf[2] = {(float*)(ws + L.off_flowA), (float*)(ws + L.off_flowB)};

  // consecutive frames of one stack per group: per-image work once per frame (fb_image_of)
  int chain_f = 0;
  if (next == prev + prev_stride && next_stride == prev_stride && n_pairs % pairs_per_group == 0 &&
      pairs_per_group + 1 <= 0x7fffffffLL && !getenv("PV_FARNEBACK_NO_FRAME_CHAIN"))
    chain_f = (int)(pairs_per_group + 1);
  const long long n_img = chain_f ? (n_pairs / pairs_per_group) * chain_f : n_pairs * 2;
  FbPoly pk;
  host_poly_tables(p->poly_n, p->poly_sigma, &pk);
  FbTaps win;
  host_window_taps(p->winsize, &win);

  const int levels = fb_num_levels(h, w, p->pyr_scale, p->levels);
  float* prev_flow = nullptr;
  int prev_w = 0, prev_h = 0;
  int pingpong = 0;
  for (int k = levels; k >= 0; --k) {
    double scale = 1;
    for (int i = 0; i < k; ++i) scale *= p->pyr_scale;
    double sigma = (1. / scale - 1) * 0.5;
    int smooth_sz = host_cv_round(sigma * 5) | 1;
    smooth_sz = std::max(smooth_sz, 3);
    PV_REQUIRE(smooth_sz <= 63, PV_ESIZE, "pv_farneback_batch_u8: pyramid too deep (smoothing kernel %d taps)", smooth_sz);
    const int lw = host_cv_round(w * scale), lh = host_cv_round(h * scale);
    PV_REQUIRE(lw >= 2 && lh >= 2, PV_ESIZE, "pv_farneback_batch_u8: level smaller than 2x2");
    const long long lpx = (long long)lw * lh;
    float* flow = k > 0 ? flow_buf[pingpong] : flow0;
    pingpong ^= 1;
    const bool coarse = k > 0;
    // the level's starting flow (zero, or the coarser level's result resized) is consumed only by the first
    // UpdateMatrices, which forms it on the fly; it is materialised only when no iteration follows to overwrite it
    const bool fuse_init = p->iterations >= 1;
    if (!fuse_init) {
      stage_mark(coarse ? "farneback.coarse.flow_init" : "farneback.level0.flow_upsample", st);
      if (!prev_flow) {
        hipError_t e = hipMemsetAsync(flow, 0, (size_t)n_pairs * lpx * 2 * sizeof(float), st);
        PV_REQUIRE(e == hipSuccess, PV_ELAUNCH, "pv_farneback_batch_u8: memset failed");
      } else {
        if ((long long)n_pairs * lpx < 0x7fffffffLL)
          hipLaunchKernelGGL(fb_flow_upsample_kernel<int>, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                             (const float*)prev_flow, flow, (long long)n_pairs, prev_h, prev_w, lh, lw, (double)prev_w / lw,
                             (double)prev_h / lh, (float)(1. / p->pyr_scale));
        else
          hipLaunchKernelGGL(fb_flow_upsample_kernel<long long>, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0,
                             st, (const float*)prev_flow, flow, (long long)n_pairs, prev_h, prev_w, lh, lw,
                             (double)prev_w / lw, (double)prev_h / lh, (float)(1. / p->pyr_scale));
      }
    }
    FbTaps sm;
    host_smooth_taps(smooth_sz, sigma, &sm);
    int mode;
    double inv_fx = (double)w / lw, inv_fy = (double)h / lh;
    if (lw == w && lh == h) mode = 0;
    else if (fabs(inv_fx - 2.0) < 2.220446049250313e-16 && fabs(inv_fy - 2.0) < 2.220446049250313e-16) mode = 1;
    else mode = 2;
    const bool tile_path = lw <= 64 && lh <= 64 && n_pairs <= 0x7fffffffLL;   // any window: it is a precomputed matrix
    // Levels up to 64 x 64 whose source images are up to 64 x 64 (the PV-site tiles): ONE launch per level does every
    // iteration -- UpdateMatrices, the window blur and the solve (fb_level_u_kernel); M is never written.  33..64-pixel
    // levels: one pair per 64 x 64 tile; levels up to 32 x 32: four pairs per tile (MOSAIC; its buffer offsets are 31-bit).
    // Rows of whole 4-pixel quads and a 16-byte-aligned flow: R and the flow move as 16-byte vectors.  Everything else (a
    // width that is no multiple of 4, larger source images, PV_FARNEBACK_TWO_LAUNCH_ITERATION=1) takes the two-launch
    // form below: UpdateMatrices writes M, the window blur + solve read it back.
    const bool small_level = lh <= 32 && lw <= 32;
    const bool polyexp_tile = h <= 64 && w <= 64 && smooth_sz <= 63;
    const bool fused_iter = tile_path && fuse_init && polyexp_tile && (lw & 3) == 0 && ((uintptr_t)flow & 15) == 0 &&
                            !getenv("PV_FARNEBACK_TWO_LAUNCH_ITERATION") &&
                            (!small_level || ((long long)n_img * lpx * 20 < 0x7fffffffLL && n_pairs * lpx * 8 < 0x7fffffffLL &&
                                              (long long)prev_h * prev_w * n_pairs * 8 < 0x7fffffffLL &&
                                              !getenv("PV_FARNEBACK_TWO_LAUNCH_SMALL_LEVELS")));
    const bool iter_v2 = fused_iter;      // R pair-planar for the level kernel, [h][w][5] for the two-launch form
    stage_mark(coarse ? "farneback.coarse.prep_polyexp" : "farneback.level0.prep_polyexp", st);
    // (its pre-filter stage is written for the 3-tap Gaussian, a copy or the exact 2 x 2 mean as the resize, and 8-byte words
    // of the source: any other case keeps the vector-ALU kernel)
    const bool polyexp_mfma = polyexp_tile && iter_v2 && smooth_sz == 3 && mode != 2 && w >= 2 && (h * w) % 8 == 0 &&
                              (((uintptr_t)prev | (uintptr_t)next | (uintptr_t)prev_stride | (uintptr_t)next_stride |
                                (uintptr_t)group_stride) & 7) == 0 &&
                              !getenv("PV_FARNEBACK_POLYEXP_VALU") && !getenv("PV_FARNEBACK_POLYEXP_F64");
    // frames: a tile of the level image per workgroup, its stages in LDS (fb_prep_polyexp_frame_kernel).  Levels at the source's
    // scale: one launch.  Coarse levels: smoothing + resize of tiles without halo -> I, then PolyExp from I (a tile + PolyExp halo
    // of a coarse level would filter a source window many times the tile).  Tiles: the largest of a short list whose windows fit
    // 150 KB (a level that samples the source very sparsely keeps the three kernels).
    FbFrameTile frame_tile = {0, 0, 0, 0, 0, 0}, frame_tile_poly = {0, 0, 0, 0, 0, 0};
    size_t frame_lds = 0, frame_lds_poly = 0;
    if (!polyexp_tile && !getenv("PV_FARNEBACK_THREE_KERNEL_POLYEXP")) {
      static const int cand[][2] = {{32, 64}, {16, 64}, {16, 32}, {8, 32}, {8, 16}};
      const int halo = mode == 0 ? pk.n : 0;
      // the largest tile whose windows leave room for two workgroups per CU (75 KB); failing that, the largest that fits at all
      for (size_t limit : {(size_t)75 * 1024, (size_t)150 * 1024}) {
        for (auto& c : cand) {
          const int ty = c[0], tx = c[1], IY = ty + 2 * halo, IX = tx + 2 * halo, r = smooth_sz >> 1;
          const int by = mode == 0 ? IY : (mode == 1 ? 2 * IY : (int)ceil(IY * inv_fy) + 3);
          const int bx = mode == 0 ? IX : (mode == 1 ? 2 * IX : (int)ceil(IX * inv_fx) + 3);
          const int cap_sy = by + 2 * r, cap_sx = std::max(bx + 2 * r, (IY * IX + by + 2 * r - 1) / (by + 2 * r));
          const size_t words = 2 * (size_t)cap_sy * cap_sx + (mode == 0 ? 3 * (size_t)ty * IX : 0);   // (the T planes: stage 0 only)
          if (words * 4 <= limit) {
            frame_tile = {ty, tx, (lh + ty - 1) / ty, (lw + tx - 1) / tx, cap_sy, cap_sx};
            frame_lds = words * 4;
            break;
          }
        }
        if (frame_tile.ty) break;
      }
      const int ty = 32, tx = 64, IY = ty + 2 * pk.n, IX = tx + 2 * pk.n;
      frame_tile_poly = {ty, tx, (lh + ty - 1) / ty, (lw + tx - 1) / tx, IY, IX};
      frame_lds_poly = (2 * (size_t)IY * IX + 3 * (size_t)ty * IX) * 4;
    }
    if (polyexp_mfma) {
      // the two PolyExp passes as nine products on the f16 matrix cores (fb_prep_polyexp_mfma_kernel)
      float* P6 = (float*)(ws + L.off_G) + 2 * 64 * 64;
      FbPolyScales sc;
      host_polyexp_scales(pk, lh, lw, &sc);
      hipLaunchKernelGGL(fb_polyexp_matrix_kernel, dim3(96), dim3(256), 0, st, P6, lh, lw, pk, small_level ? 1 : 0);
      const long long n_units = small_level ? (n_img + 3) / 4 : n_img;
      const unsigned grid = (unsigned)std::min<long long>(n_units, kNumCU);
      if (small_level)
        hipLaunchKernelGGL(fb_prep_polyexp_mfma_kernel<true>, dim3(grid), dim3(512), 0, st, prev, next, (long long)prev_stride,
                           (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, R, n_img, chain_f, h, w,
                           lh, lw, mode, inv_fx, inv_fy, sm, (const float*)P6, sc);
      else
        hipLaunchKernelGGL(fb_prep_polyexp_mfma_kernel<false>, dim3(grid), dim3(512), 0, st, prev, next, (long long)prev_stride,
                           (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, R, n_img, chain_f, h, w,
                           lh, lw, mode, inv_fx, inv_fy, sm, (const float*)P6, sc);
    } else if (polyexp_tile) {
      const unsigned grid = (unsigned)std::min<long long>(n_img, 4096);
      // (PV_FARNEBACK_POLYEXP_F64=1: the horizontal pass on the reference's double accumulators, as in rounds 1-3)
      if (getenv("PV_FARNEBACK_POLYEXP_F64"))
        hipLaunchKernelGGL(fb_prep_polyexp_tile_kernel<true>, dim3(grid), dim3(FB_PP_NT), 0, st, prev, next, (long long)prev_stride,
                           (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, R,
                           n_img, chain_f, h, w, lh, lw, mode, inv_fx, inv_fy, sm, pk, iter_v2 ? 1 : 0);
      else
        hipLaunchKernelGGL(fb_prep_polyexp_tile_kernel<false>, dim3(grid), dim3(FB_PP_NT), 0, st, prev, next, (long long)prev_stride,
                           (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, R,
                           n_img, chain_f, h, w, lh, lw, mode, inv_fx, inv_fy, sm, pk, iter_v2 ? 1 : 0);
    } else if (frame_tile.ty > 0) {
      auto launch = [&](const FbFrameTile& ft, size_t lds, int stage) {
        const long long n_tiles = n_img * ft.n_ty * ft.n_tx;
        if (hipFuncSetAttribute((const void*)fb_prep_polyexp_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)std::max(frame_lds, frame_lds_poly)) != hipSuccess) return false;
        hipLaunchKernelGGL(fb_prep_polyexp_frame_kernel, dim3((unsigned)std::min<long long>(n_tiles, 16 * kNumCU)), dim3(FB_FR_NT),
                           lds, st, prev, next, (long long)prev_stride, (long long)next_stride, (long long)pairs_per_group,
                           (long long)group_stride, R, n_img, chain_f, h, w, lh, lw, mode, inv_fx, inv_fy, sm, pk, ft, stage, I);
        return true;
      };
      const bool ok = mode == 0 ? launch(frame_tile, frame_lds, 0)
                                : launch(frame_tile, frame_lds, 1) && launch(frame_tile_poly, frame_lds_poly, 2);
      PV_REQUIRE(ok, PV_ELAUNCH, "pv_farneback_batch_u8: LDS size refused");
    } else {
    hipLaunchKernelGGL(fb_prep_kernel, dim3(stream_grid((size_t)(n_img * lpx), 256)), dim3(256), 0, st, prev, next,
                       (long long)prev_stride, (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, I,
                       n_img, chain_f, h, w, lh, lw, mode, inv_fx,
                       inv_fy, sm);
    hipLaunchKernelGGL(fb_polyexp_v_kernel, dim3(stream_grid((size_t)(n_img * lpx), 256)), dim3(256), 0, st,
                       (const float*)I, T, n_img, lh, lw, pk);
    hipLaunchKernelGGL(fb_polyexp_h_kernel, dim3(stream_grid((size_t)(n_img * lpx), 256)), dim3(256), 0, st,
                       (const float*)T, R, n_img, lh, lw, pk);
    }
    if (fused_iter) {
      float* Gv = (float*)(ws + L.off_G);
      float* Gh = lh == lw ? Gv : Gv + 64 * 64;
      stage_mark(coarse ? "farneback.coarse.iterations_fused" : "farneback.level0.iterations_fused", st);
      hipLaunchKernelGGL(fb_window_matrix_kernel, dim3(16), dim3(256), 0, st, Gv, lh, win, small_level ? 1 : 0);
      if (Gh != Gv) hipLaunchKernelGGL(fb_window_matrix_kernel, dim3(16), dim3(256), 0, st, Gh, lw, win, small_level ? 1 : 0);
      FbUpsample up = {prev_h, prev_w, prev_flow ? (double)prev_w / lw : 1.0, prev_flow ? (double)prev_h / lh : 1.0,
                       (float)(1. / p->pyr_scale)};
      const long long n_units = small_level ? (n_pairs + 3) / 4 : n_pairs;
      const unsigned grid = (unsigned)std::min<long long>(n_units, kNumCU);      // one workgroup per CU (LDS)
#define PV_LEVEL_U(INIT, MOS)                                                                                             \
  hipLaunchKernelGGL((fb_level_u_kernel<INIT, MOS>), dim3(grid), dim3(512), 0, st, (const float*)R,                       \
                     (const float*)prev_flow, flow, (const float*)Gv, (const float*)Gh, lh, lw, (long long)n_pairs,       \
                     (long long)pairs_per_group, chain_f, up, (int)p->iterations)
      if (prev_flow) { if (small_level) PV_LEVEL_U(1, true); else PV_LEVEL_U(1, false); }
      else { if (small_level) PV_LEVEL_U(2, true); else PV_LEVEL_U(2, false); }
#undef PV_LEVEL_U
      prev_flow = flow;
      prev_w = lw;
      prev_h = lh;
      continue;
    }
    stage_mark(coarse ? "farneback.coarse.update_matrices" : "farneback.level0.update_matrices", st);
    {
      const dim3 um_grid(stream_grid((size_t)(n_pairs * lpx), 256));
      FbUpsample up = {prev_h, prev_w, prev_flow ? (double)prev_w / lw : 1.0, prev_flow ? (double)prev_h / lh : 1.0,
                       (float)(1. / p->pyr_scale)};
      if (!fuse_init)
        hipLaunchKernelGGL(fb_update_matrices_kernel<0>, um_grid, dim3(256), 0, st, (const float*)R, (const float*)flow, M,
                           (long long)n_pairs, lh, lw, tile_path ? 1 : 0, (long long)pairs_per_group, chain_f, up);
      else if (prev_flow)
        hipLaunchKernelGGL(fb_update_matrices_kernel<1>, um_grid, dim3(256), 0, st, (const float*)R, (const float*)prev_flow, M,
                           (long long)n_pairs, lh, lw, tile_path ? 1 : 0, (long long)pairs_per_group, chain_f, up);
      else
        hipLaunchKernelGGL(fb_update_matrices_kernel<2>, um_grid, dim3(256), 0, st, (const float*)R, (const float*)nullptr, M,
                           (long long)n_pairs, lh, lw, tile_path ? 1 : 0, (long long)pairs_per_group, chain_f, up);
    }
    for (int it = 0; it < p->iterations; ++it) {
      const int update = it < p->iterations - 1 ? 1 : 0;
      if (tile_path) {
        // window blur + solve on the matrix cores (UpdateMatrices stays a separate high-occupancy launch: its
        // flow-dependent R1 gathers need many waves in flight)
        float* Gv = (float*)(ws + L.off_G);
        float* Gh = lh == lw ? Gv : Gv + 64 * 64;   // square levels: the vertical and horizontal window matrices coincide
        stage_mark(coarse ? "farneback.coarse.window_blur_solve" : "farneback.level0.window_blur_solve", st);
        if (it == 0) {
          hipLaunchKernelGGL(fb_window_matrix_kernel, dim3(16), dim3(256), 0, st, Gv, lh, win, 0);
          if (Gh != Gv) hipLaunchKernelGGL(fb_window_matrix_kernel, dim3(16), dim3(256), 0, st, Gh, lw, win, 0);
        }
        if (lh <= 32 && lw <= 32) {
          const unsigned grid = (unsigned)std::min<long long>((n_pairs + 3) / 4, 2048);
          hipLaunchKernelGGL(fb_tile_mfma_kernel<1>, dim3(grid), dim3(256), 0, st, (const float*)M, (const float*)Gv,
                             (const float*)Gh, flow, lh, lw, (long long)n_pairs);
        } else {
          // (width % 4 and the alignment of M decide between the quadrant kernel's 16-byte staging and the strip kernel)
          if ((lw & 3) == 0 && ((uintptr_t)M & 15) == 0 && !getenv("PV_FARNEBACK_STRIP_KERNEL")) {
            const unsigned grid = (unsigned)std::min<long long>(n_pairs, 4096);
            hipLaunchKernelGGL(fb_tile_mfma_q_kernel, dim3(grid), dim3(256), 0, st, (const float*)M, (const float*)Gv,
                               (const float*)Gh, flow, lh, lw, (long long)n_pairs);
          } else {
            const unsigned grid = (unsigned)std::min<long long>((n_pairs + 1) / 2, 2048);
            hipLaunchKernelGGL(fb_tile_mfma_kernel<2>, dim3(grid), dim3(256), 0, st, (const float*)M, (const float*)Gv,
                               (const float*)Gh, flow, lh, lw, (long long)n_pairs);
          }
        }
        if (update) {
          stage_mark(coarse ? "farneback.coarse.update_matrices" : "farneback.level0.update_matrices", st);
          hipLaunchKernelGGL(fb_update_matrices_kernel<0>, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                             (const float*)R, (const float*)flow, M, (long long)n_pairs, lh, lw, 1, (long long)pairs_per_group,
                             chain_f, FbUpsample{});
        }
        continue;
      }
      stage_mark(coarse ? "farneback.coarse.window_blur_solve" : "farneback.level0.window_blur_solve", st);
      // the reference's 41-tap window (winsize 40): register-window passes; any other window: one load per tap
      const int hp = (lh + FB_RUN_V - 1) / FB_RUN_V * FB_RUN_V;
      const long long v_runs = (lh + FB_RUN_V - 1) / FB_RUN_V, v_xb = (lw * 5 + 255) / 256;
      const long long h_runs = (lw + FB_RUN_H - 1) / FB_RUN_H, h_yb = (lh + 63) / 64;
      const long long v_strips = n_pairs * v_xb, h_strips = n_pairs * h_yb;
      const long long v_blocks = (v_strips + 7) / 8 * 8 * v_runs, h_blocks = (h_strips + 7) / 8 * 8 * h_runs;
      if (win.n == 20 && ((uintptr_t)V & 15) == 0 && ((uintptr_t)flow & 7) == 0 && v_blocks < 0x7fffffffLL &&
          h_blocks < 0x7fffffffLL && 5LL * lw * hp < 0x7fffffffLL && !getenv("PV_FARNEBACK_TAP_LOADS")) {
        hipLaunchKernelGGL((fb_blur_v_run_kernel<20, FB_RUN_V>), dim3((unsigned)v_blocks), dim3(256), 0, st, (const float*)M, V,
                           lh, lw, hp, (int)v_runs, (int)v_xb, v_strips, win);
        hipLaunchKernelGGL((fb_blur_h_solve_run_kernel<20, FB_RUN_H>), dim3((unsigned)h_blocks), dim3(64), 0, st,
                           (const float*)V, flow, lh, lw, hp, (int)h_runs, (int)h_yb, h_strips, win);
      } else {
      hipLaunchKernelGGL(fb_blur_v_kernel, dim3(stream_grid((size_t)(n_pairs * lpx * 5), 256)), dim3(256), 0, st,
                         (const float*)M, V, (long long)n_pairs, lh, lw, win);
      hipLaunchKernelGGL(fb_blur_h_solve_kernel, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                         (const float*)V, flow, (long long)n_pairs, lh, lw, win);
      }
      if (update) {
        stage_mark(coarse ? "farneback.coarse.update_matrices" : "farneback.level0.update_matrices", st);
        hipLaunchKernelGGL(fb_update_matrices_kernel<0>, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                           (const float*)R, (const float*)flow, M, (long long)n_pairs, lh, lw, 0, (long long)pairs_per_group,
                           chain_f, FbUpsample{});
      }
    }
    prev_flow = flow;
    prev_w = lw;
    prev_h = lh;
  }
  stage_mark(nullptr, st);
  return check_launch("pv_farneback_batch_u8");
}

}  // extern "C"

#ifdef PV_DIAG_STAMPS
extern "C" int pv_diag_read_fb_polyexp(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::fb_polyexp_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int pv_diag_read_fb_iter(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(pv::fb_iter_diag), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
