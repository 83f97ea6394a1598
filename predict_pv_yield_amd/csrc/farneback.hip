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
#include "pv_common.h"

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

// images: pair p has prev at prev + p*prev_stride, next at next + p*next_stride.  I: [n_pairs][2][lh][lw]
__global__ __launch_bounds__(256) void fb_prep_kernel(const uint8_t* __restrict__ prev, const uint8_t* __restrict__ next,
                                                       long long prev_stride, long long next_stride,
                                                       long long pairs_per_group, long long group_stride,
                                                       float* __restrict__ I, long long n_pairs, int h, int w, int lh,
                                                       int lw, int mode /*0 copy, 1 area 2x2, 2 bilinear*/,
                                                       double inv_fx, double inv_fy, FbTaps kt) {
  const long long per_img = (long long)lh * lw;
  const long long total = n_pairs * 2 * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long im = i / per_img;
    int rem = (int)(i - im * per_img);
    int y = rem / lw, x = rem - y * lw;
    long long p = im >> 1;
    const long long grp = p / pairs_per_group, q = p - grp * pairs_per_group;
    const uint8_t* img = ((im & 1) ? next + q * next_stride : prev + q * prev_stride) + grp * group_stride;
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
// 16-wave workgroups share a CU, so one's barrier-separated phases overlap the other's.
constexpr int FB_PP_NT = 1024;
__global__ __launch_bounds__(FB_PP_NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void fb_prep_polyexp_tile_kernel(const uint8_t* __restrict__ prev,
                                                                    const uint8_t* __restrict__ next, long long prev_stride,
                                                                    long long next_stride, long long pairs_per_group,
                                                                    long long group_stride, float* __restrict__ R,
                                                                    long long n_img, int h, int w, int lh, int lw, int mode,
                                                                    double inv_fx, double inv_fy, FbTaps kt, FbPoly pk) {
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
    const long long p = im >> 1;
    const long long grp = p / pairs_per_group, q = p - grp * pairs_per_group;
    const uint8_t* img = ((im & 1) ? next + q * next_stride : prev + q * prev_stride) + grp * group_stride;
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
    // PolyExp, horizontal pass (double accumulators) -> R
    for (int i = tid; i < lpx; i += FB_PP_NT) {
      const int y = row_lw(i), x = i - y * lw;
      const float* t0r = bufA + y * lw;
      const float* t1r = Tt12 + y * lw;
      const float* t2r = Tt12 + lpx + y * lw;
      float g0 = pk.g[0];
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
      float* d = R + (im * lpx + i) * 5;
      d[1] = (float)__dmul_rn(b2, pk.ig11);
      d[0] = (float)__dmul_rn(b3, pk.ig11);
      d[3] = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b4, pk.ig33));
      d[2] = (float)__dadd_rn(__dmul_rn(b1, pk.ig03), __dmul_rn(b5, pk.ig33));
      d[4] = (float)__dmul_rn(b6, pk.ig55);
    }
    __syncthreads();   // LDS is reused by the next image
  }
}

// ---- UpdateMatrices --------------------------------------------------------------------------------
// one pixel of FarnebackUpdateMatrices: R0 = this pixel's 5 coefficients, R1 = base of the second image's
// coefficient plane, (dx, dy) = current flow; out = (G11, G12, G22, h1, h2)
__device__ __forceinline__ void fb_update_pixel(const float* __restrict__ R0, const float* __restrict__ R1, float dx,
                                                float dy, int x, int y, int width, int height, float* m) {
  const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
  const size_t step1 = (size_t)width * 5;
  float fx = __fadd_rn((float)x, dx), fy = __fadd_rn((float)y, dy);
  int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
  float r2, r3, r4, r5, r6;
  fx = __fsub_rn(fx, (float)x1);
  fy = __fsub_rn(fy, (float)y1);
  if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
    const float* ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
    float a00 = __fmul_rn(1.f - fx, 1.f - fy), a01 = __fmul_rn(fx, 1.f - fy);
    float a10 = __fmul_rn(1.f - fx, fy), a11 = __fmul_rn(fx, fy);
#define PV_BILIN(c) \
  __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a00, ptr[c]), __fmul_rn(a01, ptr[5 + c])), __fmul_rn(a10, ptr[step1 + c])), \
            __fmul_rn(a11, ptr[step1 + 5 + c]))
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
    float scale = (x < 5 ? border[x] : 1.f);
    scale = __fmul_rn(scale, (x >= width - 5 ? border[width - x - 1] : 1.f));
    scale = __fmul_rn(scale, (y < 5 ? border[y] : 1.f));
    scale = __fmul_rn(scale, (y >= height - 5 ? border[height - y - 1] : 1.f));
    r2 = __fmul_rn(r2, scale); r3 = __fmul_rn(r3, scale); r4 = __fmul_rn(r4, scale);
    r5 = __fmul_rn(r5, scale); r6 = __fmul_rn(r6, scale);
  }
  m[0] = __fadd_rn(__fmul_rn(r4, r4), __fmul_rn(r6, r6));
  m[1] = __fmul_rn(__fadd_rn(r4, r5), r6);
  m[2] = __fadd_rn(__fmul_rn(r5, r5), __fmul_rn(r6, r6));
  m[3] = __fadd_rn(__fmul_rn(r4, r2), __fmul_rn(r6, r3));
  m[4] = __fadd_rn(__fmul_rn(r6, r2), __fmul_rn(r5, r3));
}

// R: [n_pairs][2][lh][lw][5] (R0 = image 0, R1 = image 1); flow: [n_pairs][lh][lw][2];
// M: [n_pairs][lh][lw][5] (planar == 0) or [n_pairs][5][lh][lw] (planar != 0, what the fused tile kernel reads)
__global__ __launch_bounds__(256) void fb_update_matrices_kernel(const float* __restrict__ R, const float* __restrict__ flow,
                                                                  float* __restrict__ M, long long n_pairs, int height,
                                                                  int width, int planar) {
  const long long per_img = (long long)height * width;
  const long long total = n_pairs * per_img;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long p = i / per_img;
    int rem = (int)(i - p * per_img);
    int y = rem / width, x = rem - y * width;
    const float* R0 = R + ((p * 2) * per_img + rem) * 5;
    const float* R1 = R + (p * 2 + 1) * per_img * 5;
    const float* fl = flow + i * 2;
    float m[5];
    fb_update_pixel(R0, R1, fl[0], fl[1], x, y, width, height, m);
    if (planar) {
#pragma unroll
      for (int c = 0; c < 5; ++c) M[(p * 5 + c) * per_img + rem] = m[c];
    } else {
      float* d = M + i * 5;
#pragma unroll
      for (int c = 0; c < 5; ++c) d[c] = m[c];
    }
  }
}

// ---- fused iteration for levels that fit one workgroup (<= 64 x 64): window blur (vertical + horizontal) of the 5
// channels through LDS, 2x2 solve, and the UpdateMatrices of the next iteration, one workgroup per image pair.
// M is planar [pair][5][lh][lw].  Thread mapping: vertical pass (x = tid & 63, 16 rows), result written TRANSPOSED
// (padded to 65) so that the horizontal pass is again a per-thread sliding window (row y = tid & 63, 16 columns).
// Summation order is FarnebackUpdateFlow_GaussianBlur's (centre tap, then pairs outward); the multiply-adds are fused
// (one rounding instead of two per tap: ~1e-7 relative, far inside the 1e-3 px parity bar) to halve the VALU work.
template <int MW, int SEG>  // MW = winsize / 2; SEG = outputs per thread; 64 * (64 / SEG) threads
__global__ __launch_bounds__(64 * (64 / SEG)) void fb_tile_iter_kernel(const float* __restrict__ Min, float* __restrict__ Mout,
                                                            const float* __restrict__ R, float* __restrict__ flow,
                                                            int height, int width, FbTaps kt, int update) {
  constexpr int TS = 64, WIN = SEG + 2 * MW, NT = 64 * (64 / SEG);
  __shared__ float A[TS * TS];
  __shared__ float Bt[TS * (TS + 1)];
  const long long p = blockIdx.x;
  const long long per_img = (long long)height * width;
  const int tid = threadIdx.x;
  const int lane64 = tid & 63, seg = tid >> 6;
  float tap[MW + 1];
#pragma unroll
  for (int k = 0; k <= MW; ++k) tap[k] = kt.k[k];
  float hres[5][SEG];
  constexpr int NE = TS * TS / NT;  // tile elements per thread
  const int npx = height * width;
  float nxt[NE];
  {
    const float* src = Min + (p * 5) * per_img;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * NT;
      nxt[e] = i < npx ? src[i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * NT;
      if (i < npx) { const int y = i / width; A[y * TS + (i - y * width)] = nxt[e]; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 5; ++c) {  // unrolled: hres must be indexed statically to stay in registers
    if (c < 4) {  // prefetch the next channel's plane under this channel's arithmetic
      const float* src = Min + (p * 5 + c + 1) * per_img;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int i = tid + e * NT;
        nxt[e] = i < npx ? src[i] : 0.f;
      }
    }
    // vertical blur: column x = lane64, rows seg*16 .. +15
    if (lane64 < width) {
      float win[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) {
        int y = seg * SEG - MW + i;
        y = y < 0 ? 0 : (y > height - 1 ? height - 1 : y);
        win[i] = A[y * TS + lane64];
      }
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        float s0 = __fmul_rn(win[j + MW], tap[0]);
#pragma unroll
        for (int k = 1; k <= MW; ++k) s0 = fmaf(__fadd_rn(win[j + MW + k], win[j + MW - k]), tap[k], s0);
        const int y = seg * SEG + j;
        if (y < height) Bt[lane64 * (TS + 1) + y] = s0;
      }
    }
    __syncthreads();  // vertical results visible; every read of A is done
    if (c < 4) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int i = tid + e * NT;
        if (i < npx) { const int y = i / width; A[y * TS + (i - y * width)] = nxt[e]; }
      }
    }
    // horizontal blur: row y = lane64, columns seg*16 .. +15 (reads the transposed image: conflict-free)
    if (lane64 < height) {
      float win[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) {
        int x = seg * SEG - MW + i;
        x = x < 0 ? 0 : (x > width - 1 ? width - 1 : x);
        win[i] = Bt[x * (TS + 1) + lane64];
      }
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        float s0 = __fmul_rn(win[j + MW], tap[0]);
#pragma unroll
        for (int k = 1; k <= MW; ++k) s0 = fmaf(tap[k], __fadd_rn(win[j + MW - k], win[j + MW + k]), s0);
        hres[c][j] = s0;
      }
    }
    __syncthreads();  // Bt may be overwritten by the next channel; the new A plane is visible
  }
  if (lane64 < height) {
    const int y = lane64;
    const float* R1 = R + (p * 2 + 1) * per_img * 5;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const int x = seg * SEG + j;
      if (x < width) {
        double g11 = hres[0][j], g12 = hres[1][j], g22 = hres[2][j], h1 = hres[3][j], h2 = hres[4][j];
        double det = __dadd_rn(__dsub_rn(__dmul_rn(g11, g22), __dmul_rn(g12, g12)), 1e-3);
        double idet = __ddiv_rn(1.0, det);
        const float fxv = (float)__dmul_rn(__dsub_rn(__dmul_rn(g11, h2), __dmul_rn(g12, h1)), idet);
        const float fyv = (float)__dmul_rn(__dsub_rn(__dmul_rn(g22, h1), __dmul_rn(g12, h2)), idet);
        const long long pix = (long long)y * width + x;
        float* fl = flow + (p * per_img + pix) * 2;
        fl[0] = fxv;
        fl[1] = fyv;
        if (update) {
          float m[5];
          fb_update_pixel(R + ((p * 2) * per_img + pix) * 5, R1, fxv, fyv, x, y, width, height, m);
#pragma unroll
          for (int c = 0; c < 5; ++c) Mout[(p * 5 + c) * per_img + pix] = m[c];
        }
      }
    }
  }
}

// Packed-f32 version of the tile kernel (v_pk_add_f32 / v_pk_fma_f32: two channels per VALU instruction).  Plain f32
// adds / fmas issue at half the rate of the packed forms on CDNA3/4, and the first tile kernel spent as many
// instructions on clamped window addresses as on arithmetic (ISA: 3240 math + ~3000 address ops per thread).  Here
//   * channels are processed in PAIRS (g11|g12, g22|h1, h2|-) as float2 lanes of the same instruction,
//   * both LDS images are stored with REPLICATED BORDERS (MW rows above / below the tile for the vertical pass, MW
//     transposed rows left / right for the horizontal pass), so every window element is one ds_read_b64 at a constant
//     offset from a per-thread base: no clamps, no address arithmetic in the loops.
// Same summation order and the same single rounding per fused multiply-add as fb_tile_iter_kernel.
typedef float fb_f2 __attribute__((ext_vector_type(2)));

template <int MW, int SEG>
__global__ __launch_bounds__(64 * (64 / SEG)) void fb_tile_iter_pk_kernel(const float* __restrict__ Min,
                                                                         const float* __restrict__ R,
                                                                         float* __restrict__ flow, int height, int width,
                                                                         FbTaps kt, long long n_pairs) {
  constexpr int TS = 64, WIN = SEG + 2 * MW, NT = 64 * (64 / SEG), PR = TS + 2 * MW, TP = TS + 1;
  __shared__ fb_f2 A2[PR * TS];   // rows -MW .. TS+MW-1 (replicated above row 0 / below row height-1)
  __shared__ fb_f2 B2[PR * TP];   // transposed: rows = x + MW (replicated left of x = 0 / right of x = width-1), cols = y
  const long long per_img = (long long)height * width;
  const int tid = threadIdx.x;
  const int lane64 = tid & 63, seg = tid >> 6;
  float tap[MW + 1];
#pragma unroll
  for (int k = 0; k <= MW; ++k) tap[k] = kt.k[k];
  fb_f2 hres[3][SEG];
  constexpr int NE = TS * TS / NT;
  const int npx = height * width;
  fb_f2 nxt[NE];
  auto load_pair = [&](long long p, int cp) {
    const float* s0 = Min + (p * 5 + 2 * cp) * per_img;
    const float* s1 = Min + (p * 5 + (2 * cp + 1 < 5 ? 2 * cp + 1 : 4)) * per_img;   // channel 4 is paired with itself
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * NT;
      nxt[e] = i < npx ? (fb_f2){s0[i], s1[i]} : (fb_f2){0.f, 0.f};
    }
  };
  auto store_pair = [&]() {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = tid + e * NT;
      if (i < npx) {
        const int y = i / width, x = i - y * width;
        A2[(y + MW) * TS + x] = nxt[e];
        if (y == 0) {
#pragma unroll
          for (int r = 0; r < MW; ++r) A2[r * TS + x] = nxt[e];
        }
        if (y == height - 1) {
#pragma unroll
          for (int r = 0; r < MW; ++r) A2[(height + MW + r) * TS + x] = nxt[e];
        }
      }
    }
  };
  // persistent workgroups: each walks image pairs blockIdx.x, += gridDim.x, and the first planes of its NEXT image pair
  // are fetched under the last channel pair of the current one (one resident workgroup per CU: nothing else would hide
  // that latency)
  load_pair(blockIdx.x, 0);
  for (long long p = blockIdx.x; p < n_pairs; p += gridDim.x) {
  store_pair();
  __syncthreads();
#pragma unroll
  for (int cp = 0; cp < 3; ++cp) {
    if (cp < 2) load_pair(p, cp + 1);   // next channel pair's planes in flight under this pair's arithmetic
    else if (p + gridDim.x < n_pairs) load_pair(p + gridDim.x, 0);
    // vertical blur: column x = lane64, output rows seg*SEG .. +SEG-1; window rows at constant offsets
    if (lane64 < width) {
      const fb_f2* base = A2 + (seg * SEG) * TS + lane64;
      fb_f2 win[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) win[i] = base[i * TS];
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        fb_f2 s0 = win[j + MW] * tap[0];
#pragma unroll
        for (int k = 1; k <= MW; ++k)
          s0 = __builtin_elementwise_fma(win[j + MW + k] + win[j + MW - k], (fb_f2){tap[k], tap[k]}, s0);
        const int y = seg * SEG + j;
        if (y < height) B2[(lane64 + MW) * TP + y] = s0;
      }
    }
    __syncthreads();   // vertical results visible; every read of A2 is done
    if (cp < 2) store_pair();
    // replicate the first / last transposed rows MW times on either side
    for (int idx = tid; idx < 2 * MW * height; idx += NT) {
      const int side = idx / (MW * height), rem = idx - side * (MW * height);
      const int r = rem / height, y = rem - r * height;
      if (side == 0) B2[r * TP + y] = B2[MW * TP + y];
      else B2[(width + MW + r) * TP + y] = B2[(width - 1 + MW) * TP + y];
    }
    __syncthreads();
    // horizontal blur: row y = lane64, output columns seg*SEG .. +SEG-1
    if (lane64 < height) {
      const fb_f2* base = B2 + (seg * SEG) * TP + lane64;
      fb_f2 win[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) win[i] = base[i * TP];
#pragma unroll
      for (int j = 0; j < SEG; ++j) {
        fb_f2 s0 = win[j + MW] * tap[0];
#pragma unroll
        for (int k = 1; k <= MW; ++k)
          s0 = __builtin_elementwise_fma((fb_f2){tap[k], tap[k]}, win[j + MW - k] + win[j + MW + k], s0);
        hres[cp][j] = s0;
      }
    }
    __syncthreads();   // B2 may be overwritten by the next pair; the new A2 planes are visible
  }
  if (lane64 < height) {
    const int y = lane64;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
      const int x = seg * SEG + j;
      if (x < width) {
        double g11 = hres[0][j].x, g12 = hres[0][j].y, g22 = hres[1][j].x, h1 = hres[1][j].y, h2 = hres[2][j].x;
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
  }  // image pairs
}

// ---- window blur on the f32 matrix cores ----------------------------------------------------------------------------
// For images up to 64 x 64 the separable, border-replicated window blur is two small matrix products per channel,
//   U = X Gh^T (along x),  Out = Gv U (along y),   G[y][y'] = sum of the taps k with clamp(y + k) == y'
// (a banded 64 x 64 matrix that already contains the border replication), evaluated with v_mfma_f32_32x32x2_f32: exact
// f32 products, f32 accumulation; only the summation order differs from the tap loop (~1e-7 relative, like the fused
// multiply-adds of the VALU kernels; the parity bar is 1e-3 px).  640 MFMAs replace ~2800 VALU instructions per wave.
// Workgroup = 4 waves.  Images wider / taller than 32: one image pair per workgroup, wave (yt, xt) owns a 32 x 32 output
// quadrant (the horizontal pass of an x half is computed by both waves that need it).  Images up to 32 x 32 (the coarse
// pyramid level): one image pair PER WAVE.  The horizontal result stays in the accumulator registers and is consumed as
// the B operand of the vertical pass (its contraction index simply follows the accumulator's row order).
typedef float fb_v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void fb_window_matrix_kernel(float* __restrict__ G, int n, FbTaps kt) {
  // G[64][64]; rows / columns >= n stay zero
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * 64; i += gridDim.x * blockDim.x) {
    const int y = i >> 6, yp = i & 63;
    float s = 0.f;
    if (y < n && yp < n) {
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

template <bool SMALL>   // SMALL: height, width <= 32, one image pair per wave
__global__ __launch_bounds__(256) void fb_tile_mfma_kernel(const float* __restrict__ Min, const float* __restrict__ Gv,
                                                            const float* __restrict__ Gh, float* __restrict__ flow,
                                                            int height, int width, long long n_pairs) {
  constexpr int XS = 65;                        // LDS row stride (words)
  constexpr int IMG = SMALL ? 32 * XS : 64 * XS;
  __shared__ float Xs[SMALL ? 4 * IMG : 2 * IMG];   // SMALL: one image per wave; else: double-buffered channel image
  __shared__ float Us[SMALL ? 1 : IMG];             // horizontal result of the current channel (4 tiles, one per wave)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int xt = SMALL ? 0 : (wave & 1), yt = SMALL ? 0 : (wave >> 1);
  const long long per_img = (long long)height * width;
  // operands that do not change: Gh^T as B operand (k = x', column x), Gv as A operand (row y, k = accumulator row order)
  float ghreg[32], gvreg[2][16];
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) ghreg[kk] = Gh[(32 * xt + col) * 64 + 2 * kk + half];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) gvreg[a][kk] = Gv[(32 * yt + col) * 64 + 32 * a + fb_acc_row(kk, half)];
  float gvnat[32];   // !SMALL: Gv as A operand with the contraction index in natural order (k = y' = 2 kk + half)
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) gvnat[kk] = SMALL ? 0.f : Gv[(32 * yt + col) * 64 + 2 * kk + half];

  const long long groups = SMALL ? (n_pairs + 3) / 4 : n_pairs;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long p = SMALL ? grp * 4 + wave : grp;
    const bool p_ok = p < n_pairs;
    fb_v16f res[5];
    float* Xw = SMALL ? Xs + wave * IMG : Xs;
    float stg[16];   // one channel image in flight: the loads of channel c+1 are issued before the MFMAs of channel c
    auto load_channel = [&](int c) {
      const float* src = Min + ((p_ok ? p : 0) * 5 + c) * per_img;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = SMALL ? lane + 64 * e : tid + 256 * e;
        const int y = SMALL ? i >> 5 : i >> 6, x = SMALL ? i & 31 : i & 63;
        stg[e] = (p_ok && y < height && x < width) ? src[(long long)y * width + x] : 0.f;
      }
    };
    load_channel(0);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      // ---- channel c (zero padded to the tile) into LDS; next channel's loads start right away ---------------------------
      float* Xc = SMALL ? Xw : Xs + (c & 1) * IMG;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = SMALL ? lane + 64 * e : tid + 256 * e;
        if (SMALL) Xc[(i >> 5) * XS + (i & 31)] = stg[e];
        else Xc[(i >> 6) * XS + (i & 63)] = stg[e];
      }
      if (!SMALL) __syncthreads();   // image visible to the 4 waves (double buffered: the previous channel may still be read)
      if (c < 4) load_channel(c + 1);
      fb_v16f o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      if constexpr (SMALL) {
        // ---- horizontal: U[y'][x] = sum_x' X[y'][x'] Gh[x][x'] (x' < 32) -------------------------------------------------
        fb_v16f u;
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = 0.f;
        const float* xa = Xc + col * XS + half;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) u = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * kk], ghreg[kk], u, 0, 0, 0);
        // ---- vertical: Out[y][x] = sum_y' Gv[y][y'] U[y'][x]; contraction slot kk of lane-half h is row fb_acc_row(kk, h):
        // the horizontal result goes from the accumulator straight into the B operand
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(gvreg[0][kk], u[kk], o, 0, 0, 0);
      } else {
        // ---- horizontal: this wave's 32 x 32 tile U[32 yt + y'][32 xt + x], shared with the other waves through LDS ------
        fb_v16f u;
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = 0.f;
        const float* xa = Xc + (32 * yt + col) * XS + half;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) u = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * kk], ghreg[kk], u, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Us[(32 * yt + fb_acc_row(r, half)) * XS + 32 * xt + col] = u[r];
        __syncthreads();   // all four U tiles visible
        // ---- vertical: Out[y][x] = sum_y' Gv[y][y'] U[y'][x], y' = 0..63 in natural order ----------------------------------
        const float* ub = Us + half * XS + 32 * xt + col;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(gvnat[kk], ub[2 * kk * XS], o, 0, 0, 0);
      }
      res[c] = o;
    }
    if (!SMALL) __syncthreads();   // both image buffers free before the next pair's channel 0 / 1 are staged
    // ---- 2x2 solve; accumulator register r = row y, lane = column x: coalesced flow rows ---------------------------------
    if (p_ok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int y = 32 * yt + fb_acc_row(r, half), x = 32 * xt + col;
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
  L.off_V = o; o = align(o + (size_t)n_pairs * px * 20);
  L.off_flowA = o; o = align(o + (size_t)n_pairs * px * 8);
  L.off_flowB = o; o = align(o + (size_t)n_pairs * px * 8);
  L.off_G = o; o = align(o + 2 * 64 * 64 * sizeof(float));   // window matrices of the current pyramid level
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
    FbTaps sm;
    host_smooth_taps(smooth_sz, sigma, &sm);
    int mode;
    double inv_fx = (double)w / lw, inv_fy = (double)h / lh;
    if (lw == w && lh == h) mode = 0;
    else if (fabs(inv_fx - 2.0) < 2.220446049250313e-16 && fabs(inv_fy - 2.0) < 2.220446049250313e-16) mode = 1;
    else mode = 2;
    stage_mark(coarse ? "farneback.coarse.prep_polyexp" : "farneback.level0.prep_polyexp", st);
    if (h <= 64 && w <= 64 && smooth_sz <= 63) {
      const unsigned grid = (unsigned)std::min<long long>(n_pairs * 2, 4096);
      hipLaunchKernelGGL(fb_prep_polyexp_tile_kernel, dim3(grid), dim3(FB_PP_NT), 0, st, prev, next, (long long)prev_stride,
                         (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, R,
                         (long long)n_pairs * 2, h, w, lh, lw, mode, inv_fx, inv_fy, sm, pk);
    } else {
    hipLaunchKernelGGL(fb_prep_kernel, dim3(stream_grid((size_t)(n_pairs * 2 * lpx), 256)), dim3(256), 0, st, prev, next,
                       (long long)prev_stride, (long long)next_stride, (long long)pairs_per_group, (long long)group_stride, I,
                       (long long)n_pairs, h, w, lh, lw, mode, inv_fx,
                       inv_fy, sm);
    hipLaunchKernelGGL(fb_polyexp_v_kernel, dim3(stream_grid((size_t)(n_pairs * 2 * lpx), 256)), dim3(256), 0, st,
                       (const float*)I, T, (long long)n_pairs * 2, lh, lw, pk);
    hipLaunchKernelGGL(fb_polyexp_h_kernel, dim3(stream_grid((size_t)(n_pairs * 2 * lpx), 256)), dim3(256), 0, st,
                       (const float*)T, R, (long long)n_pairs * 2, lh, lw, pk);
    }
    const bool tile_path = lw <= 64 && lh <= 64 && n_pairs <= 0x7fffffffLL;   // any window: it is a precomputed matrix
    stage_mark(coarse ? "farneback.coarse.update_matrices" : "farneback.level0.update_matrices", st);
    hipLaunchKernelGGL(fb_update_matrices_kernel, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                       (const float*)R, (const float*)flow, M, (long long)n_pairs, lh, lw, tile_path ? 1 : 0);
    for (int it = 0; it < p->iterations; ++it) {
      const int update = it < p->iterations - 1 ? 1 : 0;
      if (tile_path) {
        // window blur + solve on the matrix cores (UpdateMatrices stays a separate high-occupancy launch: its
        // flow-dependent R1 gathers need many waves in flight)
        float* Gv = (float*)(ws + L.off_G);
        float* Gh = lh == lw ? Gv : Gv + 64 * 64;   // square levels: the vertical and horizontal window matrices coincide
        stage_mark(coarse ? "farneback.coarse.window_blur_solve" : "farneback.level0.window_blur_solve", st);
        if (it == 0) {
          hipLaunchKernelGGL(fb_window_matrix_kernel, dim3(16), dim3(256), 0, st, Gv, lh, win);
          if (Gh != Gv) hipLaunchKernelGGL(fb_window_matrix_kernel, dim3(16), dim3(256), 0, st, Gh, lw, win);
        }
        if (lh <= 32 && lw <= 32) {
          const unsigned grid = (unsigned)std::min<long long>((n_pairs + 3) / 4, 2048);
          hipLaunchKernelGGL(fb_tile_mfma_kernel<true>, dim3(grid), dim3(256), 0, st, (const float*)M, (const float*)Gv,
                             (const float*)Gh, flow, lh, lw, (long long)n_pairs);
        } else {
          const unsigned grid = (unsigned)std::min<long long>(n_pairs, 2048);
          hipLaunchKernelGGL(fb_tile_mfma_kernel<false>, dim3(grid), dim3(256), 0, st, (const float*)M, (const float*)Gv,
                             (const float*)Gh, flow, lh, lw, (long long)n_pairs);
        }
        if (update) {
          stage_mark(coarse ? "farneback.coarse.update_matrices" : "farneback.level0.update_matrices", st);
          hipLaunchKernelGGL(fb_update_matrices_kernel, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                             (const float*)R, (const float*)flow, M, (long long)n_pairs, lh, lw, 1);
        }
        continue;
      }
      stage_mark(coarse ? "farneback.coarse.window_blur_solve" : "farneback.level0.window_blur_solve", st);
      hipLaunchKernelGGL(fb_blur_v_kernel, dim3(stream_grid((size_t)(n_pairs * lpx * 5), 256)), dim3(256), 0, st,
                         (const float*)M, V, (long long)n_pairs, lh, lw, win);
      hipLaunchKernelGGL(fb_blur_h_solve_kernel, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                         (const float*)V, flow, (long long)n_pairs, lh, lw, win);
      if (update) {
        stage_mark(coarse ? "farneback.coarse.update_matrices" : "farneback.level0.update_matrices", st);
        hipLaunchKernelGGL(fb_update_matrices_kernel, dim3(stream_grid((size_t)(n_pairs * lpx), 256)), dim3(256), 0, st,
                           (const float*)R, (const float*)flow, M, (long long)n_pairs, lh, lw, 0);
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
