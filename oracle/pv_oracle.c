/*
 * pv_oracle.c — CPU restatement (plain C, scalar, single thread) of the optical-flow advection
 * half of the hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker; never by the product path.
 *
 * PARITY UNPINNED for Farnebäck and remap: the arithmetic lives in OpenCV (cv2), which the
 * reference neither pins nor declares (absent from requirements.txt / environment.yml; notebooks
 * link the 4.5.0 docs) and which is not installed here; the reference has no tests or golden
 * vectors for this path.  What follows is a C transcription (see the licence note below) of
 * modules/video/src/optflowgf.cpp (FarnebackPolyExp / FarnebackUpdateMatrices /
 * FarnebackUpdateFlow_GaussianBlur / FarnebackOpticalFlowImpl::calc) and
 * modules/imgproc/src/imgwarp.cpp (remapBilinear, 1/32-px fixed-point maps) of OpenCV 4.5,
 * anchored on the reference call sites:
 *   cv.calcOpticalFlowFarneback  notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:133-135
 *   cv.remap                     notebooks/13_...ipynb:275-281, notebooks/optical_flow_1.ipynb:430
 *   convert_10bpp_to_uint8       notebooks/13_...ipynb:112-119, notebooks/optical_flow_1.ipynb:129-134
 *   weighted_average             notebooks/optical_flow_1.ipynb:293-294
 *   normalisation                notebooks/13_...ipynb:345-346,463-464
 * and pinned by the analytic known-answer tests in tests/test_oracle_flow.py (SURVEY.md §8c K1-K5,
 * R1-R5).  Build with -ffp-contract=off so every product is rounded before it is added.
 *
 * Attribution and licence.  poly_exp / update_matrices / update_flow_gaussian_blur / prepare_gaussian / the pyramid driver
 * and remap_bilinear below are TRANSCRIBED into C from OpenCV 4.5
 *   modules/video/src/optflowgf.cpp   (FarnebackPolyExp, FarnebackUpdateMatrices, FarnebackUpdateFlow_GaussianBlur,
 *                                      FarnebackPrepareGaussian, FarnebackOpticalFlowImpl::calc)
 *   modules/imgproc/src/imgwarp.cpp   (remapBilinear, the 1/32-px fixed-point map conversion)
 * statement by statement (same operation order, temporaries, border tables and fixed-point quantisation), so that this
 * file can serve as a bit-level checker of those functions.  OpenCV is
 *   Copyright (C) 2000-2020, Intel Corporation, Willow Garage Inc., NVIDIA Corporation, Advanced Micro Devices, Inc.,
 *   OpenCV Foundation, Itseez Inc., Xperience AI and the other OpenCV contributors, all rights reserved,
 * and is licensed under the Apache License, Version 2.0 (http://www.apache.org/licenses/LICENSE-2.0); this derived file is
 * distributed under the same licence: you may not use it except in compliance with the Licence; software distributed
 * under the Licence is distributed on an "AS IS" BASIS, WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or
 * implied.  Changes made: C++ templates / cv::Mat / parallel_for_ replaced by plain C arrays and loops, f32 only, the
 * OPTFLOW_FARNEBACK_GAUSSIAN branch only, no SIMD paths.  The method is G. Farnebäck, "Two-Frame Motion Estimation Based
 * on Polynomial Expansion", SCIA 2003.  OpenCV is not part of /root/reference (the reference calls it through cv2), and
 * nothing in the product path includes, links or imports this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <limits.h>

#define PVO_BORDER_CONSTANT 0
#define PVO_BORDER_REPLICATE 1

/* cvRound: round half to even (SSE cvtsd2si); NaN / out of range -> INT_MIN */
static int cv_round_d(double v) {
  if (!(fabs(v) < 2147483648.0)) return INT_MIN;
  return (int)nearbyint(v);
}
static int cv_round_f(float v) {
  if (!(fabsf(v) < 2147483648.0f)) return INT_MIN;
  return (int)nearbyintf(v);
}
static int cv_floor_f(float v) { return (int)floorf(v); }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------ */
/* u8 conversion — notebooks/13_...ipynb:112-119 (mode 0) / optical_flow_1.ipynb:129-134 (mode 1) */
/* ------------------------------------------------------------------------------------------ */
static int u8_from_f32(float x, int mode, int* bad) {
  float v;
  if (mode == 0) {
    v = nearbyintf(x / 4.0f);
  } else {
    v = x / 1023.0f;
    v = v * 255.0f;
    v = truncf(v);
  }
  if (!(v >= 0.0f && v <= 255.0f)) {
    *bad = 1;
    v = v > 255.0f ? 255.0f : 0.0f;
  }
  return (int)v;
}

void pvo_u8_from_10bit_f32(const float* src, uint8_t* dst, size_t n, int mode, int* range_flag) {
  int bad = 0;
  for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)u8_from_f32(src[i], mode, &bad);
  if (range_flag) *range_flag = bad;
}

void pvo_u8_from_10bit_i16(const int16_t* src, uint8_t* dst, size_t n, int mode, int* range_flag) {
  int bad = 0;
  for (size_t i = 0; i < n; ++i) {
    if (mode == 0) {
      /* int16 / 4.0 is float64 in NumPy; np.round is half-to-even */
      double v = nearbyint((double)src[i] / 4.0);
      if (!(v >= 0.0 && v <= 255.0)) {
        bad = 1;
        v = v > 255.0 ? 255.0 : 0.0;
      }
      dst[i] = (uint8_t)(int)v;
    } else {
      dst[i] = (uint8_t)u8_from_f32((float)src[i], mode, &bad);
    }
  }
  if (range_flag) *range_flag = bad;
}

/* ------------------------------------------------------------------------------------------ */
/* weighted mean — np.average(flows, axis=0, weights=w).astype(f32), optical_flow_1.ipynb:293-294 */
/* ------------------------------------------------------------------------------------------ */
void pvo_weighted_mean_f32(const float* flows, const double* weights, float* out, int64_t n_groups,
                           int n_per_group, int64_t elems) {
  double wsum = 0.0;
  for (int k = 0; k < n_per_group; ++k) wsum += weights ? weights[k] : (double)(k + 1);
  for (int64_t g = 0; g < n_groups; ++g) {
    for (int64_t e = 0; e < elems; ++e) {
      double acc = 0.0;
      for (int k = 0; k < n_per_group; ++k) {
        double w = weights ? weights[k] : (double)(k + 1);
        acc += (double)flows[(g * n_per_group + k) * elems + e] * w;
      }
      out[g * elems + e] = (float)(acc / wsum);
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* normalisation — notebooks/13_...ipynb:463-464; constants netcdf_dataset.py:19-32              */
/* ------------------------------------------------------------------------------------------ */
void pvo_normalise_f32(const float* src, float* dst, size_t n, int64_t inner, int n_channels,
                       const float* mean, const float* std_) {
  for (size_t i = 0; i < n; ++i) {
    int c = (int)((i / (size_t)inner) % (size_t)n_channels);
    float v = src[i] - mean[c];
    dst[i] = v / std_[c];
  }
}
void pvo_normalise_i16(const int16_t* src, float* dst, size_t n, int64_t inner, int n_channels,
                       const float* mean, const float* std_) {
  for (size_t i = 0; i < n; ++i) {
    int c = (int)((i / (size_t)inner) % (size_t)n_channels);
    float v = (float)src[i] - mean[c];
    dst[i] = v / std_[c];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* cv.remap(INTER_LINEAR) — SURVEY.md Appendix A.2; reference map = meshgrid - flow*k             */
/* (notebooks/13_...ipynb:270-273: remap = -flow.copy(); remap[...,0] += arange(width) ...)        */
/* idx_out (may be NULL): int32 [h*w*4] = (ix, iy, fx, fy) per pixel, the bit-exact index contract  */
/* ------------------------------------------------------------------------------------------ */
static int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

static void remap_coords(float flow_x, float flow_y, float k, int x, int y, int* ix, int* iy, int* fx,
                         int* fy) {
  float ax = flow_x * k;
  float ay = flow_y * k;
  float mx = -ax + (float)x;
  float my = -ay + (float)y;
  int sx = cv_round_f(mx * 32.0f);
  int sy = cv_round_f(my * 32.0f);
  *fx = sx & 31;
  *fy = sy & 31;
  *ix = sat_short(sx >> 5);
  *iy = sat_short(sy >> 5);
}

void pvo_remap_bilinear_f32(const float* src, const float* flow, float k, float* dst, int h, int w,
                            int border_mode, float border_value, int32_t* idx_out) {
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      int ix, iy, fx, fy;
      const float* f = flow + ((size_t)y * w + x) * 2;
      remap_coords(f[0], f[1], k, x, y, &ix, &iy, &fx, &fy);
      if (idx_out) {
        int32_t* o = idx_out + ((size_t)y * w + x) * 4;
        o[0] = ix; o[1] = iy; o[2] = fx; o[3] = fy;
      }
      float p00, p01, p10, p11;
      if ((unsigned)ix < (unsigned)(w - 1) && (unsigned)iy < (unsigned)(h - 1)) {
        const float* p = src + (size_t)iy * w + ix;
        p00 = p[0]; p01 = p[1]; p10 = p[w]; p11 = p[w + 1];
      } else if (border_mode == PVO_BORDER_REPLICATE) {
        int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
        int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
        p00 = src[(size_t)y0 * w + x0]; p01 = src[(size_t)y0 * w + x1];
        p10 = src[(size_t)y1 * w + x0]; p11 = src[(size_t)y1 * w + x1];
      } else {
        if (ix >= w || ix + 1 < 0 || iy >= h || iy + 1 < 0) {
          dst[(size_t)y * w + x] = border_value;
          continue;
        }
        int x0in = (unsigned)ix < (unsigned)w, x1in = (unsigned)(ix + 1) < (unsigned)w;
        int y0in = (unsigned)iy < (unsigned)h, y1in = (unsigned)(iy + 1) < (unsigned)h;
        p00 = (x0in && y0in) ? src[(size_t)iy * w + ix] : border_value;
        p01 = (x1in && y0in) ? src[(size_t)iy * w + ix + 1] : border_value;
        p10 = (x0in && y1in) ? src[(size_t)(iy + 1) * w + ix] : border_value;
        p11 = (x1in && y1in) ? src[(size_t)(iy + 1) * w + ix + 1] : border_value;
      }
      float ax = (float)fx * (1.0f / 32.0f), ay = (float)fy * (1.0f / 32.0f);
      float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax;
      float w10 = ay * (1.0f - ax), w11 = ay * ax;
      float r = p00 * w00;
      r = r + p01 * w01;
      r = r + p10 * w10;
      r = r + p11 * w11;
      dst[(size_t)y * w + x] = r;
    }
  }
}

void pvo_remap_bilinear_u8(const uint8_t* src, const float* flow, float k, uint8_t* dst, int h, int w,
                           int border_mode, uint8_t border_value, int32_t* idx_out) {
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      int ix, iy, fx, fy;
      const float* f = flow + ((size_t)y * w + x) * 2;
      remap_coords(f[0], f[1], k, x, y, &ix, &iy, &fx, &fy);
      if (idx_out) {
        int32_t* o = idx_out + ((size_t)y * w + x) * 4;
        o[0] = ix; o[1] = iy; o[2] = fx; o[3] = fy;
      }
      int p00, p01, p10, p11;
      if ((unsigned)ix < (unsigned)(w - 1) && (unsigned)iy < (unsigned)(h - 1)) {
        const uint8_t* p = src + (size_t)iy * w + ix;
        p00 = p[0]; p01 = p[1]; p10 = p[w]; p11 = p[w + 1];
      } else if (border_mode == PVO_BORDER_REPLICATE) {
        int x0 = clampi(ix, 0, w - 1), x1 = clampi(ix + 1, 0, w - 1);
        int y0 = clampi(iy, 0, h - 1), y1 = clampi(iy + 1, 0, h - 1);
        p00 = src[(size_t)y0 * w + x0]; p01 = src[(size_t)y0 * w + x1];
        p10 = src[(size_t)y1 * w + x0]; p11 = src[(size_t)y1 * w + x1];
      } else {
        if (ix >= w || ix + 1 < 0 || iy >= h || iy + 1 < 0) {
          dst[(size_t)y * w + x] = border_value;
          continue;
        }
        int x0in = (unsigned)ix < (unsigned)w, x1in = (unsigned)(ix + 1) < (unsigned)w;
        int y0in = (unsigned)iy < (unsigned)h, y1in = (unsigned)(iy + 1) < (unsigned)h;
        p00 = (x0in && y0in) ? src[(size_t)iy * w + ix] : border_value;
        p01 = (x1in && y0in) ? src[(size_t)iy * w + ix + 1] : border_value;
        p10 = (x0in && y1in) ? src[(size_t)(iy + 1) * w + ix] : border_value;
        p11 = (x1in && y1in) ? src[(size_t)(iy + 1) * w + ix + 1] : border_value;
      }
      /* BilinearTab_i: saturate_cast<short>(w * 32768); only the (0,0) entry saturates */
      int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
      int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
      if (w00 > 32767) w00 = 32767;
      int acc = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
      int r = (acc + (1 << 14)) >> 15;
      dst[(size_t)y * w + x] = (uint8_t)clampi(r, 0, 255);
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Farnebäck — SURVEY.md Appendix A.1                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct pvo_farneback_params {
  double pyr_scale;
  int32_t levels;
  int32_t winsize;
  int32_t iterations;
  int32_t poly_n;
  double poly_sigma;
  int32_t flags;
} pvo_farneback_params;

static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * n - 2 - i;
  }
  return i;
}

/* cv::getGaussianKernel(n, sigma, CV_32F) */
static void gaussian_kernel_f32(int n, double sigma, float* k) {
  static const float tab1[] = {1.f};
  static const float tab3[] = {0.25f, 0.5f, 0.25f};
  static const float tab5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
  static const float tab7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
  const float* fixed = NULL;
  if ((n & 1) && n <= 7 && sigma <= 0) fixed = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
    k[i] = (float)t;
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

/* cv::GaussianBlur(f32, ksize x ksize, sigma) with BORDER_REFLECT_101: separable, rows then columns */
static void gaussian_blur_f32(const float* src, float* dst, int h, int w, int ksize, double sigma) {
  float* k = (float*)malloc(sizeof(float) * ksize);
  float* tmp = (float*)malloc(sizeof(float) * (size_t)h * w);
  gaussian_kernel_f32(ksize, sigma, k);
  int r = ksize / 2;
  for (int y = 0; y < h; ++y) {
    const float* s = src + (size_t)y * w;
    float* d = tmp + (size_t)y * w;
    for (int x = 0; x < w; ++x) {
      float acc;
      if (ksize <= 5) {
        /* SymmRowSmallFilter: centre * k0 + (left + right) * k1 (+ ... k2) */
        acc = s[x] * k[r];
        for (int i = 1; i <= r; ++i)
          acc = acc + (s[reflect101(x - i, w)] + s[reflect101(x + i, w)]) * k[r + i];
      } else {
        /* generic RowFilter: sum_{i} S[x + i - r] * k[i], left to right */
        acc = s[reflect101(x - r, w)] * k[0];
        for (int i = 1; i < ksize; ++i) acc = acc + s[reflect101(x + i - r, w)] * k[i];
      }
      d[x] = acc;
    }
  }
  for (int y = 0; y < h; ++y) {
    float* d = dst + (size_t)y * w;
    for (int x = 0; x < w; ++x) {
      /* SymmColumn(Small)Filter */
      float acc = tmp[(size_t)y * w + x] * k[r];
      for (int i = 1; i <= r; ++i)
        acc = acc + (tmp[(size_t)reflect101(y - i, h) * w + x] + tmp[(size_t)reflect101(y + i, h) * w + x]) * k[r + i];
      d[x] = acc;
    }
  }
  free(k);
  free(tmp);
}

/* cv::resize(..., INTER_LINEAR) for cn-channel f32 images.  Exact 2x2 decimation takes OpenCV's
 * INTER_AREA fast path ((a+b)+(c+d))*0.25; everything else is the generic bilinear resizer. */
static void resize_linear_f32(const float* src, int sh, int sw, float* dst, int dh, int dw, int cn) {
  double inv_fx = (double)sw / dw, inv_fy = (double)sh / dh;
  int iscale_x = (int)inv_fx, iscale_y = (int)inv_fy; /* saturate_cast<int> of an exact integer */
  int is_area_fast = fabs(inv_fx - iscale_x) < DBL_EPSILON && fabs(inv_fy - iscale_y) < DBL_EPSILON;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int c = 0; c < cn; ++c) {
          const float* s0 = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
          const float* s1 = s0 + (size_t)sw * cn;
          float a = s0[0] + s0[cn];
          float b = s1[0] + s1[cn];
          dst[((size_t)y * dw + x) * cn + c] = (a + b) * 0.25f;
        }
    return;
  }
  int* xofs = (int*)malloc(sizeof(int) * dw);
  float* alpha = (float*)malloc(sizeof(float) * dw * 2);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * inv_fx - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    alpha[dx * 2] = 1.f - fx;
    alpha[dx * 2 + 1] = fx;
  }
  float* row0 = (float*)malloc(sizeof(float) * (size_t)dw * cn);
  float* row1 = (float*)malloc(sizeof(float) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * inv_fy - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
    int sy1 = clampi(sy + 1, 0, sh - 1);
    float b0 = 1.f - fy, b1 = fy;
    const float* s0 = src + (size_t)sy * sw * cn;
    const float* s1 = src + (size_t)sy1 * sw * cn;
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sx;
      float a0 = alpha[dx * 2], a1 = alpha[dx * 2 + 1];
      for (int c = 0; c < cn; ++c) {
        if (sx + 1 < sw) {
          row0[dx * cn + c] = s0[sx * cn + c] * a0 + s0[sx1 * cn + c] * a1;
          row1[dx * cn + c] = s1[sx * cn + c] * a0 + s1[sx1 * cn + c] * a1;
        } else { /* dx >= xmax: D = S[sx] * 1 */
          row0[dx * cn + c] = s0[sx * cn + c];
          row1[dx * cn + c] = s1[sx * cn + c];
        }
      }
    }
    for (int i = 0; i < dw * cn; ++i) dst[(size_t)dy * dw * cn + i] = row0[i] * b0 + row1[i] * b1;
  }
  free(xofs); free(alpha); free(row0); free(row1);
}

/* FarnebackPrepareGaussian */
static void prepare_gaussian(int n, double sigma, float* g, float* xg, float* xxg, double* ig11,
                             double* ig03, double* ig33, double* ig55) {
  if (sigma < FLT_EPSILON) sigma = n * 0.3;
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
  double G[6][6];
  memset(G, 0, sizeof(G));
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
  /* invG = G^-1 (Cholesky in OpenCV; Gauss-Jordan with partial pivoting here, double) */
  double A[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 12; ++j) A[i][j] = j < 6 ? G[i][j] : (j - 6 == i ? 1.0 : 0.0);
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 12; ++j) { double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
    double d = 1.0 / A[c][c];
    for (int j = 0; j < 12; ++j) A[c][j] *= d;
    for (int r = 0; r < 6; ++r)
      if (r != c) {
        double f = A[r][c];
        if (f != 0.0)
          for (int j = 0; j < 12; ++j) A[r][j] -= f * A[c][j];
      }
  }
  *ig11 = A[1][6 + 1];
  *ig03 = A[0][6 + 3];
  *ig33 = A[3][6 + 3];
  *ig55 = A[5][6 + 5];
}

/* exported for the K4 known-answer test (Gaussian tables and inverse-moment constants) */
void pvo_farneback_poly_tables(int n, double sigma, float* g_out, float* xg_out, float* xxg_out,
                               double* ig4) {
  float buf[3 * 64];
  float* g = buf + n;
  float* xg = g + 2 * n + 1;
  float* xxg = xg + 2 * n + 1;
  prepare_gaussian(n, sigma, g, xg, xxg, &ig4[0], &ig4[1], &ig4[2], &ig4[3]);
  for (int i = -n; i <= n; ++i) {
    g_out[i + n] = g[i];
    xg_out[i + n] = xg[i];
    xxg_out[i + n] = xxg[i];
  }
}

/* FarnebackPolyExp: src f32 [h,w] -> dst f32 [h,w,5] */
static void poly_exp(const float* src, float* dst, int height, int width, int n, double sigma) {
  float* kbuf = (float*)malloc(sizeof(float) * (n * 6 + 3));
  float* rowbuf = (float*)malloc(sizeof(float) * (size_t)(width + n * 2) * 3);
  float* g = kbuf + n;
  float* xg = g + n * 2 + 1;
  float* xxg = xg + n * 2 + 1;
  float* row = rowbuf + n * 3;
  double ig11, ig03, ig33, ig55;
  prepare_gaussian(n, sigma, g, xg, xxg, &ig11, &ig03, &ig33, &ig55);

  for (int y = 0; y < height; y++) {
    float g0 = g[0], g1, g2;
    const float* srow0 = src + (size_t)y * width;
    const float* srow1 = 0;
    float* drow = dst + (size_t)y * width * 5;

    /* vertical part of convolution */
    for (int x = 0; x < width; x++) {
      row[x * 3] = srow0[x] * g0;
      row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
    }
    for (int k = 1; k <= n; k++) {
      g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
      srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * width;
      srow1 = src + (size_t)(y + k < height - 1 ? y + k : height - 1) * width;
      for (int x = 0; x < width; x++) {
        float p = srow0[x] + srow1[x];
        float t0 = row[x * 3] + g0 * p;
        float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
        float t2 = row[x * 3 + 2] + g2 * p;
        row[x * 3] = t0;
        row[x * 3 + 1] = t1;
        row[x * 3 + 2] = t2;
      }
    }
    /* horizontal part of convolution: replicate the edge triples n times each side */
    for (int x = 0; x < n * 3; x++) {
      row[-1 - x] = row[2 - x];
      row[width * 3 + x] = row[width * 3 + x - 3];
    }
    for (int x = 0; x < width; x++) {
      g0 = g[0];
      double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0,
             b6 = 0;
      for (int k = 1; k <= n; k++) {
        double tg = row[(x + k) * 3] + row[(x - k) * 3];
        g0 = g[k];
        b1 += tg * g0;
        b4 += tg * xxg[k];
        b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
        b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
        b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
        b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
      }
      /* do not store r1 */
      drow[x * 5 + 1] = (float)(b2 * ig11);
      drow[x * 5] = (float)(b3 * ig11);
      drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
      drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
      drow[x * 5 + 4] = (float)(b6 * ig55);
    }
  }
  free(kbuf);
  free(rowbuf);
}

/* FarnebackUpdateMatrices over rows [y0, y1) */
static void update_matrices(const float* R0a, const float* R1, const float* flowa, float* Ma, int height,
                            int width, int y0, int y1) {
  enum { BORDER = 5 };
  static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
  size_t step1 = (size_t)width * 5;
  for (int y = y0; y < y1; y++) {
    const float* flow = flowa + (size_t)y * width * 2;
    const float* R0 = R0a + (size_t)y * width * 5;
    float* M = Ma + (size_t)y * width * 5;
    for (int x = 0; x < width; x++) {
      float dx = flow[x * 2], dy = flow[x * 2 + 1];
      float fx = x + dx, fy = y + dy;
      int x1 = cv_floor_f(fx), y1_ = cv_floor_f(fy);
      float r2, r3, r4, r5, r6;
      fx -= x1;
      fy -= y1_;
      if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1_ < (unsigned)(height - 1)) {
        const float* ptr = R1 + (size_t)y1_ * step1 + (size_t)x1 * 5;
        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
        r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
        r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
        r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
        r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
        r4 = (R0[x * 5 + 2] + r4) * 0.5f;
        r5 = (R0[x * 5 + 3] + r5) * 0.5f;
        r6 = (R0[x * 5 + 4] + r6) * 0.25f;
      } else {
        r2 = r3 = 0.f;
        r4 = R0[x * 5 + 2];
        r5 = R0[x * 5 + 3];
        r6 = R0[x * 5 + 4] * 0.5f;
      }
      r2 = (R0[x * 5] - r2) * 0.5f;
      r3 = (R0[x * 5 + 1] - r3) * 0.5f;
      r2 += r4 * dy + r6 * dx;
      r3 += r6 * dy + r5 * dx;
      if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
          (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
        float scale = (x < BORDER ? border[x] : 1.f) * (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                      (y < BORDER ? border[y] : 1.f) * (y >= height - BORDER ? border[height - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
      }
      M[x * 5] = r4 * r4 + r6 * r6;      /* G(1,1) */
      M[x * 5 + 1] = (r4 + r5) * r6;     /* G(1,2) */
      M[x * 5 + 2] = r5 * r5 + r6 * r6;  /* G(2,2) */
      M[x * 5 + 3] = r4 * r2 + r6 * r3;  /* h(1)   */
      M[x * 5 + 4] = r6 * r2 + r5 * r3;  /* h(2)   */
    }
  }
}

/* exported: Gaussian window taps k[0..m] of FarnebackUpdateFlow_GaussianBlur (K4 test) */
void pvo_farneback_window_taps(int winsize, float* kernel) {
  int m = winsize / 2;
  double sigma = m * 0.3, s = 1;
  kernel[0] = (float)s;
  for (int i = 1; i <= m; i++) {
    float t = (float)exp(-i * i / (2 * sigma * sigma));
    kernel[i] = t;
    s += t * 2;
  }
  s = 1. / s;
  for (int i = 0; i <= m; i++) kernel[i] = (float)(kernel[i] * s);
}

/* FarnebackUpdateFlow_GaussianBlur (scalar path), including the lagging row-stripe matrix update */
static void update_flow_gaussian_blur(const float* R0, const float* R1, float* flowa, float* matM,
                                      int height, int width, int block_size, int update_matrices_flag,
                                      float* blurred_out /* may be NULL: the window-blurred M, [h][w][5] */) {
  int m = block_size / 2;
  int y0 = 0, y1;
  int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
  float* vsum_buf = (float*)malloc(sizeof(float) * ((size_t)(width + m * 2 + 2) * 5 + 16));
  float* hsum = (float*)malloc(sizeof(float) * ((size_t)width * 5 + 16));
  float* kernel = (float*)malloc(sizeof(float) * ((m + 1) * 5 + 16));
  const float** srow = (const float**)malloc(sizeof(float*) * (m * 2 + 1));
  float* vsum = vsum_buf + (m + 1) * 5;
  pvo_farneback_window_taps(block_size, kernel);

  for (int y = 0; y < height; y++) {
    double g11, g12, g22, h1, h2;
    float* flow = flowa + (size_t)y * width * 2;
    /* vertical blur */
    for (int i = 0; i <= m; i++) {
      srow[m - i] = matM + (size_t)(y - i > 0 ? y - i : 0) * width * 5;
      srow[m + i] = matM + (size_t)(y + i < height - 1 ? y + i : height - 1) * width * 5;
    }
    for (int x = 0; x < width * 5; x++) {
      float s0 = srow[m][x] * kernel[0];
      for (int i = 1; i <= m; i++) s0 += (srow[m + i][x] + srow[m - i][x]) * kernel[i];
      vsum[x] = s0;
    }
    /* update borders */
    for (int x = 0; x < m * 5; x++) {
      vsum[-1 - x] = vsum[4 - x];
      vsum[width * 5 + x] = vsum[width * 5 + x - 5];
    }
    /* horizontal blur */
    for (int x = 0; x < width * 5; x++) {
      float sum = vsum[x] * kernel[0];
      for (int i = 1; i <= m; i++) sum += kernel[i] * (vsum[x - i * 5] + vsum[x + i * 5]);
      hsum[x] = sum;
    }
    if (blurred_out) memcpy(blurred_out + (size_t)y * width * 5, hsum, sizeof(float) * (size_t)width * 5);
    for (int x = 0; x < width; x++) {
      g11 = hsum[x * 5];
      g12 = hsum[x * 5 + 1];
      g22 = hsum[x * 5 + 2];
      h1 = hsum[x * 5 + 3];
      h2 = hsum[x * 5 + 4];
      double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
      flow[x * 2] = (float)((g11 * h2 - g12 * h1) * idet);
      flow[x * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
    }
    y1 = y == height - 1 ? height : y - block_size;
    if (update_matrices_flag && (y1 == height || y1 >= y0 + min_update_stripe)) {
      update_matrices(R0, R1, flowa, matM, height, width, y0, y1);
      y0 = y1;
    }
  }
  free(vsum_buf); free(hsum); free(kernel); free(srow);
}

/* number of coarse pyramid levels actually used (K3 test) */
int pvo_farneback_num_levels(int h, int w, double pyr_scale, int levels) {
  const int min_size = 32;
  int k;
  double scale;
  for (k = 0, scale = 1; k < levels; k++) {
    scale *= pyr_scale;
    if (w * scale < min_size || h * scale < min_size) break;
  }
  return k;
}

/* FarnebackOpticalFlowImpl::calc — prev/next u8 [h,w] -> flow f32 [h,w,2]; returns 0 or -1 */
int pvo_farneback_u8(const uint8_t* prev0, const uint8_t* next0, float* flow0, int rows, int cols,
                     const pvo_farneback_params* p) {
  if (!(p->flags & 256) || (p->poly_n != 5 && p->poly_n != 7) || p->winsize < 2) return -1;
  const uint8_t* img[2] = {prev0, next0};
  int levels = pvo_farneback_num_levels(rows, cols, p->pyr_scale, p->levels);
  float* prevFlow = NULL;
  int prev_w = 0, prev_h = 0;
  size_t npx0 = (size_t)rows * cols;
  float* fimg = (float*)malloc(sizeof(float) * npx0);
  float* fblur = (float*)malloc(sizeof(float) * npx0);

  for (int k = levels; k >= 0; k--) {
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= p->pyr_scale;
    double sigma = (1. / scale - 1) * 0.5;
    int smooth_sz = cv_round_d(sigma * 5) | 1;
    smooth_sz = smooth_sz > 3 ? smooth_sz : 3;
    int width = cv_round_d(cols * scale);
    int height = cv_round_d(rows * scale);
    size_t npx = (size_t)width * height;

    float* flow = k > 0 ? (float*)malloc(sizeof(float) * npx * 2) : flow0;
    if (!prevFlow) {
      memset(flow, 0, sizeof(float) * npx * 2);
    } else {
      resize_linear_f32(prevFlow, prev_h, prev_w, flow, height, width, 2);
      float mul = (float)(1. / p->pyr_scale);
      for (size_t i = 0; i < npx * 2; ++i) flow[i] *= mul; /* Mat *= double: f32 * (f32)scale per elem */
    }

    float* R[2];
    float* I = (float*)malloc(sizeof(float) * npx);
    for (int i = 0; i < 2; i++) {
      for (size_t j = 0; j < npx0; ++j) fimg[j] = (float)img[i][j];
      gaussian_blur_f32(fimg, fblur, rows, cols, smooth_sz, sigma);
      if (width == cols && height == rows) memcpy(I, fblur, sizeof(float) * npx0);
      else resize_linear_f32(fblur, rows, cols, I, height, width, 1);
      R[i] = (float*)malloc(sizeof(float) * npx * 5);
      poly_exp(I, R[i], height, width, p->poly_n, p->poly_sigma);
    }
    free(I);
    float* M = (float*)malloc(sizeof(float) * npx * 5);
    update_matrices(R[0], R[1], flow, M, height, width, 0, height);
    for (int i = 0; i < p->iterations; i++)
      update_flow_gaussian_blur(R[0], R[1], flow, M, height, width, p->winsize, i < p->iterations - 1, NULL);
    free(M); free(R[0]); free(R[1]);
    if (prevFlow) free(prevFlow);
    prevFlow = k > 0 ? flow : NULL;
    prev_w = width; prev_h = height;
  }
  free(fimg); free(fblur);
  return 0;
}

/* intermediate access for stage-by-stage HIP parity tests: poly expansion of one u8 image at one level */
int pvo_farneback_level_polyexp(const uint8_t* img, int rows, int cols, int level, double pyr_scale,
                                int poly_n, double poly_sigma, float* I_out, float* R_out) {
  double scale = 1;
  for (int i = 0; i < level; i++) scale *= pyr_scale;
  double sigma = (1. / scale - 1) * 0.5;
  int smooth_sz = cv_round_d(sigma * 5) | 1;
  smooth_sz = smooth_sz > 3 ? smooth_sz : 3;
  int width = cv_round_d(cols * scale), height = cv_round_d(rows * scale);
  size_t npx0 = (size_t)rows * cols;
  float* fimg = (float*)malloc(sizeof(float) * npx0);
  float* fblur = (float*)malloc(sizeof(float) * npx0);
  for (size_t j = 0; j < npx0; ++j) fimg[j] = (float)img[j];
  gaussian_blur_f32(fimg, fblur, rows, cols, smooth_sz, sigma);
  if (width == cols && height == rows) memcpy(I_out, fblur, sizeof(float) * npx0);
  else resize_linear_f32(fblur, rows, cols, I_out, height, width, 1);
  poly_exp(I_out, R_out, height, width, poly_n, poly_sigma);
  free(fimg); free(fblur);
  return 0;
}

/* ---- stage-level access for the analytic pins of tests/test_oracle_flow.py (each is the static routine above, unchanged) ---- */

/* FarnebackPolyExp of a float image: R[h][w][5] = (r_y, r_x, r_yy, r_xx, r_xy) */
void pvo_poly_exp_f32(const float* src, int rows, int cols, int poly_n, double poly_sigma, float* R_out) {
  poly_exp(src, R_out, rows, cols, poly_n, poly_sigma);
}

/* FarnebackUpdateMatrices over the whole image: M[h][w][5] = (G11, G12, G22, h1, h2) */
void pvo_update_matrices_f32(const float* R0, const float* R1, const float* flow, int rows, int cols, float* M_out) {
  update_matrices(R0, R1, flow, M_out, rows, cols, 0, rows);
}

/* one FarnebackUpdateFlow_GaussianBlur pass WITHOUT the matrix update: M (read only) -> window-blurred M and the flow of
 * the 2x2 solve */
void pvo_window_blur_solve_f32(const float* M, int rows, int cols, int winsize, float* blurred_out, float* flow_out) {
  float* Mc = (float*)malloc(sizeof(float) * (size_t)rows * cols * 5);
  memcpy(Mc, M, sizeof(float) * (size_t)rows * cols * 5);
  update_flow_gaussian_blur(NULL, NULL, flow_out, Mc, rows, cols, winsize, 0, blurred_out);
  free(Mc);
}
