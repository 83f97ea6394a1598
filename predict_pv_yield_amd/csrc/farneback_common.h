// Shared declarations of the Farneback kernels (farneback.hip): tap tables, image indexing, border helpers.
#pragma once
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


}  // namespace pv
