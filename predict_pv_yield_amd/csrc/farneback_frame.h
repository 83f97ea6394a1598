// Farneback, FRAME family: levels larger than one 64 x 64 tile (the 704 x 548 images of the notebooks, 12_just_3d_conv.ipynb:611 /
// 13_...ipynb:415-441) -- global-memory prep / PolyExp kernels (also the bit-identity references of the tests), the one-launch
// prep + PolyExp tile kernel with halos, UpdateMatrices, the window blur as sliding register windows, the flow upsample.
// Included once, by farneback.hip (one translation unit: the host dispatcher launches these templates directly).
#pragma once
#include "farneback_common.h"

namespace pv {

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


}  // namespace pv
