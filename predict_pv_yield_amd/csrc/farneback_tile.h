// Farneback, TILE family: levels of at most 64 x 64 pixels (the PV-site tiles of the advection pipeline) -- prep + PolyExp in
// LDS (vector-ALU form and the f16 matrix-core form), the window matrices, the two-launch iteration (UpdateMatrices + blur on
// the matrix cores) and the one-launch-per-level kernel with uniform waves (fb_level_u_kernel).
// Included once, by farneback.hip, after farneback_frame.h (FbUpsample, fb_update_matrices_kernel).
#pragma once
#include "farneback_common.h"

namespace pv {

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

// ---- window blur as matrix products (fb_level_u_kernel below) -------------------------------------------------------------------
// For images up to 64 x 64 the separable, border-replicated window blur is two small matrix products per channel,
//   U = X Gh^T (along x),  Out = Gv U (along y),   G[y][y'] = sum of the taps k with clamp(y + k) == y'
// (a banded 64 x 64 matrix that already contains the border replication, built for every tile level of a call by fb_level_tables_kernel).
// (Rounds 2-5 also kept a two-launch iteration for the same levels -- UpdateMatrices writing M, fb_tile_mfma_kernel /
// fb_tile_mfma_q_kernel blurring it on the bf16 matrix cores with three-term operands -- for widths that are no multiple of
// four and as a cross-check; removed in round 6: those levels take the frame family's kernels (farneback_frame.h), which are
// also the level kernel's cross-check now.)
typedef float fb_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void fb_window_matrix_fill(float* __restrict__ G, int n, const FbTaps& kt, int mosaic, int first, int stride) {
  // G[64][64]; rows / columns >= n stay zero.  mosaic (n <= 32): the n x n matrix twice on the diagonal, at 0 and at 32 --
  // the blur of a 64 x 64 image made of 2 x 2 independent tiles (fb_level_u_kernel<.., MOSAIC = true>)
  for (int i = first; i < 64 * 64; i += stride) {
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
  // MOSAIC (round 6): the window matrices are block-diagonal (two 32 x 32 blocks), so of the four 16-wide contraction steps of a
  // product only the two inside this wave's own block multiply anything but zeros, and the second product needs the first one's
  // block mbo alone: 12 matrix-instruction triples per channel instead of 36 -- the same sums, the zero terms left out.
  constexpr int KSE = MOSAIC ? 2 : KS;
  FbSplit2 gv[KSE], gh[KSE];
#pragma unroll
  for (int ks = 0; ks < KSE; ++ks) {
    const int ksv = MOSAIC ? 2 * strip + ks : ks, ksh = MOSAIC ? 2 * mbo + ks : ks;      // the steps' numbers in the full matrices
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gv[(32 * strip + col) * 64 + 16 * ksv + 8 * half + i];
    gv[ks] = fb_split2(t, FB_G_SCALE);
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = Gh[(32 * mbo + col) * 64 + 32 * (ksh >> 1) + fb_acc_row(8 * (ksh & 1) + i, half)];
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
    constexpr int NMB = MOSAIC ? 1 : 2;      // row blocks of the first product this wave needs (MOSAIC: its own tile's)
    fb_v16f u[NMB];
#pragma unroll
    for (int mbi = 0; mbi < NMB; ++mbi) {
      const int mb = MOSAIC ? mbo : mbi;
#pragma unroll
      for (int r = 0; r < 16; ++r) u[mbi][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSE; ++ks) {
        FbSplit2 a;
        const uint16_t* xa = Xc + (32 * mb + col) * XS + 16 * (MOSAIC ? 2 * strip + ks : ks) + 8 * half;
        a.h = *reinterpret_cast<const fb_f16x8*>(xa);
        a.l = *reinterpret_cast<const fb_f16x8*>(xa + PLANE);
        u[mbi] = fb_mfma2(a, gv[ks], u[mbi]);
      }
    }
    fb_v16f res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSE; ++ks) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = u[MOSAIC ? 0 : ks >> 1][8 * (ks & 1) + i];
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
// border replication (fb_polyexp_matrix_fill): vertically t_j = V_j I for V_g, V_xg, V_xxg, horizontally
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

__device__ __forceinline__ void fb_polyexp_matrix_fill(float* __restrict__ P6, int lh, int lw, const FbPoly& pk, int mosaic, int first, int stride) {
  // P6[3 v + j][64][64]: v = 0 vertical (size lh), 1 horizontal (size lw); j = 0: g, 1: x g (odd), 2: x^2 g.  Row y, column y':
  // the weight of input y' in output y, border replicated.  mosaic (sizes <= 32): the matrix twice on the diagonal (0 and 32)
  for (int i = first; i < 6 * 64 * 64; i += stride) {
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

// Every matrix of every tile level of a call in ONE launch (round 6: two launches of ~5 us per level -- the window matrices and the
// PolyExp matrices -- were 26 us of the 0.94 ms pipeline with their kernel boundaries).  blockIdx.y = level slot; per slot 8
// matrices of 64 x 64 floats: Gv, Gh (window), P6 (PolyExp: vertical g / xg / xxg, horizontal g / xg / xxg).  blockIdx.x 0..15: Gv,
// 16..31: Gh, 32..127: P6.
constexpr int FB_MAX_TABLE_LEVELS = 8;
struct FbLevelTables { int n; int lh[FB_MAX_TABLE_LEVELS], lw[FB_MAX_TABLE_LEVELS], mosaic[FB_MAX_TABLE_LEVELS]; };
__global__ __launch_bounds__(256) void fb_level_tables_kernel(float* __restrict__ G, FbLevelTables lv, FbTaps win, FbPoly pk) {
  const int slot = blockIdx.y;
  if (slot >= lv.n) return;
  float* base = G + (size_t)slot * 8 * 64 * 64;
  const int lh = lv.lh[slot], lw = lv.lw[slot], mosaic = lv.mosaic[slot];
  const int bx = blockIdx.x;
  if (bx < 16) fb_window_matrix_fill(base, lh, win, mosaic, bx * 256 + threadIdx.x, 16 * 256);
  else if (bx < 32) fb_window_matrix_fill(base + 64 * 64, lw, win, mosaic, (bx - 16) * 256 + threadIdx.x, 16 * 256);
  else fb_polyexp_matrix_fill(base + 2 * 64 * 64, lh, lw, pk, mosaic, (bx - 32) * 256 + threadIdx.x, 96 * 256);
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
  // MOSAIC (round 6): the matrices are block-diagonal (two 32 x 32 blocks): of a product's four 16-wide contraction steps only
  // the two inside the wave's own block multiply anything but zeros, and a second product needs the first one's block mbo alone
  // (as in fb_level_u_kernel): KSE steps per operand, numbered ksv / ksh in the full matrices.  One image per tile: the
  // matrices are banded, poly_n <= 7 on either side, so the 32 output rows of block s reach inputs 32 s - 7 .. 32 s + 38: three
  // of the four steps, numbers s .. s + 2 (round 5 skipped the fourth by a run-time vote on the operand registers and gained
  // nothing; here the loops are simply shorter)
  constexpr int KSE = MOSAIC ? 2 : 3, KSB = MOSAIC ? 2 : KS;
  FbSplit2 gvA[KSE], opX[KSE], ghA[KSE], ghB[KSE];
  {
    const float* va = P6 + (grp ? 1 : 0) * 4096;
    const float sva = grp ? sc.sV[1] : sc.sV[0];
#pragma unroll
    for (int ks = 0; ks < KSE; ++ks) {
      const int ksv = MOSAIC ? 2 * strip + ks : strip + ks, ksh = MOSAIC ? 2 * mbo + ks : mbo + ks;
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = va[(32 * strip + col) * 64 + 16 * ksv + 8 * half + i];
      gvA[ks] = fb_split2(t, sva);
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = P6[3 * 4096 + (32 * mbo + col) * 64 + 32 * (ksh >> 1) + fb_acc_row(8 * (ksh & 1) + i, half)];
      ghA[ks] = fb_split2(t, sc.sH[0]);
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = P6[4 * 4096 + (32 * mbo + col) * 64 + 32 * (ksh >> 1) + fb_acc_row(8 * (ksh & 1) + i, half)];
      ghB[ks] = fb_split2(t, sc.sH[1]);
      // (one address expression for both groups: the group picks the matrix, the row and the column order by arithmetic)
      const int xrow = grp ? 32 * strip + col : 32 * mbo + col;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int xcol = grp ? 16 * ksv + 8 * half + i : 32 * (ksh >> 1) + fb_acc_row(8 * (ksh & 1) + i, half);
        t[i] = P6[(grp ? 2 : 5) * 4096 + xrow * 64 + xcol];
      }
      opX[ks] = fb_split2(t, grp ? sc.sV[2] : sc.sH[2]);
    }
  }
  auto first = [&](const FbSplit2 (&gv)[KSE], float f, FbSplit2 (&b)[KSB]) __attribute__((always_inline)) {
#pragma unroll
    for (int mbi = 0; mbi < (MOSAIC ? 1 : 2); ++mbi) {      // row block mb of U becomes k-steps 2 mb, 2 mb + 1 of the second product: one block alive
      const int mb = MOSAIC ? mbo : mbi;
      fb_v16f u;
#pragma unroll
      for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSE; ++ks) {
        FbSplit2 a;
        const uint16_t* xa = &Xs[0][0] + (32 * mb + col) * XS + 16 * (MOSAIC ? 2 * strip + ks : strip + ks) + 8 * half;
        a.h = *reinterpret_cast<const fb_f16x8*>(xa);
        a.l = *reinterpret_cast<const fb_f16x8*>(xa + PLANE);
        u = fb_mfma2(a, gv[ks], u);
      }
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = u[8 * k2 + i];
        b[(MOSAIC ? 0 : 2 * mbi) + k2] = fb_split2(t, f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto second = [&](const FbSplit2 (&gh)[KSE], const FbSplit2 (&b)[KSB]) __attribute__((always_inline)) -> fb_v16f {
    fb_v16f res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.f;
#pragma unroll
    // U as the A operand, the window matrix as B (the same register contents: the two operand layouts mirror each other): the
    // result is the block itself rather than its transpose -- lane = COLUMN x, registers = rows -- so that the 32 lanes of a
    // half wave store 256 contiguous bytes of a row
    for (int ks = 0; ks < KSE; ++ks) {
      if constexpr (MOSAIC) {
        res = fb_mfma2(b[ks], gh[ks], res);
      } else {      // steps mbo .. mbo + 2 of the first product's four (a wave-uniform choice between two registers' worth)
        FbSplit2 bs;
        bs.h = mbo ? b[ks + 1].h : b[ks].h;
        bs.l = mbo ? b[ks + 1].l : b[ks].l;
        res = fb_mfma2(bs, gh[ks], res);
      }
    }
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
    FbSplit2 b[KSB];
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


}  // namespace pv
