// Farneback, host side: tap tables (same arithmetic as oracle/pv_oracle.c), pyramid level rule, workspace layout, parameter
// checks.  Included once, by farneback.hip.
#pragma once
#include "farneback_common.h"

namespace pv {

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
  L.off_G = o; o = align(o + (size_t)FB_MAX_TABLE_LEVELS * 8 * 64 * 64 * sizeof(float));   // per tile level: window matrices (2) and PolyExp matrices (6)
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
