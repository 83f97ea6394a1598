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
//
// Layout of the sources (round 5): farneback_common.h (tap tables, image indexing), farneback_frame.h (levels larger than a
// tile), farneback_tile.h (levels up to 64 x 64: the advection pipeline's kernels), farneback_tables.h (host-side tables and
// the workspace layout); this file is the dispatcher behind the C ABI.  One translation unit: the parts are included here.
#include "farneback_common.h"
#include "farneback_frame.h"
#include "farneback_tile.h"
#include "farneback_tables.h"

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
  // the window and PolyExp matrices of every tile level (<= 64 x 64), all in one launch; slot_of[k] = the level's slot or -1
  int slot_of[64];
  {
    FbLevelTables lv;
    lv.n = 0;
    for (int k = levels; k >= 0; --k) {
      double scale = 1;
      for (int i = 0; i < k; ++i) scale *= p->pyr_scale;
      const int lw = host_cv_round(w * scale), lh = host_cv_round(h * scale);
      slot_of[k] = -1;
      if (k < 64 && lw <= 64 && lh <= 64 && lw >= 2 && lh >= 2 && lv.n < FB_MAX_TABLE_LEVELS) {
        slot_of[k] = lv.n;
        lv.lh[lv.n] = lh, lv.lw[lv.n] = lw, lv.mosaic[lv.n] = (lh <= 32 && lw <= 32) ? 1 : 0;
        ++lv.n;
      }
    }
    if (lv.n > 0)
      hipLaunchKernelGGL(fb_level_tables_kernel, dim3(128, (unsigned)lv.n), dim3(256), 0, st, (float*)(ws + L.off_G), lv, win, pk);
  }
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
    // width that is no multiple of 4, larger source images, PV_FARNEBACK_TWO_LAUNCH_ITERATION=1: the tests' cross-check) takes
    // the frame family's kernels below: UpdateMatrices writes M, the window passes + solve read it back.
    const bool small_level = lh <= 32 && lw <= 32;
    const bool polyexp_tile = h <= 64 && w <= 64 && smooth_sz <= 63;
    const int slot = k < 64 ? slot_of[k] : -1;
    float* level_tables = (float*)(ws + L.off_G) + (size_t)(slot < 0 ? 0 : slot) * 8 * 64 * 64;
    const bool fused_iter = tile_path && slot >= 0 && fuse_init && polyexp_tile && (lw & 3) == 0 && ((uintptr_t)flow & 15) == 0 &&
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
      float* P6 = level_tables + 2 * 64 * 64;
      FbPolyScales sc;
      host_polyexp_scales(pk, lh, lw, &sc);
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
      float* Gv = level_tables;
      float* Gh = Gv + 64 * 64;
      stage_mark(coarse ? "farneback.coarse.iterations_fused" : "farneback.level0.iterations_fused", st);
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
                           (long long)n_pairs, lh, lw, 0, (long long)pairs_per_group, chain_f, up);
      else if (prev_flow)
        hipLaunchKernelGGL(fb_update_matrices_kernel<1>, um_grid, dim3(256), 0, st, (const float*)R, (const float*)prev_flow, M,
                           (long long)n_pairs, lh, lw, 0, (long long)pairs_per_group, chain_f, up);
      else
        hipLaunchKernelGGL(fb_update_matrices_kernel<2>, um_grid, dim3(256), 0, st, (const float*)R, (const float*)nullptr, M,
                           (long long)n_pairs, lh, lw, 0, (long long)pairs_per_group, chain_f, up);
    }
    for (int it = 0; it < p->iterations; ++it) {
      const int update = it < p->iterations - 1 ? 1 : 0;
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

