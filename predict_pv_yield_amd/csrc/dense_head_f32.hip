// The small fully-connected head of the PV-yield models (fc2 -> fc3 -> fc4 of predict_pv_yield/models/conv3d/model.py:126,
// 151-156: at most 128 features a layer, a batch of at most 32 rows) as ONE launch each way.
//
// As three launches each way (linear_fwd_small_f32 / linear_bwd_small_f32, dense_f32.hip) the head cost 18 + 26 us of the
// 1.54 ms train step -- nine kernel boundaries for 1.4 MFLOP.  A first one-launch form (round 4) walked the layers with one
// thread per output and read every weight row through the vector cache once per batch row: slower than the launches.  Here a
// single 1024-thread workgroup keeps the activations in LDS and runs every product on the f32 matrix instruction
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation; M = the 32 batch rows), with a weight element read from
// memory exactly once, by the lane whose B operand it is, and all layers' weight loads issued at the top of the kernel.
//
// Work split of one product  out[32][N] = in[32][K] . B[K][N]  over the 16 waves: `ntp` column tiles of 32 (1, 2 or 4) times
// `ks` = 16 / ntp slices of the contraction index; slice partials go through LDS and are added in ascending slice order by the
// thread that owns the output.  Inside a slice of length Kc the instruction's two k-slots walk the two halves of the slice:
// step i multiplies k = kb + i (lanes 0..31) and k = kb + Kc/2 + i (lanes 32..63), so a lane reads one contiguous run.
// The summation order is therefore fixed (deterministic, the same for every launch) but not the per-layer kernels' order:
// results agree with them to f32 rounding (tests/test_gpu_dense_head.py).  The weight gradients contract over the batch rows
// in ascending order, one fused multiply-add per row -- the per-layer kernel's order, bit for bit.
#include "pv_common.h"

namespace pv {

constexpr int DC_MAXL = PV_DENSE_CHAIN_MAX_LAYERS, DC_MAXF = 128, DC_THREADS = 1024, DC_WAVES = 16;
// LDS row stride of every [row][feature] image: odd, so that the 32 lanes of an operand read (32 rows, one column) fall into
// 32 different banks
constexpr int DC_LD = DC_MAXF + 1;
constexpr int DC_WPT = DC_MAXF * DC_MAXF / DC_THREADS;   // weight elements per thread and layer (16)

struct DenseChainFwdArgs {
  const float* x;
  const float* w[DC_MAXL];
  const float* bias[DC_MAXL];
  float* y[DC_MAXL];
  int m, k0, n_layers;
  int n[DC_MAXL], relu[DC_MAXL];
};

struct DenseChainBwdArgs {
  const float* x;              // input of layer 0 [m, k0]
  const float* w[DC_MAXL];
  const float* y[DC_MAXL];     // layer outputs (after the ReLU where there is one)
  const float* dy;             // gradient of the last layer's output [m, n_last]
  float* dw[DC_MAXL];
  float* db[DC_MAXL];          // may be NULL per layer
  float* dx;                   // may be NULL
  int m, k0, n_layers;
  int n[DC_MAXL], relu[DC_MAXL];
};

// how the 16 waves split a product with `cols` output columns and a contraction of length `len`
struct DcSplit {
  int ntp, ks, chunk, half;    // column tiles (power of two), contraction slices, slice length (even), half of it
};
__device__ __forceinline__ DcSplit dc_split(int cols, int len) {
  DcSplit s;
  const int nt = (cols + 31) >> 5;
  s.ntp = nt <= 1 ? 1 : (nt == 2 ? 2 : 4);
  s.ks = DC_WAVES / s.ntp;
  s.chunk = (((len + s.ks - 1) / s.ks) + 1) & ~1;
  s.half = s.chunk >> 1;
  return s;
}

// part[slice][row][col] -> sum over slices in ascending order
__device__ __forceinline__ float dc_sum_slices(const float* part, const DcSplit& s, int row, int col) {
  const int stride = 32 * s.ntp * 32;
  const float* p = part + row * (s.ntp * 32) + col;
  float v = p[0];
  for (int q = 1; q < s.ks; ++q) v += p[q * stride];
  return v;
}

__device__ __forceinline__ void dc_store_partial(float* part, const DcSplit& s, int slice, int tile, int lane, const f32x16& acc) {
  const int r = lane & 31, hh = lane >> 5;
  float* p = part + (size_t)slice * (32 * s.ntp * 32) + tile * 32 + r;
#pragma unroll
  for (int j = 0; j < 16; ++j) p[((j & 3) + 8 * (j >> 2) + 4 * hh) * (s.ntp * 32)] = acc[j];
}

__global__ __launch_bounds__(DC_THREADS) void dense_chain_fwd_kernel(DenseChainFwdArgs a) {
  __shared__ float act[32 * DC_LD];               // the current layer's input [row][feature]; rewritten in place by its output
  __shared__ float w_s[DC_MAXF * DC_LD];          // the current layer's weights [n][k]
  __shared__ float part[DC_WAVES * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * 32, rows = min(32, a.m - row0);

  // ---- every layer's weights, requested before anything waits: thread t holds elements t, t + 1024, ... of the row-major
  // matrix (a wave instruction reads 256 contiguous bytes); they go to LDS when the buffer is free.  (A lane fetching its own
  // B operand -- 32 lanes, 32 different rows -- made the address unit walk ~25 000 cache lines: 20 us for 100 KB.) ---------
  float wreg[DC_MAXL][DC_WPT];
  {
    int k = a.k0;
#pragma unroll
    for (int l = 0; l < DC_MAXL; ++l) {
      if (l < a.n_layers) {
        const int total = a.n[l] * k;
#pragma unroll
        for (int i = 0; i < DC_WPT; ++i) {
          const int idx = tid + i * DC_THREADS;
          wreg[l][i] = idx < total ? a.w[l][idx] : 0.f;
        }
        k = a.n[l];
      }
    }
  }
  auto park_weights = [&](int l, int k) {       // wreg[l] -> w_s[n][k]
    const int total = a.n[l] * k;
#pragma unroll
    for (int i = 0; i < DC_WPT; ++i) {
      const int idx = tid + i * DC_THREADS;
      if (idx < total) {
        const int nn = idx / k;
        w_s[nn * DC_LD + (idx - nn * k)] = wreg[l][i];
      }
    }
  };
  // ---- the block's input rows into LDS (zero beyond the rows / features that exist) ---------------------------------------
  for (int i = tid; i < 32 * DC_MAXF; i += DC_THREADS) {
    const int rr = i >> 7, c = i & 127;
    act[rr * DC_LD + c] = (rr < rows && c < a.k0) ? a.x[(size_t)(row0 + rr) * a.k0 + c] : 0.f;
  }
  park_weights(0, a.k0);
  __syncthreads();

  int k = a.k0;
#pragma unroll
  for (int l = 0; l < DC_MAXL; ++l) {
    if (l < a.n_layers) {
      const int n = a.n[l];
      const DcSplit s = dc_split(n, k);
      const int tile = wave % s.ntp, slice = wave / s.ntp;
      const int kb = slice * s.chunk + hh * s.half;
      const int col = tile * 32 + r;
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
      const float* arow = act + r * DC_LD;
      const float* wrow = w_s + min(col, n - 1) * DC_LD;
      const bool col_ok = col < n;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < s.half) {   // wave-uniform
          const int kk = kb + i;
          const bool ok = col_ok && kk < k;                 // (beyond k / n: a zero weight operand, a finite activation)
          const int kc = min(kk, DC_MAXF - 1);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[kc], ok ? wrow[kc] : 0.f, acc, 0, 0, 0);
        }
      }
      dc_store_partial(part, s, slice, tile, lane, acc);
      __syncthreads();      // partials complete; nobody reads act / w_s of this layer any more
      for (int i = tid; i < 32 * DC_MAXF; i += DC_THREADS) {
        const int rr = i >> 7, c = i & 127;
        float v = 0.f;
        if (c < n) {
          v = dc_sum_slices(part, s, rr, c) + a.bias[l][c];
          if (a.relu[l]) v = v > 0.f ? v : 0.f;
          if (rr < rows) a.y[l][(size_t)(row0 + rr) * n + c] = v;
        }
        act[rr * DC_LD + c] = v;
      }
      if (l + 1 < a.n_layers) park_weights(l + 1, n);
      __syncthreads();
      k = n;
    }
  }
}

// Backward of the chain for one block of at most 32 rows.  Per layer, last to first:
//   g      = gradient at the layer's output, already multiplied by its ReLU derivative        (LDS, [row][feature])
//   dW     = g^T . in        tiles of 32 x 32 over the waves, contraction over the rows in ascending order
//   db     = column sums of g, rows in ascending order
//   d(in)  = g . W           split like the forward products (column tiles x slices of the contraction over the layer's outputs)
__global__ __launch_bounds__(DC_THREADS) void dense_chain_bwd_kernel(DenseChainBwdArgs a) {
  __shared__ float in_s[DC_MAXL][32 * DC_LD];     // in_s[l] = input of layer l (x, y[0], y[1])
  __shared__ float g_s[2][32 * DC_LD];
  __shared__ float part[DC_WAVES * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const int L = a.n_layers, rows = a.m;

  // ---- B operands of the d(in) products: W_l[cb .. cb + half)[k tile * 32 + r], for every layer, requested first ----------
  float wreg[DC_MAXL][16];
  {
#pragma unroll
    for (int l = 0; l < DC_MAXL; ++l) {
      if (l < L && (l > 0 || a.dx)) {
        const int k = l == 0 ? a.k0 : a.n[l - 1];
        const DcSplit s = dc_split(k, a.n[l]);
        const int tile = wave % s.ntp, slice = wave / s.ntp;
        const int col = tile * 32 + r, cb = slice * s.chunk + hh * s.half;
        const bool col_ok = col < k;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int nn = cb + i;
          wreg[l][i] = (col_ok && i < s.half && nn < a.n[l]) ? a.w[l][(size_t)nn * k + col] : 0.f;
        }
      }
    }
  }
  // ---- stage the layer inputs and the incoming gradient ----------------------------------------------------------------------
#pragma unroll
  for (int l = 0; l < DC_MAXL; ++l) {
    if (l < L) {
      const int k = l == 0 ? a.k0 : a.n[l - 1];
      const float* src = l == 0 ? a.x : a.y[l - 1];
      for (int i = tid; i < 32 * DC_MAXF; i += DC_THREADS) {
        const int rr = i >> 7, c = i & 127;
        in_s[l][rr * DC_LD + c] = (rr < rows && c < k) ? src[(size_t)rr * k + c] : 0.f;
      }
    }
  }
  {
    const int n = a.n[L - 1];
    const float* yl = a.y[L - 1];
    for (int i = tid; i < 32 * DC_MAXF; i += DC_THREADS) {
      const int rr = i >> 7, c = i & 127;
      float v = 0.f;
      if (rr < rows && c < n) {
        v = a.dy[(size_t)rr * n + c];
        if (a.relu[L - 1] && !(yl[(size_t)rr * n + c] > 0.f)) v = 0.f;
      }
      g_s[0][rr * DC_LD + c] = v;
    }
  }
  __syncthreads();

  int cur = 0;
#pragma unroll
  for (int l = DC_MAXL - 1; l >= 0; --l) {
    if (l < L) {
      const int n = a.n[l], k = l == 0 ? a.k0 : a.n[l - 1];
      const float* g = g_s[cur];
      // ---- dW tile of this wave: rows nt0 .. +31 of dW (the layer's outputs), columns kt0 .. +31 (its inputs) -------------
      {
        const int ktiles = (k + 31) >> 5, tiles = ((n + 31) >> 5) * ktiles;
        if (wave < tiles) {
          const int nt0 = (wave / ktiles) * 32, kt0 = (wave % ktiles) * 32;
          f32x16 acc;
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] = 0.f;
          const float* gp = g + hh * DC_LD + min(nt0 + r, DC_MAXF - 1);
          const float* xp = in_s[l] + hh * DC_LD + min(kt0 + r, DC_MAXF - 1);
#pragma unroll
          for (int i = 0; i < 16; ++i)      // batch rows 2 i, 2 i + 1 (zero beyond `rows`)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[2 * i * DC_LD], xp[2 * i * DC_LD], acc, 0, 0, 0);
          const int col = kt0 + r;
          if (col < k) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int row = nt0 + (j & 3) + 8 * (j >> 2) + 4 * hh;
              if (row < n) a.dw[l][(size_t)row * k + col] = acc[j];
            }
          }
        }
      }
      // ---- db -------------------------------------------------------------------------------------------------------------
      if (a.db[l] && tid < n) {
        float sacc = 0.f;
        for (int rr = 0; rr < rows; ++rr) sacc += g[rr * DC_LD + tid];
        a.db[l][tid] = sacc;
      }
      // ---- d(in) = g . W_l ------------------------------------------------------------------------------------------------
      if (l > 0 || a.dx) {
        const DcSplit s = dc_split(k, n);
        const int tile = wave % s.ntp, slice = wave / s.ntp;
        const int cb = slice * s.chunk + hh * s.half;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const float* grow = g + r * DC_LD;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < s.half) {
            const int nn = min(cb + i, DC_MAXF - 1);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(grow[nn], wreg[l][i], acc, 0, 0, 0);
          }
        }
        dc_store_partial(part, s, slice, tile, lane, acc);
        __syncthreads();
        for (int i = tid; i < 32 * DC_MAXF; i += DC_THREADS) {
          const int rr = i >> 7, c = i & 127;
          float v = 0.f;
          if (c < k) {
            v = dc_sum_slices(part, s, rr, c);
            if (l == 0) {
              if (rr < rows) a.dx[(size_t)rr * k + c] = v;
            } else if (a.relu[l - 1] && !(in_s[l][rr * DC_LD + c] > 0.f)) {
              v = 0.f;      // the producing layer's ReLU derivative (in_s[l] is its output)
            }
          }
          g_s[cur ^ 1][rr * DC_LD + c] = v;
        }
        __syncthreads();
        cur ^= 1;
      }
    }
  }
}

static bool dc_desc_ok(const pv_dense_chain* d) {
  if (!d || d->n_layers < 1 || d->n_layers > DC_MAXL || d->m < 1 || d->k0 < 1 || d->k0 > DC_MAXF) return false;
  for (int l = 0; l < d->n_layers; ++l)
    if (d->n[l] < 1 || d->n[l] > DC_MAXF) return false;
  return true;
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_dense_chain_fwd_f32(const float* x, const float* const* w, const float* const* bias, float* const* y,
                           const pv_dense_chain* d, void* stream) {
  PV_REQUIRE(x && w && bias && y && d, PV_EINVAL, "pv_dense_chain_fwd_f32: null pointer");
  PV_REQUIRE(dc_desc_ok(d), PV_ESIZE, "pv_dense_chain_fwd_f32: 1..%d layers of at most %d features", DC_MAXL, DC_MAXF);
  DenseChainFwdArgs a{};
  a.x = x;
  a.m = d->m, a.k0 = d->k0, a.n_layers = d->n_layers;
  for (int l = 0; l < d->n_layers; ++l) {
    PV_REQUIRE(w[l] && bias[l] && y[l], PV_EINVAL, "pv_dense_chain_fwd_f32: null pointer (layer %d)", l);
    a.w[l] = w[l], a.bias[l] = bias[l], a.y[l] = y[l], a.n[l] = d->n[l], a.relu[l] = d->relu[l] ? 1 : 0;
  }
  hipLaunchKernelGGL(dense_chain_fwd_kernel, dim3((unsigned)((d->m + 31) / 32)), dim3(DC_THREADS), 0, as_stream(stream), a);
  return check_launch("pv_dense_chain_fwd_f32");
}

int pv_dense_chain_bwd_f32(const float* x, const float* const* w, const float* const* y, const float* dy, float* const* dw,
                           float* const* db, float* dx, const pv_dense_chain* d, void* stream) {
  PV_REQUIRE(x && w && y && dy && dw && db && d, PV_EINVAL, "pv_dense_chain_bwd_f32: null pointer");
  PV_REQUIRE(dc_desc_ok(d), PV_ESIZE, "pv_dense_chain_bwd_f32: 1..%d layers of at most %d features", DC_MAXL, DC_MAXF);
  PV_REQUIRE(d->m <= 32, PV_ESIZE, "pv_dense_chain_bwd_f32: m=%d > 32 rows (one block: the weight gradients contract over it)", d->m);
  DenseChainBwdArgs a{};
  a.x = x, a.dy = dy, a.dx = dx;
  a.m = d->m, a.k0 = d->k0, a.n_layers = d->n_layers;
  for (int l = 0; l < d->n_layers; ++l) {
    PV_REQUIRE(w[l] && y[l] && dw[l], PV_EINVAL, "pv_dense_chain_bwd_f32: null pointer (layer %d)", l);
    a.w[l] = w[l], a.y[l] = y[l], a.dw[l] = dw[l], a.db[l] = db[l], a.n[l] = d->n[l], a.relu[l] = d->relu[l] ? 1 : 0;
  }
  hipLaunchKernelGGL(dense_chain_bwd_kernel, dim3(1), dim3(DC_THREADS), 0, as_stream(stream), a);
  return check_launch("pv_dense_chain_bwd_f32");
}

}  // extern "C"
