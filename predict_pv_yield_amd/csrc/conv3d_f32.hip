// Exact-fp32 Conv3D (3x3x3, stride 1) on the reference layout NCDHW — the tight-parity path.
// replaces: F.conv3d + F.relu and their autograd (dgrad / wgrad) as called by
//   predict_pv_yield/models/conv3d/model.py:80-90,117-120  (padding 0)
//   predict_pv_yield/models/conv3d/model_sat_nwp.py:102-115 (padding (1,0,0))
// Direct convolution on the f32 VALU: one f32 FMA chain per output, weights broadcast from LDS.
// The MFMA bf16 kernels in conv3d_bf16.hip are the throughput path; this file is the one whose
// results are compared with the torch-CPU oracle at rtol 1e-4.
#include "pv_common.h"

namespace pv {

constexpr int CO_BLK = 8;  // output channels per thread (fwd / dgrad)

// One thread = one output voxel x CO_BLK output channels.  flip != 0 turns the kernel into dgrad:
// "input" is dy (gated by the ReLU mask), channels swap roles and taps are mirrored.
__global__ __launch_bounds__(256) void conv3d_direct_f32(
    const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ y, int c_in, int c_out, int t_in, int h_in,
    int w_in, int t_out, int h_out, int w_out, int pad_t, int pad_h, int pad_w, int relu, int flip,
    int w_ci_dim /* second dim of the ORIGINAL weight tensor */) {
  extern __shared__ float wl[];  // [c_in][27][CO_BLK]
  const int co0 = blockIdx.y * CO_BLK;
  const int b = blockIdx.z;
  for (int i = threadIdx.x; i < c_in * 27 * CO_BLK; i += blockDim.x) {
    int j = i % CO_BLK;
    int tap = (i / CO_BLK) % 27;
    int ci = i / (CO_BLK * 27);
    int co = co0 + j;
    float v = 0.f;
    if (co < c_out) {
      // forward: w[co][ci][tap]; dgrad: w_orig[ci'(=orig co)][co'(=orig ci)][26 - tap]
      v = flip ? w[((size_t)ci * w_ci_dim + co) * 27 + (26 - tap)] : w[((size_t)co * w_ci_dim + ci) * 27 + tap];
    }
    wl[i] = v;
  }
  __syncthreads();
  const int plane_out = h_out * w_out;
  const int vox_out = t_out * plane_out;
  const size_t plane_in = (size_t)h_in * w_in;
  const size_t vox_in = (size_t)t_in * plane_in;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < vox_out; v += gridDim.x * blockDim.x) {
    int to = v / plane_out;
    int r = v - to * plane_out;
    int ho = r / w_out;
    int wo = r - ho * w_out;
    float acc[CO_BLK];
#pragma unroll
    for (int j = 0; j < CO_BLK; ++j) acc[j] = (bias && co0 + j < c_out) ? bias[co0 + j] : 0.f;
    for (int ci = 0; ci < c_in; ++ci) {
      const float* xc = x + ((size_t)b * c_in + ci) * vox_in;
      const float* gc = gate ? gate + ((size_t)b * c_in + ci) * vox_in : nullptr;
      const float* wc = wl + (size_t)ci * 27 * CO_BLK;
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        int ti = to + kt - pad_t;
        bool t_ok = (unsigned)ti < (unsigned)t_in;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          int hi = ho + kh - pad_h;
          bool h_ok = t_ok && (unsigned)hi < (unsigned)h_in;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            int wi = wo + kw - pad_w;
            float xv = 0.f;
            if (h_ok && (unsigned)wi < (unsigned)w_in) {
              size_t off = (size_t)ti * plane_in + (size_t)hi * w_in + wi;
              xv = xc[off];
              if (gc && !(gc[off] > 0.f)) xv = 0.f;
            }
            const float* wt = wc + (kt * 9 + kh * 3 + kw) * CO_BLK;
#pragma unroll
            for (int j = 0; j < CO_BLK; ++j) acc[j] = fmaf(xv, wt[j], acc[j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < CO_BLK; ++j) {
      if (co0 + j < c_out) {
        float o = acc[j];
        if (relu) o = o > 0.f ? o : 0.f;
        y[((size_t)b * c_out + co0 + j) * vox_out + v] = o;
      }
    }
  }
}

// wgrad: one block = (group of CO_G output channels, one input channel); every thread keeps
// 27 x CO_G partial sums over its share of the voxels, then a deterministic block reduction.
constexpr int CO_G = 4;

__global__ __launch_bounds__(256) void conv3d_wgrad_f32(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gate,
    float* __restrict__ dw, int batch, int c_in, int c_out, int t_in, int h_in, int w_in, int t_out,
    int h_out, int w_out, int pad_t, int pad_h, int pad_w) {
  const int ci = blockIdx.x;
  const int co0 = blockIdx.y * CO_G;
  const int plane_out = h_out * w_out;
  const int vox_out = t_out * plane_out;
  const size_t plane_in = (size_t)h_in * w_in;
  const size_t vox_in = (size_t)t_in * plane_in;
  float acc[CO_G][27];
#pragma unroll
  for (int g = 0; g < CO_G; ++g)
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[g][k] = 0.f;

  const long long total = (long long)batch * vox_out;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    int b = (int)(i / vox_out);
    int v = (int)(i - (long long)b * vox_out);
    int to = v / plane_out;
    int r = v - to * plane_out;
    int ho = r / w_out;
    int wo = r - ho * w_out;
    float g[CO_G];
#pragma unroll
    for (int j = 0; j < CO_G; ++j) {
      float d = 0.f;
      if (co0 + j < c_out) {
        size_t off = ((size_t)b * c_out + co0 + j) * vox_out + v;
        d = dy[off];
        if (gate && !(gate[off] > 0.f)) d = 0.f;
      }
      g[j] = d;
    }
    const float* xc = x + ((size_t)b * c_in + ci) * vox_in;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      int ti = to + kt - pad_t;
      bool t_ok = (unsigned)ti < (unsigned)t_in;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        int hi = ho + kh - pad_h;
        bool h_ok = t_ok && (unsigned)hi < (unsigned)h_in;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          int wi = wo + kw - pad_w;
          float xv = 0.f;
          if (h_ok && (unsigned)wi < (unsigned)w_in) xv = xc[(size_t)ti * plane_in + (size_t)hi * w_in + wi];
#pragma unroll
          for (int j = 0; j < CO_G; ++j) acc[j][kt * 9 + kh * 3 + kw] = fmaf(g[j], xv, acc[j][kt * 9 + kh * 3 + kw]);
        }
      }
    }
  }
  // block reduction: wave shuffle, then 4 waves through LDS
  __shared__ float red[4][CO_G * 27];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < CO_G; ++j)
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      float v = acc[j][k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][j * 27 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < CO_G * 27) {
    int j = threadIdx.x / 27, k = threadIdx.x % 27;
    if (co0 + j < c_out) {
      float s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
      dw[((size_t)(co0 + j) * c_in + ci) * 27 + k] = s;
    }
  }
}

// dbias[co] = sum over (b, voxels) of dy ⊙ (y > 0); one block per channel
__global__ __launch_bounds__(256) void conv3d_dbias_f32(const float* __restrict__ dy,
                                                         const float* __restrict__ gate,
                                                         float* __restrict__ db, int batch, int c_out,
                                                         int vox_out) {
  const int co = blockIdx.x;
  float s = 0.f;
  const long long total = (long long)batch * vox_out;
  for (long long i = threadIdx.x; i < total; i += blockDim.x) {
    int b = (int)(i / vox_out);
    int v = (int)(i - (long long)b * vox_out);
    size_t off = ((size_t)b * c_out + co) * vox_out + v;
    float d = dy[off];
    if (gate && !(gate[off] > 0.f)) d = 0.f;
    s += d;
  }
  __shared__ float red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) db[co] = ((red[0] + red[1]) + red[2]) + red[3];
}

static int check_dims(const pv_conv3d_dims* d, const char* who) {
  PV_REQUIRE(d, PV_EINVAL, "%s: null dims", who);
  PV_REQUIRE(d->batch > 0 && d->c_in > 0 && d->c_out > 0 && d->t_in > 0 && d->h_in > 0 && d->w_in > 0,
             PV_EINVAL, "%s: non-positive dimension", who);
  PV_REQUIRE(d->pad_t >= 0 && d->pad_t <= 2 && d->pad_h >= 0 && d->pad_h <= 2 && d->pad_w >= 0 && d->pad_w <= 2,
             PV_EINVAL, "%s: padding must be 0..2", who);
  PV_REQUIRE(d->t_in + 2 * d->pad_t >= 3 && d->h_in + 2 * d->pad_h >= 3 && d->w_in + 2 * d->pad_w >= 3, PV_ESIZE,
             "%s: input smaller than the 3x3x3 kernel", who);
  return PV_OK;
}

}  // namespace pv

using namespace pv;

extern "C" {

int pv_conv3d_fwd_f32(const float* x, const float* w, const float* bias, float* y, const pv_conv3d_dims* d,
                      int relu, void* stream) {
  int rc = check_dims(d, "pv_conv3d_fwd_f32");
  if (rc) return rc;
  PV_REQUIRE(x && w && y, PV_EINVAL, "pv_conv3d_fwd_f32: null pointer");
  int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  size_t lds = (size_t)d->c_in * 27 * CO_BLK * sizeof(float);
  PV_REQUIRE(lds <= 160 * 1024, PV_ESIZE, "pv_conv3d_fwd_f32: c_in=%d too large for the LDS weight tile", d->c_in);
  int vox = to * ho * wo;
  dim3 grid((unsigned)std::min((vox + 255) / 256, 4096), (unsigned)((d->c_out + CO_BLK - 1) / CO_BLK), (unsigned)d->batch);
  hipLaunchKernelGGL(conv3d_direct_f32, grid, dim3(256), lds, as_stream(stream), x, (const float*)nullptr, w, bias,
                     y, d->c_in, d->c_out, d->t_in, d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w,
                     relu ? 1 : 0, 0, d->c_in);
  return check_launch("pv_conv3d_fwd_f32");
}

int pv_conv3d_bwd_data_f32(const float* dy, const float* y_relu_mask, const float* w, float* dx,
                           const pv_conv3d_dims* d, void* stream) {
  int rc = check_dims(d, "pv_conv3d_bwd_data_f32");
  if (rc) return rc;
  PV_REQUIRE(dy && w && dx, PV_EINVAL, "pv_conv3d_bwd_data_f32: null pointer");
  // dgrad = correlation of dy (extent To,Ho,Wo; padding 2-p) with mirrored, channel-swapped weights
  int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  size_t lds = (size_t)d->c_out * 27 * CO_BLK * sizeof(float);
  PV_REQUIRE(lds <= 160 * 1024, PV_ESIZE, "pv_conv3d_bwd_data_f32: c_out=%d too large", d->c_out);
  int vox = d->t_in * d->h_in * d->w_in;
  dim3 grid((unsigned)std::min((vox + 255) / 256, 4096), (unsigned)((d->c_in + CO_BLK - 1) / CO_BLK), (unsigned)d->batch);
  hipLaunchKernelGGL(conv3d_direct_f32, grid, dim3(256), lds, as_stream(stream), dy, y_relu_mask, w,
                     (const float*)nullptr, dx, /*c_in=*/d->c_out, /*c_out=*/d->c_in, to, ho, wo, d->t_in, d->h_in,
                     d->w_in, 2 - d->pad_t, 2 - d->pad_h, 2 - d->pad_w, 0, 1, d->c_in);
  return check_launch("pv_conv3d_bwd_data_f32");
}

int pv_conv3d_bwd_weight_f32(const float* x, const float* dy, const float* y_relu_mask, float* dw, float* dbias,
                             const pv_conv3d_dims* d, void* stream) {
  int rc = check_dims(d, "pv_conv3d_bwd_weight_f32");
  if (rc) return rc;
  PV_REQUIRE(x && dy, PV_EINVAL, "pv_conv3d_bwd_weight_f32: null pointer");
  int to = d->t_in + 2 * d->pad_t - 2, ho = d->h_in + 2 * d->pad_h - 2, wo = d->w_in + 2 * d->pad_w - 2;
  if (dw) {
    dim3 grid((unsigned)d->c_in, (unsigned)((d->c_out + CO_G - 1) / CO_G));
    hipLaunchKernelGGL(conv3d_wgrad_f32, grid, dim3(256), 0, as_stream(stream), x, dy, y_relu_mask, dw, d->batch,
                       d->c_in, d->c_out, d->t_in, d->h_in, d->w_in, to, ho, wo, d->pad_t, d->pad_h, d->pad_w);
    rc = check_launch("pv_conv3d_bwd_weight_f32");
    if (rc) return rc;
  }
  if (dbias) {
    hipLaunchKernelGGL(conv3d_dbias_f32, dim3((unsigned)d->c_out), dim3(256), 0, as_stream(stream), dy, y_relu_mask,
                       dbias, d->batch, d->c_out, to * ho * wo);
    rc = check_launch("pv_conv3d_bwd_weight_f32(dbias)");
  }
  return rc;
}

}  // extern "C"
