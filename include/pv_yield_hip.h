/*
 * pv_yield_hip.h — C ABI of libpvyield_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the dense hot path of openclimatefix/predict_pv_yield
 * (SURVEY.md §8b).  Every entry point replaces one third-party operator call
 * site of the reference; the citation after "replaces:" is the reference
 * file:line (paths under the upstream repo; notebook lines are raw .ipynb
 * JSON lines).
 *
 * Conventions
 *   - plain pointers + sizes, no torch / C++ types; all pointers are DEVICE
 *     pointers unless a parameter says "host";
 *   - the caller owns every buffer (including workspaces, sized by the
 *     *_workspace_bytes queries);
 *   - every launch goes to the caller's stream (`stream` is a hipStream_t
 *     passed as void*; NULL = the null stream); calls are asynchronous and
 *     never synchronise, allocate or free (graph-capturable);
 *   - return value: PV_OK (0) or a negative PV_E* code, never throws;
 *     pv_last_error() returns a thread-local description of the last failure;
 *   - no global state; re-entrant across streams and devices.
 */
#ifndef PV_YIELD_HIP_H
#define PV_YIELD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_ABI_VERSION 1

enum {
  PV_OK = 0,
  PV_EINVAL = -1,   /* bad argument (null pointer, non-positive size, bad enum) */
  PV_ESIZE = -2,    /* size not supported by this kernel / workspace too small  */
  PV_ELAUNCH = -3   /* hip launch error (hipGetLastError() != hipSuccess)       */
};

/* cv2 border modes (same numeric values as OpenCV's cv::BorderTypes). */
enum { PV_BORDER_CONSTANT = 0, PV_BORDER_REPLICATE = 1 };

/* u8 conversions of 10-bit satellite counts. */
enum {
  PV_U8_ROUND_DIV4 = 0,   /* round_half_even(x / 4)      notebooks/13_...ipynb:112-119 */
  PV_U8_TRUNC_SCALE = 1   /* trunc(x / 1023 * 255)       notebooks/optical_flow_1.ipynb:129-134 */
};

int pv_abi_version(void);
const char* pv_last_error(void);

/* ------------------------------------------------------------------------ */
/* Optical-flow advection (SURVEY.md §8a rows a-9 … a-15)                    */
/* ------------------------------------------------------------------------ */

/* replaces: convert_10bpp_to_uint8, notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:112-119
 * (mode PV_U8_ROUND_DIV4) and the truncating variant notebooks/optical_flow_1.ipynb:129-134.
 * `range_flag` (device int32, may be NULL) is set to 1 if any value leaves
 * [0,255] after conversion (the reference asserts on that); values saturate. */
int pv_u8_from_10bit_i16(const int16_t* src, uint8_t* dst, size_t n, int mode,
                         int32_t* range_flag, void* stream);
int pv_u8_from_10bit_f32(const float* src, uint8_t* dst, size_t n, int mode,
                         int32_t* range_flag, void* stream);

/* Farnebäck parameters = the positional arguments of cv.calcOpticalFlowFarneback.
 * Reference call site: notebooks/13_...ipynb:133-135 (0.5, 2, 40, 3, 5, 0.7,
 * OPTFLOW_FARNEBACK_GAUSSIAN); identical in optical_flow_1.ipynb:217. */
typedef struct pv_farneback_params {
  double pyr_scale;   /* 0.5 */
  int32_t levels;     /* 2   */
  int32_t winsize;    /* 40  */
  int32_t iterations; /* 3   */
  int32_t poly_n;     /* 5   (5 or 7) */
  double poly_sigma;  /* 0.7 */
  int32_t flags;      /* 256 = OPTFLOW_FARNEBACK_GAUSSIAN (the only mode built) */
} pv_farneback_params;

#define PV_OPTFLOW_FARNEBACK_GAUSSIAN 256

/* Bytes of scratch pv_farneback_batch_u8 needs for n_pairs images of h×w. */
int pv_farneback_workspace_bytes(int64_t n_pairs, int32_t h, int32_t w,
                                 const pv_farneback_params* params, size_t* bytes);

/* replaces: cv.calcOpticalFlowFarneback(prev, next, None, ...) called once per
 * consecutive pair by compute_optical_flow / _compute_optical_flow
 * (notebooks/13_...ipynb:122-135, 175-240) — the whole process-pool fan-out is
 * one batched launch sequence here.
 *   prev, next : u8 images, row-major h×w.  Pair i = (group g = i / pairs_per_group,
 *                q = i % pairs_per_group) reads prev + g*group_stride + q*prev_stride (bytes) and
 *                next + g*group_stride + q*next_stride.  For B frame stacks [B,T,h,w]:
 *                next = prev + h*w, strides h*w, pairs_per_group = T-1, group_stride = T*h*w.
 *                pairs_per_group <= 0 means one group.
 *   flow       : f32 [n_pairs, h, w, 2] (x then y displacement, OpenCV layout) */
int pv_farneback_batch_u8(const uint8_t* prev, const uint8_t* next,
                          int64_t prev_stride, int64_t next_stride,
                          int64_t pairs_per_group, int64_t group_stride,
                          float* flow, int64_t n_pairs, int32_t h, int32_t w,
                          const pv_farneback_params* params,
                          void* workspace, size_t workspace_bytes, void* stream);

/* replaces: weighted_average = np.average(flows, axis=0, weights=range(1,N+1)).astype(f32)
 * (notebooks/optical_flow_1.ipynb:293-294).  flows: f32 [n_groups, n_per_group, elems];
 * weights: host double[n_per_group] (NULL = 1..n_per_group); out: f32 [n_groups, elems].
 * Accumulates in float64 in index order, divides by the float64 weight sum, rounds once. */
int pv_flow_weighted_mean_f32(const float* flows, const double* weights_host,
                              float* out, int64_t n_groups, int32_t n_per_group,
                              int64_t elems, void* stream);

/* replaces: remap_image → cv.remap(src, map1 = meshgrid − flow, None, INTER_LINEAR,
 * borderMode, borderValue)  (notebooks/13_...ipynb:259-281 BORDER_CONSTANT/NaN;
 * notebooks/optical_flow_1.ipynb:415-430 BORDER_REPLICATE), batched over images
 * and over the `n_steps` extrapolation steps of compute_optical_flow_predictions
 * (notebooks/13_...ipynb:317-323: flow * forecast_step).
 *   src  : [n_images, h, w]            image i at src + i*src_stride (elements)
 *   flow : f32 [n_images, h, w, 2]     field i at flow + i*flow_stride (elements)
 *   dst  : [n_images, n_steps, h, w]   image (i,s) at dst + i*dst_image_stride + s*dst_step_stride
 *   step s uses the displacement field  flow * (step0 + s)  (f32 multiply),
 *   map = (x − flow.x*k, y − flow.y*k) in f32, quantised to 1/32 px as cv.remap does. */
int pv_remap_bilinear_f32(const float* src, int64_t src_stride,
                          const float* flow, int64_t flow_stride,
                          float* dst, int64_t dst_image_stride, int64_t dst_step_stride,
                          int64_t n_images, int32_t n_steps, float step0,
                          int32_t h, int32_t w, int border_mode, float border_value,
                          void* stream);
int pv_remap_bilinear_u8(const uint8_t* src, int64_t src_stride,
                         const float* flow, int64_t flow_stride,
                         uint8_t* dst, int64_t dst_image_stride, int64_t dst_step_stride,
                         int64_t n_images, int32_t n_steps, float step0,
                         int32_t h, int32_t w, int border_mode, uint8_t border_value,
                         void* stream);

/* Config-3 prologue in one pass: raw counts [B, T, C, frame] (time-major) -> u8 [B, C, T, frame] (pv_u8_from_10bit
 * rounding, `mode`) and the normalised frames (x - mean[c]) / std[c] written into slices 0..T-1 of out
 * [B, C, t_out, frame] f32 (t_out >= T: room for the advected frames).  Same bits as pv_u8_from_10bit + pv_normalise on
 * the permuted tensor; frame must be a multiple of 8. */
int pv_prepare_stacks_i16(const int16_t* raw, uint8_t* u8, float* out, int64_t batch, int32_t t, int32_t c, int64_t frame,
                          int32_t t_out, int mode, const float* mean, const float* std_, int32_t* range_flag, void* stream);
int pv_prepare_stacks_f32(const float* raw, uint8_t* u8, float* out, int64_t batch, int32_t t, int32_t c, int64_t frame,
                          int32_t t_out, int mode, const float* mean, const float* std_, int32_t* range_flag, void* stream);

/* replaces: satellite_data -= SAT_IMAGE_MEAN; satellite_data /= SAT_IMAGE_STD
 * (notebooks/13_...ipynb:345-346, 463-464; per-channel constants
 * predict_pv_yield/netcdf_dataset.py:19-32).  dst[i] = (src[i] − mean[c]) / std[c]
 * with c = (i / inner) % n_channels; true f32 division.  mean/std: device f32[n_channels]. */
int pv_normalise_i16(const int16_t* src, float* dst, size_t n, int64_t inner,
                     int32_t n_channels, const float* mean, const float* std_,
                     void* stream);
int pv_normalise_f32(const float* src, float* dst, size_t n, int64_t inner,
                     int32_t n_channels, const float* mean, const float* std_,
                     void* stream);

/* ------------------------------------------------------------------------ */
/* Conv3D PV-yield model (SURVEY.md §8a rows a-2 … a-5)                      */
/* ------------------------------------------------------------------------ */

/* Geometry of one 3x3x3, stride-1, dilation-1, groups-1 convolution
 * (F.conv3d as called by nn.Conv3d in predict_pv_yield/models/conv3d/model.py:80-90,117-120;
 * padding (1,0,0): model_sat_nwp.py:102-115; padding 1: perceiver_conv3d_nwp_sat.py:47-53). */
typedef struct pv_conv3d_dims {
  int32_t batch;
  int32_t c_in, c_out;
  int32_t t_in, h_in, w_in;     /* input extent                       */
  int32_t pad_t, pad_h, pad_w;  /* symmetric zero padding, 0..2       */
  /* output extent is t_in + 2*pad_t - 2 etc. */
} pv_conv3d_dims;

/* ---- general Conv3D / MaxPool3d / MSE (optical-flow notebook model, Conv3dMaxPool) ---------------- */
/* Geometry of one Conv3D or MaxPool3d with kernel extents 1..3, any stride, symmetric zero (conv) / -inf (pool)
 * padding 0..2, dilation 1, groups 1.  Output extent = (in + 2*pad - k) / stride + 1 (floor).
 * replaces: nn.Conv3d(kernel_size=(2,3,3), padding=(0,1,1)[, stride=(1,2,2)]) of LitAutoEncoder
 * (notebooks/13_3d_conv_with_optical_flow_predictions.ipynb:969-985) and nn.MaxPool3d(3, stride=(1,2,2), padding=1)
 * (predict_pv_yield/models/perceiver/perceiver_conv3d_nwp_sat.py:53-57; there c_out is ignored, set it to c_in). */
typedef struct pv_conv3d_geom {
  int32_t batch;
  int32_t c_in, c_out;
  int32_t t_in, h_in, w_in;
  int32_t k_t, k_h, k_w;
  int32_t stride_t, stride_h, stride_w;
  int32_t pad_t, pad_h, pad_w;
} pv_conv3d_geom;

int pv_conv3d_general_out_extent(const pv_conv3d_geom* d, int32_t* t_out, int32_t* h_out, int32_t* w_out);
/* y[B,Co,To,Ho,Wo] = conv3d(x[B,Ci,Ti,Hi,Wi], w[Co,Ci,kt,kh,kw], stride, padding) + bias, optional fused ReLU. */
int pv_conv3d_general_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                              const pv_conv3d_geom* d, int relu, void* stream);
/* dx from dy ⊙ (y_relu_mask > 0 if given); `d` describes the FORWARD conv.  x_relu_mask (may be NULL): the layer
 * input when it is itself a ReLU output — dx is then zeroed where x <= 0, i.e. it leaves this kernel already gated
 * for the previous layer (which passes y_relu_mask = NULL). */
int pv_conv3d_general_bwd_data_f32(const float* dy, const float* y_relu_mask, const float* w, float* dx,
                                   const float* x_relu_mask, const pv_conv3d_geom* d, void* stream);
/* dw[Co,Ci,kt,kh,kw], dbias[Co] (either may be NULL); split over position slabs held in `ws`, summed in slab order.
 * Kernel extents (2,3,3), (3,3,3), (1,3,3), (1,1,1). */
int pv_conv3d_general_bwd_weight_workspace_bytes(const pv_conv3d_geom* d, size_t* bytes);
int pv_conv3d_general_bwd_weight_f32(const float* x, const float* dy, const float* y_relu_mask, float* dw,
                                     float* dbias, const pv_conv3d_geom* d, void* ws, size_t ws_bytes,
                                     void* stream);
/* MaxPool3d over x[B*C planes][Ti,Hi,Wi]; argmax (may be NULL) = flat winner offset inside the plane stack, first
 * maximum wins, NaN propagates (torch CPU semantics).  bwd overwrites dx by gathering dy through argmax. */
int pv_maxpool3d_fwd_f32(const float* x, float* y, int32_t* argmax, const pv_conv3d_geom* d, void* stream);
int pv_maxpool3d_bwd_f32(const float* dy, const int32_t* argmax, float* dx, const pv_conv3d_geom* d, void* stream);
/* replaces: F.mse_loss(y_hat, y) (13_…ipynb:1008).  out: device f32[1]; grad (may be NULL) = 2 (y_hat-y)/n * grad_scale */
int pv_mse_loss_f32(const float* y_hat, const float* y, int64_t n, float grad_scale, float* out, float* grad,
                    void* stream);

/* ---- bf16 MFMA path (activations NDHWC bf16, channel count padded) ------ */
/* Channel padding used by the bf16 path for a layer with c real channels. */
int pv_bf16_cpad(int32_t c);   /* 16 for c<=16, 32 for c<=32, else PV_ESIZE */

/* x[B,C,T,H,W] f32 (reference layout) → xp[B,T,H,W,CPAD] bf16 (zero channel padding). */
/* The two-term HALF-FLOAT split x s = h + l (h = rne_f16(x s), l = rne_f16(x s - h): 22 significant bits), s = 2^(14 - e) from
 * the tensor's largest magnitude (max |x| < 2^e; found by a pass of the same call): three matrix-core launches
 * (pv_conv3d_bwd_weight_f16 on (l,h), (h,l), (h,h), summed in that order, un-scaled by state[2] of both tensors) give the weight
 * gradient of the f32 model (predict_pv_yield/models/conv3d/model.py:80-90 under autograd) where the three-term bf16 split
 * needs six.  state: 3 device words: [0] bits of max |x| (scratch), [1] = s, [2] = 1 / s.  have_max != 0: state[0] already holds
 * the bits of max |x| (pv_relu_gate_max_f32 produced x) and the pass that finds it is skipped.  have_max & 2: images of 32 channels
 * (64-byte voxels) whatever c, zero padded -- what pv_conv3d_fwd_f16_f32out reads.  Same alignment rules as above. */
int pv_pack_split2_ncdhw_f32_to_ndhwc_f16(const float* x, uint16_t* xp_h, uint16_t* xp_l, float* state, int32_t have_max, int32_t batch,
                                          int32_t c, int32_t t, int32_t h, int32_t w, void* stream);
int pv_pack_ncdhw_f32_to_ndhwc_bf16(const float* x, uint16_t* xp, int32_t batch, int32_t c,
                                    int32_t t, int32_t h, int32_t w, void* stream);
/* inverse, dropping pad channels (used by tests and for the fc head's NCDHW flatten). */
int pv_unpack_ndhwc_bf16_to_ncdhw_f32(const uint16_t* xp, float* x, int32_t batch, int32_t c,
                                      int32_t t, int32_t h, int32_t w, void* stream);

/* out[B,T,H,W,32] bf16 (NDHWC) = dy[B,C,T,H,W] ⊙ (y[B,C,T,H,W] > 0), both bf16 NCDHW (y may be NULL):
 * brings fc1's input gradient (flatten order of model.py:122) back to the conv layout. */
int pv_repack_gate_ncdhw_to_ndhwc_bf16(const uint16_t* dy, const uint16_t* y_relu_mask, uint16_t* out,
                                       int32_t batch, int32_t c, int32_t t, int32_t h, int32_t w,
                                       void* stream);

/* w[Co,Ci,3,3,3] f32 → MFMA A-fragments bf16 [27][CPAD/16][64 lanes][8]
 * (transpose_flip != 0: the dgrad weights w'[ci][co][26-tap]). c_out must be 32 (or <=32, zero padded). */
size_t pv_conv3d_packed_weight_elems(int32_t c_in_or_out_as_k);
int pv_conv3d_pack_weight_bf16(const float* w, uint16_t* wp, int32_t c_out, int32_t c_in,
                               int transpose_flip, void* stream);

/* y = relu?(conv3d(x ⊙ (gate>0 if gate given)) + bias), x/gate/y NDHWC bf16 with CPAD channels.
 * The same kernel computes dgrad when fed dy (with pad 2−p) and transpose_flip weights.
 * out_gate (may be NULL; NDHWC [B,To,Ho,Wo,32] like y): y is zeroed where out_gate <= 0 -- used by dgrad to
 * apply the ReLU derivative of the layer that produced this layer's input (out_gate = that input), so the
 * next backward kernels read an already-gated gradient.
 * y_ncdhw != 0: y is written as [B,32,To,Ho,Wo] bf16 (the flatten order fc1 expects,
 * predict_pv_yield/models/conv3d/model.py:122) instead of NDHWC. */
/* Multi-job form of pv_conv3d_pack_weight_bf16: one launch packs up to PV_PACK_MAX_JOBS (weight, orientation) pairs
 * (all conv layers, forward and dgrad operators) right after the optimiser step.  `jobs` is a HOST array. */
#define PV_PACK_MAX_JOBS 16
typedef struct pv_pack_job {
  const float* w;      /* [c_out, c_in, 3,3,3] f32 */
  uint16_t* wp;        /* pv_conv3d_packed_weight_elems(contraction channels) bf16 */
  int32_t c_out, c_in, transpose_flip;
} pv_pack_job;
int pv_conv3d_pack_weights_multi_bf16(const pv_pack_job* jobs, int32_t n_jobs, void* stream);

/* 1-bit ReLU masks (both may be NULL): relu_mask_out receives, per output voxel, a u32 whose bit c is (y[.., c] > 0)
 * (NDHWC output only) -- 4 bytes per voxel instead of the 64-byte bf16 voxel; out_gate_mask is such a mask OF out_gate
 * (pass both: kernels without a mask path read the bf16 tensor) and replaces the read of out_gate in the dgrad epilogue.
 * A mask is laid out [B][T][hp][wp] u32 with the plane padded to whole 8 x 32 tiles: pv_relu_mask_dims(h, w, &hp, &wp);
 * the padding words are never read for a voxel that exists. */
int pv_relu_mask_dims(int32_t h, int32_t w, int32_t* hp, int32_t* wp);
int pv_conv3d_fwd_bf16(const uint16_t* x, const uint16_t* gate, const uint16_t* wp,
                       const float* bias, uint16_t* y, const uint16_t* out_gate,
                       const uint32_t* out_gate_mask, uint32_t* relu_mask_out,
                       const pv_conv3d_dims* d, int relu, int y_ncdhw, void* stream);

/* First-layer form: x is the reference's own input tensor, f32 NCDHW [B, c_in, T, H, W] (sat_data of
 * predict_pv_yield/models/conv3d/model.py:112-117), c_in <= 16.  Same result as pv_pack_ncdhw_f32_to_ndhwc_bf16 followed
 * by pv_conv3d_fwd_bf16 (NDHWC output), in one pass over the input: the staging rounds to bf16 on the way into LDS.
 * xp_out (may be NULL): receives the NDHWC bf16 [B,T,H,W,16] image of x that pv_conv3d_bwd_weight_bf16 consumes. */
int pv_conv3d_fwd_bf16_f32in(const float* x, uint16_t* xp_out, const uint16_t* wp, const float* bias, uint16_t* y,
                             uint32_t* relu_mask_out, const pv_conv3d_dims* d, int relu, void* stream);

/* dw[Co,Ci,3,3,3] f32 and dbias[Co] f32 from x (NDHWC bf16) and dy ⊙ (y>0) (NDHWC bf16).
 * workspace: pv_conv3d_bwd_weight_bf16_workspace_bytes(d). Overwrites dw/dbias. */
int pv_conv3d_bwd_weight_bf16_workspace_bytes(const pv_conv3d_dims* d, size_t* bytes);
int pv_conv3d_bwd_weight_bf16(const uint16_t* x, const uint16_t* dy, const uint16_t* y_relu_mask,
                              float* dw, float* dbias, const pv_conv3d_dims* d,
                              void* workspace, size_t workspace_bytes, void* stream);
/* The same on HALF-FLOAT operand images (pv_pack_split2_...; the LDS-staged kernel only: no mask, 16-byte aligned x / dy,
 * workspace as above). */
int pv_conv3d_bwd_weight_f16(const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, const pv_conv3d_dims* d,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- f32 Conv3d forward / data gradient on the 16-bit matrix cores (conv3d_f16x2.hip) --------------------------------------
 * nn.Conv3d in float32 (predict_pv_yield/models/conv3d/model.py:80-90,113-120 with precision 32) for 17..32 -> 32 channel
 * 3x3x3 stride-1 layers: x and w are split in two half-float terms each (pv_pack_split2_ncdhw_f32_to_ndhwc_f16 /
 * pv_conv3d_pack_weight_split2_f16), the three products x_l w_h, x_h w_l, x_h w_h run as three pv_conv3d_fwd_f16_f32out
 * launches into parts[0..2], pv_sum3_ndhwc_to_ncdhw_f32 adds them (in that order), un-scales, adds the bias, applies the ReLU
 * or a ReLU gate and writes the reference's NCDHW layout.  The data gradient is the same three launches on the split of dy with
 * the transposed-and-flipped operator (wp + 2 * elems / 4) and pad = 2 - forward pad.  Agrees with the f32 kernels to ~1e-6
 * relative (22-bit operands, f32 accumulation). */
/* half-float elements of wp: four operators' fragments -- forward (h, l), data gradient (h, l) */
size_t pv_conv3d_split2_weight_elems(void);
/* w [c_out, c_in, 3,3,3] f32 -> wp; state: 69 device floats (bits of max |w|, s, 1 / s, two unused, the 32 absolute row sums of
 * the forward operator, the 32 of the data-gradient operator) */
int pv_conv3d_pack_weight_split2_f16(const float* w, uint16_t* wp, float* state, int32_t c_out, int32_t c_in, void* stream);
/* x [B,T,H,W,32] half floats (one term), wp: ONE operator's fragments (elems / 4 of the above), y [B,To,Ho,Wo,32] f32 = the
 * raw accumulators (no bias, no activation, scaled by s_x s_w); y_is_f16 != 0: y is a HALF-FLOAT image of the accumulators times
 * 2^-12 -- for the two small products (x_l w_h, x_h w_l: 2^-11 of the result, 11 bits of them are all the sum can use).
 * d as for pv_conv3d_fwd_bf16 (its pad: 0..2). */
int pv_conv3d_fwd_f16_f32out(const uint16_t* x, const uint16_t* wp, void* y, int32_t y_is_f16, const pv_conv3d_dims* d, void* stream);
/* 1 when pv_conv3d_fwd_f16_f32out takes these dims (else the caller keeps pv_conv3d_general_fwd_f32 / _bwd_data_f32) */
int pv_conv3d_fwd_f16_f32out_covers(const pv_conv3d_dims* d);
/* parts [3][B][vox][32] f32 -- or, p01_f16 != NULL: parts = p2 alone and p01_f16 = [2][B][vox][32] half floats (y_is_f16 above) --
 * -> y [B,32,vox] f32 = ((p0 + p1) + p2) * sx_state[2] * sw_state[2] + bias (NULL: none), then relu != 0: max(y, 0); gate_h != NULL
 * (the h image of the gating activation, [B][vox][32] half floats): y where it is > 0, else 0.  max_state != NULL: atomicMax of the
 * bits of |y| into max_state[0] (zeroed by the caller).  out_h / out_l != NULL: y's own two-term split ([B][vox][32] half floats
 * each) with the scale s_y written to max_state[1], 1 / s_y to max_state[2]: s_y brings the bound max |x| (sx_state[0]) *
 * the largest of the operator's row sums (sw_state[5..36], data_gradient: [37..68]) + max |bias| below 2^14 -- known before the first element, so no split pass reads y again; y may
 * then be NULL (a consumer that reads the images only).  vox_per_sample % 4 == 0. */
int pv_sum3_ndhwc_to_ncdhw_f32(const float* parts, const uint16_t* p01_f16, const float* sx_state, const float* sw_state,
                               int32_t data_gradient, const float* bias, const uint16_t* gate_h, float* y, uint16_t* out_h,
                               uint16_t* out_l, float* max_state, int32_t relu, int32_t batch, int64_t vox_per_sample, void* stream);

/* ---- fc1 of the f32 model as streams over the weight (linear_f32_skinny.hip) --------------------------------------------------
 * F.linear in float32 (predict_pv_yield/models/conv3d/model.py:92-103,125-130) for m <= 32 rows, n <= 128 outputs, k % 128 == 0,
 * k >= 65 536: exact f32 products (v_mfma_f32_32x32x2_f32), f32 accumulation; the forward's split-k slabs are summed in order. */
int pv_linear_f32_skinny_covers(int32_t m, int32_t n, int64_t k);
size_t pv_linear_fwd_f32_skinny_workspace_bytes(int32_t n);
/* y[m,n] = relu?(x[m,k] w[n,k]^T + bias[n]) */
int pv_linear_fwd_f32_skinny(const float* x, const float* w, const float* bias, float* y, int32_t m, int32_t n, int64_t k, int32_t relu,
                             void* workspace, size_t workspace_bytes, void* stream);
/* dx[m,k] = g[m,n] w[n,k] (g: the output gradient, already multiplied by the ReLU derivative) */
int pv_linear_dx_f32_skinny(const float* g, const float* w, float* dx, int32_t m, int32_t n, int64_t k, void* stream);

/* ---- fully connected head (F.linear; model.py:92-103,125-152) ------------ */
/* y[M,N] = relu?(x[M,K] · w[N,K]^T + bias[N]); fp32, split-K with fp32 slab reduce.
 * workspace: pv_linear_workspace_bytes(M,N,K). */
int pv_linear_workspace_bytes(int32_t m, int32_t n, int64_t k, size_t* bytes);
int pv_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                      int32_t m, int32_t n, int64_t k, int relu,
                      void* workspace, size_t workspace_bytes, void* stream);
/* dx[M,K] = (dy ⊙ (y>0))[M,N] · w[N,K]   (dx may be NULL) ;
 * dw[N,K] = (dy ⊙ (y>0))^T · x ; db[N] = column sums   (dw/db may be NULL). */
int pv_linear_bwd_f32(const float* x, const float* w, const float* dy, const float* y_relu_mask,
                      float* dx, float* dw, float* db,
                      int32_t m, int32_t n, int64_t k, void* stream);

/* bf16 variants for the big fc1 (x bf16 [M,K], w bf16 shadow [N,K], fp32 accumulate/outputs).
 * k must be a multiple of 8, m <= 1024 for the fc1 shapes (n <= 128, n % 16 == 0; rows go in blocks of 64 or 32, each a stream
 * over the weights), m <= 128 otherwise; workspace: pv_linear_bf16_workspace_bytes(M,N,K). */
int pv_linear_bf16_workspace_bytes(int32_t m, int32_t n, int64_t k, size_t* bytes);
int pv_linear_fwd_bf16(const uint16_t* x, const uint16_t* w, const float* bias, float* y,
                       int32_t m, int32_t n, int64_t k, int relu,
                       void* workspace, size_t workspace_bytes, void* stream);
/* dx bf16 [M,K] (may be NULL), dw f32 [N,K], db f32 [N].  gate_dx_by_x != 0: x is a ReLU output; dx leaves multiplied by
 * (x > 0), the ReLU derivative of the layer that produced x (F.relu(conv(..)) in front of fc1, model.py:120-125). */
int pv_linear_bwd_bf16(const uint16_t* x, const uint16_t* w, const float* dy,
                       const float* y_relu_mask, uint16_t* dx, float* dw, float* db,
                       int32_t m, int32_t n, int64_t k, int32_t gate_dx_by_x, void* stream);

/* Fused fc1 weight-gradient + Adam: param[N,K] (f32, updated in place), exp_avg, exp_avg_sq and the bf16 shadow are
 * updated with the gradient (dy ⊙ (y>0))^T · x computed on the fly -- the 0.5 GB gradient is never materialised.
 * Identical arithmetic to pv_linear_bwd_bf16 (dw) followed by pv_adam_step_f32 (single-GPU: no all-reduce between). */
int pv_linear_wgrad_adam_bf16(const uint16_t* x, const float* dy, const float* y_relu_mask, float* param,
                              float* exp_avg, float* exp_avg_sq, uint16_t* bf16_shadow, int32_t m, int32_t n,
                              int64_t k, double lr, double beta1, double beta2, double eps, int32_t step,
                              void* stream);
/* The same pass for the f32 model (precision="fp32"): x in f32, exact f32 products accumulated in batch order, no operand copy
 * to maintain.  Replaces the f32 weight-gradient GEMM (0.5 GB written, read back by the optimiser) + torch.optim.Adam's pass
 * over fc1 (base_model.py:146). */
int pv_linear_wgrad_adam_f32(const float* x, const float* dy, const float* y_relu_mask, float* param, float* exp_avg,
                             float* exp_avg_sq, int32_t m, int32_t n, int64_t k, double lr, double beta1, double beta2,
                             double eps, int32_t step, void* stream);

/* The same pass ALSO produces dx = (dy ⊙ (y>0)) · W (bf16 [M,K], may be NULL) from the pre-update weights it streams, rounded
 * to bf16 as the operand copy held them during the forward: fc1's whole backward in one pass over the matrix (m <= 32,
 * n <= 128) and db[N] = column sums of dy ⊙ (y>0) (may be NULL).  Parameters / moments / operand copy are bit-identical
 * to pv_linear_wgrad_adam_bf16.  gate_dx_by_x != 0: dx is also multiplied by (x > 0) -- x being a ReLU output, that is the
 * ReLU derivative of the layer that produced x (the last Conv3d of model.py:117-120), applied here instead of by the
 * consumer of dx.
 * moments_tiled != 0 (k % 128 == 0): exp_avg / exp_avg_sq are stored tile by tile, [k/128][N][128] (element (n, k) at
 * (k/128)*N*128 + n*128 + k%128), so that a workgroup's share of each is one contiguous block instead of N segments
 * 4*K bytes apart; the caller owns the layout (optim.HipAdam converts at state_dict() / mode changes). */
int pv_linear_wgrad_dx_adam_bf16(const uint16_t* x, const float* dy, const float* y_relu_mask, float* param,
                                 float* exp_avg, float* exp_avg_sq, uint16_t* bf16_shadow, uint16_t* dx, float* db,
                                 int32_t m, int32_t n, int64_t k, double lr, double beta1, double beta2, double eps,
                                 int32_t step, int32_t gate_dx_by_x, int32_t moments_tiled, void* stream);

/* Data-parallel wire format for fc1's gradient (SURVEY.md §7.2 "keep bf16 grads on the wire"): the weight gradient is
 * written once as bf16 [N,K] (half the bytes of the f32 gradient on HBM and on xGMI), all-reduced by RCCL in bf16,
 * and consumed by pv_adam_step_bf16grad (same arithmetic as pv_adam_step_f32 after widening the gradient).
 * m <= 64, n <= 128, 16-byte aligned x / dw: on the matrix cores (g^T as a bf16 hi + lo pair, f32 accumulate:
 * each element within one bf16 ulp + 2^-15 of its terms of the f32 product rounded once); other shapes on the vector ALU. */
int pv_linear_wgrad_bf16out(const uint16_t* x, const float* dy, const float* y_relu_mask, uint16_t* dw_bf16,
                            int32_t m, int32_t n, int64_t k, void* stream);
/* Multi-tensor form of pv_adam_step_f32: one launch steps up to PV_ADAM_MAX_TENSORS parameter tensors that share the
 * hyper-parameters and the step count (the ~14 small tensors of the model: conv weights/biases, fc2..fc4).  `tensors`
 * is a HOST array (copied into the kernel arguments).  Element-wise arithmetic identical to pv_adam_step_f32. */
#define PV_ADAM_MAX_TENSORS 32
typedef struct pv_adam_tensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  uint16_t* bf16_shadow; /* may be NULL */
  uint64_t n;
} pv_adam_tensor;
int pv_adam_step_multi_f32(const pv_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                           double eps, int32_t step, float grad_scale, void* stream);
/* The same Adam step with its six scalars (1-b1, b2, 1-b2, sqrt(1-b2^t), eps, -lr/(1-b1^t)) and the step counter t in DEVICE
 * memory: pv_adam_scalars_advance increments t and recomputes them (one thread), the `_dev` kernels read them.  This is
 * the form a captured HIP graph of the whole train step replays (base_model.py:255-257 stepped every replay; kernel
 * arguments are frozen at capture, device memory is not). */
int pv_adam_scalars_advance(float* scalars_dev /* [6] */, int32_t* step_dev, double lr, double beta1, double beta2, double eps,
                            void* stream);
int pv_adam_step_multi_dev_f32(const pv_adam_tensor* tensors, int32_t n_tensors, const float* scalars_dev, float grad_scale,
                               void* stream);
int pv_linear_wgrad_dx_adam_dev_bf16(const uint16_t* x, const float* dy, const float* y_relu_mask, float* param, float* exp_avg,
                                     float* exp_avg_sq, uint16_t* bf16_shadow, uint16_t* dx, float* db, int32_t m, int32_t n,
                                     int64_t k, const float* adam_scalars_dev, int32_t gate_dx_by_x, int32_t moments_tiled,
                                     void* stream);

/* The same one-pass backward for the K-SHARDED data-parallel mode (HipAdam large_grad_mode "ksharded"): param / exp_avg /
 * exp_avg_sq / bf16_shadow are this rank's COLUMN shard [n][k = K / world] of fc1 (dense, row stride k), x [m][k] the exchanged
 * activations and dy [m][n] the all-gathered, already gated output gradients of ALL m = world x per-GPU-batch samples (any m;
 * taken in blocks of 32).  The weight gradient of the shard over the whole global batch is scaled by grad_scale (1 / world:
 * the averaging of DDP's all-reduce, experiments/003_...py:292-293) on its way into Adam; dx [m][k] = dy . W_old is not, and is
 * multiplied by (x > 0) when gate_dx_by_x.  No gradient or weight of fc1 crosses a link in this mode.  n <= 128, n % 8 == 0,
 * k % 8 == 0, 16-byte aligned buffers; bf16_shadow and dx may be NULL; workspace: caller-owned scratch (query below).
 * moments_tiled: exp_avg / exp_avg_sq in the [k / 128][n][128] tile layout of pv_linear_wgrad_dx_adam_bf16 (k % 128 == 0) -- with
 * world = 1, grad_scale = 1 and dy = dy (.) relu' this is that call for m > 32 rows (a per-GPU batch of 64, 512). */
int pv_linear_wgrad_dx_adam_tall_bf16(const uint16_t* x, const float* dy, float* param, float* exp_avg, float* exp_avg_sq,
                                      uint16_t* bf16_shadow, uint16_t* dx, int32_t m, int32_t n, int64_t k, double lr,
                                      double beta1, double beta2, double eps, int32_t step, float grad_scale,
                                      int32_t gate_dx_by_x, int32_t moments_tiled, void* workspace, size_t workspace_bytes,
                                      void* stream);
/* bytes of that call's workspace (dy as matrix-core operand fragments, built once per call): 40 KB per 32 rows */
int pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes(int32_t m, size_t* bytes);

int pv_adam_step_bf16grad(float* param, const uint16_t* grad_bf16, float* exp_avg, float* exp_avg_sq,
                          uint16_t* bf16_shadow, size_t n, double lr, double beta1, double beta2, double eps,
                          int32_t step, float grad_scale, void* stream);

/* replaces: the bias + ReLU epilogue of `F.relu(self.fc1(out))` (predict_pv_yield/models/conv3d/model.py:125) when fc1's
 * product was summed OUTSIDE the kernel -- the K-sharded data-parallel mode, where every rank multiplies its column shard of
 * the weight and the partial products are added in rank order -- and the 1 / world scaling of an output gradient
 * (the averaging of DDP's all-reduce, experiments/003_...py:292-293):  y[i][j] = act(alpha x[i][j] + bias[j]),
 * bias may be NULL, relu != 0 applies max(., 0) (a NaN stays a NaN); y may alias x. */
int pv_scale_bias_relu_f32(const float* x, const float* bias, float* y, int32_t m, int32_t n, float alpha, int32_t relu,
                           void* stream);

/* dst[i1][i0][:] = src[i0][i1][:] for contiguous segments of seg_bytes (a multiple of 16; 16-byte aligned buffers, not in place):
 * the chunk-major staging copies around the K-sharded fc1's two all-to-alls, [B][W][K / W] <-> [W][B][K / W] -- what
 * `.transpose(0, 1).contiguous()` does to the tensors DDP never had to move (experiments/003_...py:292-293 exchanges gradients;
 * this mode exchanges activations). */
int pv_swap01_segments(const void* src, void* dst, int64_t n0, int64_t n1, int64_t seg_bytes, void* stream);

/* dst[i] = bf16(src[i]) (round to nearest even): first fill of a parameter's bf16 shadow; afterwards
 * pv_adam_step_f32 keeps the shadow current. */
int pv_cast_f32_to_bf16(const float* src, uint16_t* dst, size_t n, void* stream);

/* replaces: the backward of F.relu (predict_pv_yield/models/conv3d/model.py:117-120 under loss.backward()) as its own pass:
 * out[i] = y[i] > 0 ? dy[i] : 0 (f32, n % 4 == 0, 16-byte aligned; out may alias dy).  The exact-f32 model uses it on fc1's
 * input gradient so that the last conv layer's dgrad / wgrad receive an already gated gradient (their matrix-core kernels
 * stage operands global -> LDS directly and cannot gate on the way). */
int pv_relu_gate_f32(const float* dy, const float* y, float* out, size_t n, void* stream);
/* The same, and state[0] receives the bits of the largest |out| (the scale of pv_pack_split2_...'s split of `out`). */
int pv_relu_gate_max_f32(const float* dy, const float* y, float* out, size_t n, float* state, void* stream);

/* replaces: nn.Embedding(num_embeddings=940, embedding_dim=16)(id) and its backward
 * (predict_pv_yield/models/conv3d/model_sat_nwp.py:149-151, 251-260).  ids: device int64[n_ids]; out-of-range ids give a
 * zero row.  bwd overwrites dtable[n_rows, dim] with the per-row sums of dout (fixed order, no atomics). */
int pv_embedding_fwd_f32(const float* table, const int64_t* ids, float* out, int32_t n_ids, int32_t dim,
                         int32_t n_rows, void* stream);
int pv_embedding_bwd_f32(const float* dout, const int64_t* ids, float* dtable, int32_t n_ids, int32_t dim,
                         int32_t n_rows, void* stream);

/* ---- Perceiver path (perceiver_pytorch.Perceiver as instantiated by predict_pv_yield/models/perceiver/perceiver.py:70-80;
 * third-party, unpinned: requirements.txt:12) -------------------------------------------------------------------------- */
/* Batched, arbitrarily strided f32 GEMM on the f32 matrix cores: C[z](m,n) = sum_k A[z](m,k) B[z](k,n) (+bias[n]) (ReLU),
 * A[z](m,k) at a + z1*a_bs1 + z2*a_bs2 + m*a_rs + k*a_cs (element strides), B likewise, C row-major with ldc; z = z1*batch2
 * + z2.  k_splits > 1: the k range is cut into k_splits chunks and partial product s goes to c + s*c_ss (the caller sums
 * them, pv_sum_slabs_f32) -- the form used for weight gradients, whose k is the (huge) row count.
 * replaces: nn.Linear of Attention.to_q/to_kv/to_out and FeedForward, einsum('b i d, b j d -> b i j') and
 * einsum('b i j, b j d -> b i d') of Attention.forward, and the autograd products of all of them. */
typedef struct pv_gemm_desc {
  int32_t m, n, k;
  int64_t a_rs, a_cs, b_rs, b_cs, ldc;
  int32_t batch1, batch2;
  int64_t a_bs1, a_bs2, b_bs1, b_bs2, c_bs1, c_bs2;
  int32_t k_splits;
  int64_t c_ss;
} pv_gemm_desc;
int pv_gemm_f32(const float* a, const float* b, const float* bias, float* c, const pv_gemm_desc* d, int relu, void* stream);
/* C = A B + bias + residual (row-major [m][ldr], 2-D products only): the `fn(x) + x` of the Perceiver's PreNorm blocks
 * (perceiver_pytorch: `x = cross_attn(x, ...) + x`, `x = cross_ff(x) + x`, ...) folded into the epilogue of fn's last Linear. */
int pv_gemm_res_f32(const float* a, const float* b, const float* bias, const float* residual, int64_t ldr, float* c,
                    const pv_gemm_desc* d, int relu, void* stream);
/* C (bf16, row-major, ldc elements per row) = A B + bias for ONE tall row-major A with K <= 64: to_kv of a cross-attention
 * (perceiver_pytorch Attention.to_kv, models/perceiver/perceiver.py:70-80) writing the K / V the bf16-operand attention kernels
 * read (pv_attention_*_bf16kv).  f32-accurate products, one rounding in the store. */
int pv_gemm_rows_bf16out_f32(const float* a, const float* b, const float* bias, uint16_t* c_bf16, const pv_gemm_desc* d,
                             int32_t flags, void* stream);
/* pv_gemm_res_f32 with flags.  PV_GEMM_BF16_OPERANDS: both operands rounded ONCE to bf16 (nearest even), one matrix-core
 * product, f32 accumulation -- what torch.autocast makes of nn.Linear / matmul under Lightning precision=16
 * (experiments/003_perceiver_processes_single_sat_image_then_rnn.py:40,288-294); default: the f32-accurate three-term form. */
#define PV_GEMM_BF16_OPERANDS 1
/* with PV_GEMM_BF16_OPERANDS: `a` points at a bf16 matrix (strides in elements) -- e.g. the dK / dV a bf16-operand attention
 * backward stored as bf16 (pv_attention_bwd_bf16kv16): the values the product would round A to anyway, half the bytes */
#define PV_GEMM_A_IS_BF16 2
int pv_gemm_ex_f32(const float* a, const float* b, const float* bias, const float* residual, int64_t ldr, float* c,
                   const pv_gemm_desc* d, int relu, int32_t flags, void* stream);
int pv_sum_slabs_f32(const float* slabs, float* out, int64_t n, int32_t n_slabs, void* stream);
/* accumulate != 0: out += the sum -- a weight that several layers share (weight_tie_layers=True,
 * predict_pv_yield/models/perceiver/perceiver.py:70-80) collects its gradient contributions in place, in arrival order,
 * instead of through one autograd add per use. */
int pv_sum_slabs_acc_f32(const float* slabs, float* out, int64_t n, int32_t n_slabs, int32_t accumulate, void* stream);
/* out[c] (+)= sum over rows of x[r][c] (the bias gradient of nn.Linear): chunk partials in `workspace`
 * (pv_colsum_workspace_floats floats), summed in chunk order. */
size_t pv_colsum_workspace_floats(int64_t rows, int32_t cols);
int pv_colsum_f32(const float* x, float* out, int64_t rows, int32_t cols, float* workspace, int32_t accumulate, void* stream);
/* Fused attention out = softmax(scale q k^T) v with an online softmax on the f32 matrix cores (the scores never reach
 * memory).  Element (b, h, i, d) of q / out at q + b*q_batch_stride + i*q_row_stride + h*64 + d; of k / v (separate base
 * pointers, e.g. the two halves of a fused kv projection) at k + b*k_batch_stride + j*k_row_stride + h*64 + d.
 * lse[b, h, i] = log sum_j exp(scale q_i.k_j) is saved for the backward.  head_dim must be 64.
 * replaces: the two einsums and the softmax of perceiver_pytorch's Attention.forward (see pv_gemm_f32). */
typedef struct pv_attention_desc {
  int32_t batch, heads, n_q, n_k, head_dim;
  int64_t q_batch_stride, q_row_stride, k_batch_stride, k_row_stride;
  float scale;
} pv_attention_desc;
int pv_attention_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, const pv_attention_desc* d,
                         void* stream);
/* Backward of pv_attention_fwd_f32 for n_q <= 128: the probabilities are recomputed tile by tile from lse; dq has the
 * layout of q, dk / dv the layout of k / v (separate base pointers); delta_ws: scratch of
 * pv_attention_bwd_workspace_floats(d) floats (row terms + per-key-range dQ partials; q must be dense per batch:
 * q_batch_stride = n_q * q_row_stride). */
size_t pv_attention_bwd_workspace_floats(const pv_attention_desc* d);
int pv_attention_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse,
                         float* delta_ws, float* dq, float* dk, float* dv, const pv_attention_desc* d, void* stream);

/* The same two entry points with bf16 OPERANDS on the bf16 matrix cores (f32 tensors in memory, f32 accumulation, f32 online
 * softmax): q, k, v, dO are rounded to bf16 on their way into the products, the probabilities and dS when they become
 * operands.  replaces: the same einsums as run under Lightning `precision=16`
 * (experiments/003_perceiver_processes_single_sat_image_then_rnn.py:40,288-294).  Same descriptor, workspace and layouts. */
size_t pv_attention_fwd_workspace_floats(const pv_attention_desc* d);   /* key-split partials of the bf16 forward; 0 = none */
int pv_attention_fwd_bf16(const float* q, const float* k, const float* v, float* o, float* lse, const pv_attention_desc* d,
                          float* workspace, void* stream);   /* workspace NULL: no key split */
int pv_attention_bwd_bf16(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse,
                          float* delta_ws, float* dq, float* dk, float* dv, const pv_attention_desc* d, int32_t accumulate_dkv,
                          void* stream);
/* The same two with K and V stored as bf16 (the values the kernels above round to on their way into LDS): half the bytes of
 * these HBM-bound launches (128 queries make 64 flop per byte of f32 K / V).  k / v: bf16 [batch, n_k, ...] with the
 * descriptor's k_*_stride counted in bf16 elements (multiples of 8); dk / dv stay f32 with the same element strides. */
int pv_attention_fwd_bf16kv(const float* q, const uint16_t* k, const uint16_t* v, float* o, float* lse, const pv_attention_desc* d,
                            float* workspace, void* stream);
int pv_attention_bwd_bf16kv(const float* q, const uint16_t* k, const uint16_t* v, const float* o, const float* dout, const float* lse,
                            float* delta_ws, float* dq, float* dk, float* dv, const pv_attention_desc* d, int32_t accumulate_dkv,
                            void* stream);   /* accumulate_dkv != 0: dk, dv += (keys / values shared by weight-tied layers) */
/* ... and dK / dV stored as bf16 too (rounded to nearest even; element strides as k / v): for a consumer that rounds them to
 * bf16 anyway -- pv_gemm_ex_f32 with PV_GEMM_BF16_OPERANDS | PV_GEMM_A_IS_BF16, the two backward products of to_kv. */
int pv_attention_bwd_bf16kv16(const float* q, const uint16_t* k, const uint16_t* v, const float* o, const float* dout,
                              const float* lse, float* delta_ws, float* dq, uint16_t* dk, uint16_t* dv, const pv_attention_desc* d,
                              void* stream);   /* always overwrites dk / dv (a context one layer consumes) */

/* F.layer_norm over the last dimension d <= 256 (PreNorm.norm / norm_context, to_logits' LayerNorm); mean / rstd [rows]
 * are saved for the backward, which also returns dw = sum dy*xhat and db = sum dy (dx may be NULL). */
int pv_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int64_t rows,
                         int32_t d, float eps, void* stream);
int pv_layernorm_bwd_workspace_bytes(int64_t rows, int32_t d, size_t* bytes);
int pv_layernorm_bwd_f32(const float* x, const float* w, const float* dy, const float* mean, const float* rstd, float* dx,
                         float* dw, float* db, int64_t rows, int32_t d, void* ws, size_t ws_bytes, int32_t accumulate,
                         const float* dx_add, void* stream);
/* accumulate != 0: dw, db += (see pv_sum_slabs_acc_f32); db == dw + d: one launch; dx_add (may be NULL): dx = ... + dx_add,
 * the gradient that reaches x past the block (the residual branch of `fn(norm(x)) + x`) folded into this kernel's store */
/* The same two sums for the LayerNorm of a cross-attention's context (PreNorm.norm_context -> Attention.to_kv,
 * perceiver_pytorch as used by predict_pv_yield/models/perceiver/perceiver.py:119-131) straight from the gradient of the
 * projection, when x itself needs no gradient: dy = dkv16 [rows, kdim] (bf16, as pv_attention_bwd_bf16kv16 stores it) times
 * the Linear weight w_kv [kdim, d] (rounded to bf16: the operands of pv_gemm_ex_f32 with PV_GEMM_BF16_OPERANDS) is formed per
 * 32-row block on the matrix cores and folded into dw / db at once -- dy [rows, d] is never written.  d <= 64, kdim 64 | 128. */
int pv_layernorm_bwd_params_from_proj_workspace_bytes(int64_t rows, int32_t d, size_t* bytes);
int pv_layernorm_bwd_params_from_proj_bf16(const uint16_t* dkv16, const float* w_kv, const float* x, const float* mean,
                                           const float* rstd, float* dw, float* db, int64_t rows, int32_t d, int32_t kdim,
                                           void* ws, size_t ws_bytes, int32_t accumulate, void* stream);
/* The forward of the chain in one pass: kv16 [rows, kdim] (bf16) = LayerNorm(x) w_kv^T with the normalised context formed
 * in the registers of the rows-form product (never written) and rounded to bf16 as pv_gemm_rows_bf16out_f32 with PV_GEMM_BF16_OPERANDS rounds
 * its operands; mean / rstd [rows] as pv_layernorm_fwd_f32 returns them.  d even, <= 48; kdim % 64 == 0; w_kv [kdim, d]. */
int pv_context_fwd_bf16(const float* x, const float* x2, int32_t d1, int64_t period, const float* ln_w, const float* ln_b,
                        const float* w_kv, uint16_t* kv16, float* mean, float* rstd, int64_t rows, int32_t d, int32_t kdim,
                        float eps, void* stream);
/* x2 != NULL (here and in pv_context_bwd_bf16): row r of the context is [ x[r][0 .. d1) | x2[r % period][0 .. d - d1) ] -- the image
 * channels followed by the Fourier features of the pixel position (perceiver_pytorch's fourier_encode + torch.cat), which every
 * image shares: the concatenated [rows, d] tensor is never written.  d1 even.  x2 == NULL: x is [rows, d]. */
/* ... and the whole backward of that chain in ONE pass over dkv16 and x: also dw_kv [kdim, d] = dkv16^T ctx with
 * ctx = LayerNorm(x) rounded to bf16 (formed per 32-row block in LDS: the operand the weight-gradient GEMM would read from
 * memory).  kdim = 128.  accumulate_kv / accumulate_ln != 0: += into dw_kv / (dln_w, dln_b). */
int pv_context_bwd_workspace_bytes(int64_t rows, int32_t d, size_t* bytes);
int pv_context_bwd_bf16(const uint16_t* dkv16, const float* w_kv, const float* x, const float* x2, int32_t d1, int64_t period,
                        const float* mean, const float* rstd, const float* ln_w, const float* ln_b, float* dw_kv, float* dln_w,
                        float* dln_b, int64_t rows, int32_t d, int32_t kdim, void* ws, size_t ws_bytes, int32_t accumulate_kv,
                        int32_t accumulate_ln, void* stream);
/* y = softmax(scale * x) over rows of `len` (sim.softmax(dim=-1) with the dim_head**-0.5 scale folded in; x == y allowed);
 * bwd: dx = scale * p * (dp - sum(dp * p)) (dx == dp allowed). */
int pv_softmax_fwd_f32(const float* x, float* y, int64_t rows, int32_t len, float scale, void* stream);
int pv_softmax_bwd_f32(const float* p, const float* dp, float* dx, int64_t rows, int32_t len, float scale, void* stream);
/* GEGLU of FeedForward: x [rows, 2h] -> y [rows, h] = x[:, :h] * gelu_erf(x[:, h:]) and its backward. */
int pv_geglu_fwd_f32(const float* x, float* y, int64_t rows, int32_t h, void* stream);
int pv_geglu_bwd_f32(const float* x, const float* dy, float* dx, int64_t rows, int32_t h, void* stream);
/* Reduce('b n d -> b d', 'mean') of to_logits and its backward. */
int pv_mean_axis1_fwd_f32(const float* x, float* y, int32_t b, int32_t n, int32_t d, void* stream);
int pv_mean_axis1_bwd_f32(const float* dy, float* dx, int32_t b, int32_t n, int32_t d, void* stream);

/* Sequential part of nn.GRU (batch_first, gate order r,z,n; predict_pv_yield/models/perceiver/perceiver.py:94-109,193-196)
 * for one layer: gi[B,T,3H] = x W_ih^T + b_ih (a GEMM), h0[B,H] or NULL (zeros) -> out[B,T,H]; saved[B,T,4H] keeps
 * (r, z, n, W_hn h + b_hn) for the backward.  bwd: dout[B,T,H] and/or dh_last[B,H] (either may be NULL) -> dgi[B,T,3H],
 * dh0[B,H] (may be NULL), dw_hh[3H,H], db_hh[3H]; ws >= B*(3H*H + 3H)*4 bytes.  H <= 64. */
int pv_gru_seq_fwd_f32(const float* gi, const float* h0, const float* w_hh, const float* b_hh, float* out, float* saved,
                       int32_t batch, int32_t t_len, int32_t hidden, void* stream);
int pv_gru_seq_bwd_f32(const float* dout, const float* dh_last, const float* h0, const float* out, const float* saved,
                       const float* w_hh, float* dgi, float* dh0, float* dw_hh, float* db_hh, int32_t batch, int32_t t_len,
                       int32_t hidden, void* ws, size_t ws_bytes, void* stream);

/* ---- loss + optimiser ----------------------------------------------------- */
/* replaces: F.mse_loss / (y_hat−y).abs().mean() and WeightedLosses.get_mse_exp/get_mae_exp
 * (predict_pv_yield/models/base_model.py:98-103).  out: device f32[4] = {mse, nmae, mse_exp, mae_exp};
 * grad (may be NULL): d nmae / d y_hat = sign(y_hat−y)/(m*n) * grad_scale.
 * per_horizon (may be NULL): device f32[2n] = {mse per forecast step [n], mae per forecast step [n]}, the batch-axis
 * means of nowcasting_utils' mse_each_forecast_horizon / mae_each_forecast_horizon (base_model.py:121-141),
 * produced by the same launch.
 * y is read with row stride y_row_stride (elements) so the slice y[:, -forecast_len:, 0]
 * (base_model.py:95) needs no copy: element (i,j) at y[i*y_row_stride + j*y_col_stride]. */
int pv_forecast_losses_f32(const float* y_hat, const float* y, int64_t y_row_stride,
                           int64_t y_col_stride, int32_t m, int32_t n, float grad_scale,
                           float* out4, float* grad, float* per_horizon, void* stream);

/* replaces: torch.optim.Adam(lr=5e-4).step()  (base_model.py:255-257); one parameter tensor.
 * Exactly torch's single-tensor Adam order of operations in f32 (no weight decay, no amsgrad);
 * lr/betas/eps are doubles because torch derives step_size and the bias corrections in Python floats.
 * The gradient is read as grad * grad_scale (1/world_size after a summing all-reduce).
 * bf16_shadow (may be NULL) receives round-to-nearest-even bf16 of the updated parameter. */
int pv_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     uint16_t* bf16_shadow, size_t n, double lr, double beta1, double beta2,
                     double eps, int32_t step, float grad_scale, void* stream);

/* replaces: skimage.metrics.structural_similarity(ground_truth_image, remapped_image) with every option at its default
 * (notebooks/optical_flow_1.ipynb:10235-10243 [cell 35], the `update` of cell 31 and compute_opt_flow_and_score of cell 38:
 * the reference's only quality score for its optical-flow forecasts, and the objective of its Farneback parameter search).
 * n_pairs image pairs [h][w], pair i at im1 + i * stride1 / im2 + i * stride2 (elements); out: device f64[n_pairs] = the mean of
 * the SSIM map (7 x 7 uniform window, float64, sample covariance, K1 = 0.01, K2 = 0.03, a border of 3 cropped).
 * data_range: 255 for uint8 images; scikit-image's default for float images is 2 (the dtype's range [-1, 1]). */
int pv_ssim_mean_u8(const uint8_t* im1, int64_t stride1, const uint8_t* im2, int64_t stride2, int64_t n_pairs, int32_t h,
                    int32_t w, double data_range, double* out, void* stream);
int pv_ssim_mean_f32(const float* im1, int64_t stride1, const float* im2, int64_t stride2, int64_t n_pairs, int32_t h,
                     int32_t w, double data_range, double* out, void* stream);

/* ---- tracing: opt-in per-stage device timing ---------------------------------------------------------------------
 * The reference times its pipeline stages with wall-clock `%%time` cells (notebooks/optical_flow_1.ipynb:269,
 * 13_...ipynb:1161); here the multi-kernel entry points (pv_farneback_batch_u8, pv_prepare_stacks_*,
 * pv_flow_weighted_mean_f32, pv_remap_bilinear_*) record one HIP event per stage boundary on the caller's stream while
 * a timing session is armed.  pv_stage_timing_end waits for the last event and returns, per distinct stage label
 * (static strings, first-seen order), the summed milliseconds and the number of occurrences.  Process-global, one
 * session / one stream at a time; nothing is recorded on a stream that is being captured into a graph. */
int pv_stage_timing_begin(void);
int pv_stage_timing_end(const char** names, float* ms, int32_t* counts, int32_t capacity, int32_t* n_out);

/* ---- device calibration (bench.py `device_calibration`) -------------------------------------------------------------
 * The reference quotes wall-clock timings per box (notebooks/optical_flow_1.ipynb:269, experiments/2021-08/2021-08-31/
 * experiments.txt:5-6); here every roofline fraction is also reported against what THIS device sustains, measured in-process
 * right before the timed steps: a plain 16-byte-per-lane copy of n floats (bytes moved = 8 n), and `workgroups` x 4 waves x
 * `iters` x 8 back-to-back v_mfma_f32_16x16x32_bf16 (16 384 flop each) on register operands.  sink: device f32[workgroups],
 * never written in practice.  Not part of the product path. */
int pv_calibrate_copy_f32(const float* src, float* dst, size_t n, void* stream);
int pv_calibrate_mfma_bf16(float* sink, int32_t workgroups, int32_t iters, void* stream);

/* (measurement) One wave that writes n_samples pairs (shader-cycle counter, 100 MHz counter) into samples[2 * n_samples],
 * sleeping ~sleep_units x 64 cycles between two pairs: launched on a side stream before the kernels of interest, the ratio of the
 * two counters' increments is the engine clock those kernels ran at. */
int pv_clock_watch(unsigned long long* samples, int32_t n_samples, int32_t sleep_units, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PV_YIELD_HIP_H */
