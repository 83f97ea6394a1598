"""Loader for the C-ABI library libpvyield_hip.so (include/pv_yield_hip.h).

The product path has NO CPU fallback: if the library is missing, or a call returns a negative status,
a RuntimeError is raised.  Tensors cross the boundary as raw device pointers + sizes; the stream is
torch's current HIP stream.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# PV_YIELD_LIB: tools/diag_stamps.py points this at lib/libpvyield_diag.so (the same sources with s_memtime stamps; `make diag`)
LIB_PATH = os.environ.get("PV_YIELD_LIB") or os.path.join(_PKG, "lib", "libpvyield_hip.so")
CSRC_DIR = os.path.join(_PKG, "csrc")

PV_BORDER_CONSTANT = 0
PV_BORDER_REPLICATE = 1
PV_U8_ROUND_DIV4 = 0
PV_U8_TRUNC_SCALE = 1
PV_OPTFLOW_FARNEBACK_GAUSSIAN = 256

c_i32, c_i64, c_f32, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double
c_vp, c_sz, c_int, c_u8 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint8


class FarnebackParams(ctypes.Structure):
    """struct pv_farneback_params: the positional arguments of cv.calcOpticalFlowFarneback."""
    _fields_ = [("pyr_scale", c_f64), ("levels", c_i32), ("winsize", c_i32), ("iterations", c_i32),
                ("poly_n", c_i32), ("poly_sigma", c_f64), ("flags", c_i32)]


class Conv3dDims(ctypes.Structure):
    """struct pv_conv3d_dims."""
    _fields_ = [("batch", c_i32), ("c_in", c_i32), ("c_out", c_i32), ("t_in", c_i32), ("h_in", c_i32),
                ("w_in", c_i32), ("pad_t", c_i32), ("pad_h", c_i32), ("pad_w", c_i32)]

    def out_shape(self):
        return (self.t_in + 2 * self.pad_t - 2, self.h_in + 2 * self.pad_h - 2, self.w_in + 2 * self.pad_w - 2)


class AdamTensor(ctypes.Structure):
    """struct pv_adam_tensor."""
    _fields_ = [("param", c_vp), ("grad", c_vp), ("exp_avg", c_vp), ("exp_avg_sq", c_vp), ("bf16_shadow", c_vp),
                ("n", ctypes.c_uint64)]


class PackJob(ctypes.Structure):
    """struct pv_pack_job."""
    _fields_ = [("w", c_vp), ("wp", c_vp), ("c_out", c_i32), ("c_in", c_i32), ("transpose_flip", c_i32)]


class GemmDesc(ctypes.Structure):
    """struct pv_gemm_desc."""
    _fields_ = [("m", c_i32), ("n", c_i32), ("k", c_i32), ("a_rs", c_i64), ("a_cs", c_i64), ("b_rs", c_i64), ("b_cs", c_i64),
                ("ldc", c_i64), ("batch1", c_i32), ("batch2", c_i32), ("a_bs1", c_i64), ("a_bs2", c_i64), ("b_bs1", c_i64),
                ("b_bs2", c_i64), ("c_bs1", c_i64), ("c_bs2", c_i64), ("k_splits", c_i32), ("c_ss", c_i64)]


class AttentionDesc(ctypes.Structure):
    """struct pv_attention_desc."""
    _fields_ = [("batch", c_i32), ("heads", c_i32), ("n_q", c_i32), ("n_k", c_i32), ("head_dim", c_i32),
                ("q_batch_stride", c_i64), ("q_row_stride", c_i64), ("k_batch_stride", c_i64), ("k_row_stride", c_i64),
                ("scale", c_f32)]


PV_ADAM_MAX_TENSORS = 32
PV_PACK_MAX_JOBS = 16


class Conv3dGeom(ctypes.Structure):
    """struct pv_conv3d_geom: kernel extents 1..3, stride, padding."""
    _fields_ = [(n, c_i32) for n in ("batch", "c_in", "c_out", "t_in", "h_in", "w_in", "k_t", "k_h", "k_w",
                                     "stride_t", "stride_h", "stride_w", "pad_t", "pad_h", "pad_w")]

    def out_shape(self):
        return ((self.t_in + 2 * self.pad_t - self.k_t) // self.stride_t + 1,
                (self.h_in + 2 * self.pad_h - self.k_h) // self.stride_h + 1,
                (self.w_in + 2 * self.pad_w - self.k_w) // self.stride_w + 1)


# name -> argtypes; every symbol declared in include/pv_yield_hip.h (tests/test_abi.py checks both ways)
_PFB = ctypes.POINTER(FarnebackParams)
_PCD = ctypes.POINTER(Conv3dDims)
_PCG = ctypes.POINTER(Conv3dGeom)
_PI32 = ctypes.POINTER(c_i32)
SIGNATURES = {
    "pv_abi_version": [],
    "pv_last_error": [],
    "pv_u8_from_10bit_i16": [c_vp, c_vp, c_sz, c_int, c_vp, c_vp],
    "pv_u8_from_10bit_f32": [c_vp, c_vp, c_sz, c_int, c_vp, c_vp],
    "pv_farneback_workspace_bytes": [c_i64, c_i32, c_i32, _PFB, ctypes.POINTER(c_sz)],
    "pv_farneback_batch_u8": [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i32, c_i32, _PFB, c_vp, c_sz, c_vp],
    "pv_flow_weighted_mean_f32": [c_vp, ctypes.POINTER(c_f64), c_vp, c_i64, c_i32, c_i64, c_vp],
    "pv_remap_bilinear_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_i32, c_i32,
                              c_int, c_f32, c_vp],
    "pv_remap_bilinear_u8": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_i32, c_i32,
                             c_int, c_u8, c_vp],
    "pv_prepare_stacks_i16": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_i32, c_int, c_vp, c_vp, c_vp, c_vp],
    "pv_prepare_stacks_f32": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_i32, c_int, c_vp, c_vp, c_vp, c_vp],
    "pv_normalise_i16": [c_vp, c_vp, c_sz, c_i64, c_i32, c_vp, c_vp, c_vp],
    "pv_normalise_f32": [c_vp, c_vp, c_sz, c_i64, c_i32, c_vp, c_vp, c_vp],
    "pv_bf16_cpad": [c_i32],
    "pv_pack_ncdhw_f32_to_ndhwc_bf16": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "pv_pack_split2_ncdhw_f32_to_ndhwc_f16": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "pv_unpack_ndhwc_bf16_to_ncdhw_f32": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "pv_repack_gate_ncdhw_to_ndhwc_bf16": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "pv_conv3d_packed_weight_elems": [c_i32],
    "pv_conv3d_pack_weight_bf16": [c_vp, c_vp, c_i32, c_i32, c_int, c_vp],
    "pv_relu_mask_dims": [c_i32, c_i32, c_vp, c_vp],
    "pv_ssim_mean_u8": [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_f64, c_vp, c_vp],
    "pv_ssim_mean_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_f64, c_vp, c_vp],
    "pv_stage_timing_begin": [],
    "pv_stage_timing_end": [c_vp, c_vp, c_vp, c_i32, c_vp],
    "pv_calibrate_copy_f32": [c_vp, c_vp, c_sz, c_vp],
    "pv_calibrate_mfma_bf16": [c_vp, c_i32, c_i32, c_vp],
    "pv_conv3d_fwd_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _PCD, c_int, c_int, c_vp],
    "pv_conv3d_fwd_bf16_f32in": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, _PCD, c_int, c_vp],
    "pv_conv3d_bwd_weight_bf16_workspace_bytes": [_PCD, ctypes.POINTER(c_sz)],
    "pv_conv3d_bwd_weight_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, _PCD, c_vp, c_sz, c_vp],
    "pv_conv3d_bwd_weight_f16": [c_vp, c_vp, c_vp, c_vp, _PCD, c_vp, c_sz, c_vp],
    "pv_conv3d_split2_weight_elems": [],
    "pv_conv3d_pack_weight_split2_f16": [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp],
    "pv_conv3d_fwd_f16_f32out": [c_vp, c_vp, c_vp, c_i32, _PCD, c_vp],
    "pv_conv3d_fwd_f16_f32out_covers": [_PCD],
    "pv_sum3_ndhwc_to_ncdhw_f32": [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp],
    "pv_linear_f32_skinny_covers": [c_i32, c_i32, c_i64],
    "pv_linear_fwd_f32_skinny_workspace_bytes": [c_i32],
    "pv_linear_fwd_f32_skinny": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_sz, c_vp],
    "pv_linear_dx_f32_skinny": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp],
    "pv_linear_workspace_bytes": [c_i32, c_i32, c_i64, ctypes.POINTER(c_sz)],
    "pv_linear_fwd_f32": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_int, c_vp, c_sz, c_vp],
    "pv_linear_bwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp],
    "pv_linear_bf16_workspace_bytes": [c_i32, c_i32, c_i64, ctypes.POINTER(c_sz)],
    "pv_linear_fwd_bf16": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_int, c_vp, c_sz, c_vp],
    "pv_linear_bwd_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp],
    "pv_linear_wgrad_adam_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64,
                                  c_i32, c_vp],
    "pv_linear_wgrad_adam_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64, c_i32, c_vp],
    "pv_linear_wgrad_dx_adam_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_f64, c_f64,
                                     c_f64, c_f64, c_i32, c_i32, c_i32, c_vp],
    "pv_linear_wgrad_bf16out": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp],
    "pv_adam_step_bf16grad": [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_f64, c_f64, c_f64, c_f64, c_i32, c_f32, c_vp],
    "pv_conv3d_general_out_extent": [_PCG, _PI32, _PI32, _PI32],
    "pv_conv3d_general_fwd_f32": [c_vp, c_vp, c_vp, c_vp, _PCG, c_int, c_vp],
    "pv_conv3d_general_bwd_data_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, _PCG, c_vp],
    "pv_conv3d_general_bwd_weight_workspace_bytes": [_PCG, ctypes.POINTER(c_sz)],
    "pv_conv3d_general_bwd_weight_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, _PCG, c_vp, c_sz, c_vp],
    "pv_maxpool3d_fwd_f32": [c_vp, c_vp, c_vp, _PCG, c_vp],
    "pv_maxpool3d_bwd_f32": [c_vp, c_vp, c_vp, _PCG, c_vp],
    "pv_mse_loss_f32": [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp],
    "pv_adam_step_multi_f32": [ctypes.POINTER(AdamTensor), c_i32, c_f64, c_f64, c_f64, c_f64, c_i32, c_f32, c_vp],
    "pv_adam_scalars_advance": [c_vp, c_vp, c_f64, c_f64, c_f64, c_f64, c_vp],
    "pv_adam_step_multi_dev_f32": [ctypes.POINTER(AdamTensor), c_i32, c_vp, c_f32, c_vp],
    "pv_linear_wgrad_dx_adam_dev_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_i32, c_i32,
                                         c_vp],
    "pv_linear_wgrad_dx_adam_tall_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64,
                                          c_i32, c_f32, c_i32, c_i32, c_vp, c_sz, c_vp],
    "pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes": [c_i32, ctypes.POINTER(c_sz)],
    "pv_swap01_segments": [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp],
    "pv_clock_watch": [c_vp, c_i32, c_i32, c_vp],
    "pv_conv3d_pack_weights_multi_bf16": [ctypes.POINTER(PackJob), c_i32, c_vp],
    "pv_gemm_f32": [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(GemmDesc), c_int, c_vp],
    "pv_gemm_res_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(GemmDesc), c_int, c_vp],
    "pv_gemm_rows_bf16out_f32": [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(GemmDesc), c_i32, c_vp],
    "pv_gemm_ex_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(GemmDesc), c_int, c_i32, c_vp],
    "pv_sum_slabs_f32": [c_vp, c_vp, c_i64, c_i32, c_vp],
    "pv_sum_slabs_acc_f32": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp],
    "pv_colsum_f32": [c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp],
    "pv_colsum_workspace_floats": [c_i64, c_i32],
    "pv_attention_fwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_vp],
    "pv_attention_bwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_vp],
    "pv_attention_fwd_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_vp, c_vp],
    "pv_attention_fwd_bf16kv": [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_vp, c_vp],
    "pv_attention_fwd_workspace_floats": [ctypes.POINTER(AttentionDesc)],
    "pv_attention_bwd_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_i32, c_vp],
    "pv_attention_bwd_bf16kv": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_i32, c_vp],
    "pv_attention_bwd_bf16kv16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(AttentionDesc), c_vp],
    "pv_attention_bwd_workspace_floats": [ctypes.POINTER(AttentionDesc)],
    "pv_layernorm_fwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp],
    "pv_layernorm_bwd_workspace_bytes": [c_i64, c_i32, ctypes.POINTER(c_sz)],
    "pv_layernorm_bwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_sz, c_i32, c_vp, c_vp],
    "pv_context_fwd_bf16": [c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, ctypes.c_float, c_vp],
    "pv_context_bwd_workspace_bytes": [c_i64, c_i32, ctypes.POINTER(c_sz)],
    "pv_context_bwd_bf16": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32,
                            c_vp, c_sz, c_i32, c_i32, c_vp],
    "pv_layernorm_bwd_params_from_proj_workspace_bytes": [c_i64, c_i32, ctypes.POINTER(c_sz)],
    "pv_layernorm_bwd_params_from_proj_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_sz, c_i32, c_vp],
    "pv_softmax_fwd_f32": [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp],
    "pv_softmax_bwd_f32": [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp],
    "pv_geglu_fwd_f32": [c_vp, c_vp, c_i64, c_i32, c_vp],
    "pv_geglu_bwd_f32": [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp],
    "pv_mean_axis1_fwd_f32": [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "pv_mean_axis1_bwd_f32": [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "pv_gru_seq_fwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "pv_gru_seq_bwd_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp],
    "pv_embedding_fwd_f32": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "pv_embedding_bwd_f32": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "pv_cast_f32_to_bf16": [c_vp, c_vp, c_sz, c_vp],
    "pv_scale_bias_relu_f32": [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_vp],
    "pv_relu_gate_f32": [c_vp, c_vp, c_vp, c_sz, c_vp],
    "pv_relu_gate_max_f32": [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp],
    "pv_forecast_losses_f32": [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp],
    "pv_adam_step_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_f64, c_f64, c_f64, c_f64, c_i32, c_f32, c_vp],
}
_RESTYPES = {"pv_last_error": ctypes.c_char_p, "pv_conv3d_packed_weight_elems": c_sz, "pv_conv3d_split2_weight_elems": c_sz,
             "pv_linear_fwd_f32_skinny_workspace_bytes": c_sz,
             "pv_attention_bwd_workspace_floats": c_sz, "pv_colsum_workspace_floats": c_sz,
             "pv_attention_fwd_workspace_floats": c_sz}


def build_library(verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip into lib/libpvyield_hip.so (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC_DIR, "-j8"]
    if not verbose:
        args.append("-s")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def get_lib():
    """Load the library once; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C {CSRC_DIR}`). predict_pv_yield_amd has no CPU fallback.")
        # torch bundles its own libamdhip64: import it FIRST so this library binds to the same HIP runtime
        # (two runtimes in one process cannot share streams or device pointers)
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        if lib.pv_abi_version() != 1:
            raise RuntimeError("libpvyield_hip.so ABI version mismatch; rebuild")
        _lib = lib
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = get_lib().pv_last_error()
        raise RuntimeError(f"libpvyield_hip {what} failed with status {status}: {msg.decode() if msg else ''}")


_raw_stream = None


def current_stream_ptr():
    """torch's current HIP stream (of the current device) as a void* for the C ABI.  Through torch's raw-stream query when it
    exists: `torch.cuda.current_stream()` builds a Stream object behind four Python-level device look-ups, ~12 us -- a sixth of
    the host's time per launch in the launch-bound steps (the K-sharded rank at 32 samples per GPU, the Perceiver's ~3 000
    launches per step)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get_raw, get_dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get_raw is not None and get_dev is not None:
            _raw_stream = lambda: get_raw(get_dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return c_vp(_raw_stream())


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return c_vp(0) if t is None else c_vp(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("predict_pv_yield_amd: tensors must live on the MI355X (no CPU path is provided)")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("predict_pv_yield_amd: tensors crossing the C ABI must be contiguous")
