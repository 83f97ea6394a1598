"""Tensor-level wrappers over the C ABI (include/pv_yield_hip.h).

Every function here launches hand-written gfx950 kernels through libpvyield_hip.so on torch's current
HIP stream.  torch is used for device memory and streams only.  There is no CPU fallback: CPU tensors
or a missing library raise.
"""
import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import (Conv3dDims, Conv3dGeom, FarnebackParams, PV_BORDER_CONSTANT, PV_BORDER_REPLICATE,
                   PV_OPTFLOW_FARNEBACK_GAUSSIAN, check, current_stream_ptr, get_lib, ptr, require_cuda)

c_i32, c_i64, c_f32, c_f64, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_size_t


# ------------------------------------------------------------------------------------------------
# optical-flow advection
# ------------------------------------------------------------------------------------------------
def u8_from_10bit(x: torch.Tensor, mode: int = 0, return_flag: bool = False):
    """convert_10bpp_to_uint8 (notebooks/13_...ipynb:112-119); x int16 or float32, any shape."""
    require_cuda(x)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    flag = torch.zeros(1, dtype=torch.int32, device=x.device)
    lib = get_lib()
    if x.dtype == torch.int16:
        fn = lib.pv_u8_from_10bit_i16
    elif x.dtype == torch.float32:
        fn = lib.pv_u8_from_10bit_f32
    else:
        raise TypeError(f"u8_from_10bit: int16 or float32 expected, got {x.dtype}")
    check(fn(ptr(x), ptr(out), x.numel(), mode, ptr(flag), current_stream_ptr()), "pv_u8_from_10bit")
    return (out, flag) if return_flag else out


def make_farneback_params(pyr_scale=0.5, levels=2, winsize=40, iterations=3, poly_n=5, poly_sigma=0.7,
                          flags=PV_OPTFLOW_FARNEBACK_GAUSSIAN) -> FarnebackParams:
    return FarnebackParams(float(pyr_scale), int(levels), int(winsize), int(iterations), int(poly_n),
                           float(poly_sigma), int(flags))


_workspaces = {}
_pinned_workspaces = {}      # key -> number of live HIP graphs whose kernel arguments hold that buffer's address


def _workspace(key: str, nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffers owned by the host side (the C ABI never allocates).  A buffer that a live HIP graph
    replays into (pin_workspaces) is never replaced: a request that outgrows it raises instead of freeing memory the graph
    still writes."""
    k = (key, str(device))
    buf = _workspaces.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and _pinned_workspaces.get(k, 0) > 0:
            raise RuntimeError(
                f"workspace '{key}' ({buf.numel()} bytes) is held by a captured HIP graph (graphs.GraphedTrainStep) and a call "
                f"now needs {nbytes} bytes: replacing it would leave the graph replaying into freed memory.  Run the larger "
                f"call (e.g. a bigger validation batch, a second model) once BEFORE capturing, or release the graph "
                f"(GraphedTrainStep.close()).")
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _workspaces[k] = buf
    return buf


def pin_workspaces():
    """Marks every workspace that exists now as held by a captured graph; returns the token for unpin_workspaces."""
    keys = list(_workspaces)
    for k in keys:
        _pinned_workspaces[k] = _pinned_workspaces.get(k, 0) + 1
    return keys


def unpin_workspaces(keys) -> None:
    for k in keys or ():
        n = _pinned_workspaces.get(k, 0) - 1
        if n > 0:
            _pinned_workspaces[k] = n
        else:
            _pinned_workspaces.pop(k, None)


def farneback_stack(frames_u8: torch.Tensor, **kwargs) -> torch.Tensor:
    """compute_optical_flow (notebooks/13_...ipynb:175-240) for a batch of frame stacks.
    frames_u8: [..., T, H, W] uint8 -> flow [..., T-1, H, W, 2] float32, one Farnebäck field per
    consecutive pair.  One batched launch sequence replaces the reference's process pool."""
    require_cuda(frames_u8)
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() < 3:
        raise TypeError("farneback_stack: uint8 [..., T, H, W] expected")
    *lead, t, h, w = frames_u8.shape
    n_stacks = int(np.prod(lead)) if lead else 1
    if t < 2:
        raise ValueError("farneback_stack: need at least 2 frames")
    params = make_farneback_params(**kwargs)
    lib = get_lib()
    flows = torch.empty((*lead, t - 1, h, w, 2), dtype=torch.float32, device=frames_u8.device)
    if n_stacks == 0:
        return flows
    n_pairs = n_stacks * (t - 1)
    need = c_sz(0)
    check(lib.pv_farneback_workspace_bytes(n_pairs, h, w, ctypes.byref(params), ctypes.byref(need)),
          "pv_farneback_workspace_bytes")
    ws = _workspace("farneback", need.value, frames_u8.device)
    hw = h * w
    base = frames_u8.data_ptr()
    check(lib.pv_farneback_batch_u8(ctypes.c_void_p(base), ctypes.c_void_p(base + hw), hw, hw, t - 1, t * hw,
                                    ptr(flows), n_pairs, h, w, ctypes.byref(params), ptr(ws), ws.numel(),
                                    current_stream_ptr()), "pv_farneback_batch_u8")
    return flows


def farneback_pairs(prev_u8: torch.Tensor, next_u8: torch.Tensor, **kwargs) -> torch.Tensor:
    """cv.calcOpticalFlowFarneback batched: prev/next uint8 [N, H, W] -> flow f32 [N, H, W, 2]."""
    require_cuda(prev_u8, next_u8)
    if prev_u8.shape != next_u8.shape or prev_u8.dtype != torch.uint8 or next_u8.dtype != torch.uint8:
        raise TypeError("farneback_pairs: two uint8 tensors of equal shape expected")
    n, h, w = prev_u8.shape
    params = make_farneback_params(**kwargs)
    lib = get_lib()
    need = c_sz(0)
    check(lib.pv_farneback_workspace_bytes(n, h, w, ctypes.byref(params), ctypes.byref(need)),
          "pv_farneback_workspace_bytes")
    ws = _workspace("farneback", need.value, prev_u8.device)
    flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=prev_u8.device)
    check(lib.pv_farneback_batch_u8(ptr(prev_u8), ptr(next_u8), h * w, h * w, 0, 0, ptr(flow), n, h, w, ctypes.byref(params),
                                    ptr(ws), ws.numel(), current_stream_ptr()), "pv_farneback_batch_u8")
    return flow


def flow_weighted_mean(flows: torch.Tensor, weights: Optional[Sequence[float]] = None) -> torch.Tensor:
    """weighted_average (notebooks/optical_flow_1.ipynb:293-294): flows [G, N, ...] -> [G, ...]."""
    require_cuda(flows)
    if flows.dtype != torch.float32 or flows.dim() < 3:
        raise TypeError("flow_weighted_mean: float32 [G, N, ...] expected")
    g, n = flows.shape[:2]
    elems = flows[0, 0].numel()
    out = torch.empty((g, *flows.shape[2:]), dtype=torch.float32, device=flows.device)
    warr = None
    if weights is not None:
        warr = (c_f64 * n)(*[float(v) for v in weights])
    check(get_lib().pv_flow_weighted_mean_f32(ptr(flows), warr, ptr(out), g, n, elems, current_stream_ptr()),
          "pv_flow_weighted_mean_f32")
    return out


def remap_bilinear(src: torch.Tensor, flow: torch.Tensor, n_steps: int = 1, step0: float = 1.0,
                   border_mode: int = PV_BORDER_CONSTANT, border_value: float = float("nan"),
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """remap_image batched (notebooks/13_...ipynb:259-281): src [N,H,W] (f32 or u8), flow f32 [N,H,W,2]
    -> dst [N, n_steps, H, W]; step s is cv.remap(src, meshgrid - flow*(step0+s), INTER_LINEAR, border)."""
    require_cuda(src, flow)
    n, h, w = src.shape
    if flow.shape != (n, h, w, 2) or flow.dtype != torch.float32:
        raise TypeError("remap_bilinear: flow must be float32 [N,H,W,2] matching src")
    if out is None:
        out = torch.empty((n, n_steps, h, w), dtype=src.dtype, device=src.device)
        img_stride, step_stride = n_steps * h * w, h * w
    else:
        require_cuda(out)
        if out.shape != (n, n_steps, h, w) or out.dtype != src.dtype:
            raise TypeError("remap_bilinear: bad out tensor")
        img_stride, step_stride = out.stride(0), out.stride(1)
    lib = get_lib()
    if src.dtype == torch.float32:
        check(lib.pv_remap_bilinear_f32(ptr(src), h * w, ptr(flow), h * w * 2, ptr(out), img_stride, step_stride, n,
                                        n_steps, step0, h, w, border_mode, border_value, current_stream_ptr()),
              "pv_remap_bilinear_f32")
    elif src.dtype == torch.uint8:
        bv = 0 if border_value != border_value else int(border_value)
        check(lib.pv_remap_bilinear_u8(ptr(src), h * w, ptr(flow), h * w * 2, ptr(out), img_stride, step_stride, n,
                                       n_steps, step0, h, w, border_mode, bv, current_stream_ptr()),
              "pv_remap_bilinear_u8")
    else:
        raise TypeError("remap_bilinear: float32 or uint8 images expected")
    return out


def remap_bilinear_strided(src_ptr: int, src_stride: int, flow: torch.Tensor, out_ptr: int, out_image_stride: int,
                           out_step_stride: int, n: int, n_steps: int, step0: float, h: int, w: int,
                           border_mode: int, border_value: float) -> None:
    """f32 remap on raw strided views (used to write advected frames straight into the conv input stack)."""
    check(get_lib().pv_remap_bilinear_f32(ctypes.c_void_p(src_ptr), src_stride, ptr(flow), h * w * 2,
                                          ctypes.c_void_p(out_ptr), out_image_stride, out_step_stride, n, n_steps,
                                          step0, h, w, border_mode, border_value, current_stream_ptr()),
          "pv_remap_bilinear_f32")


def ssim_mean(im1: torch.Tensor, im2: torch.Tensor, data_range: Optional[float] = None) -> torch.Tensor:
    """Mean structural similarity of image pairs [N, H, W] (uint8 or float32), skimage's defaults (pv_ssim_mean_*): f64 [N]."""
    require_cuda(im1, im2)
    if im1.shape != im2.shape or im1.dim() != 3 or im1.dtype != im2.dtype or im1.dtype not in (torch.uint8, torch.float32):
        raise TypeError("ssim_mean: two uint8 or two float32 tensors [N, H, W] of one shape")
    im1, im2 = im1.contiguous(), im2.contiguous()
    n, h, w = im1.shape
    if data_range is None:
        data_range = 255.0 if im1.dtype == torch.uint8 else 2.0      # scikit-image: the dtype's range
    out = torch.empty(n, dtype=torch.float64, device=im1.device)
    fn = get_lib().pv_ssim_mean_u8 if im1.dtype == torch.uint8 else get_lib().pv_ssim_mean_f32
    check(fn(ptr(im1), h * w, ptr(im2), h * w, n, h, w, float(data_range), ptr(out), current_stream_ptr()), "pv_ssim_mean")
    return out


def prepare_stacks(raw: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, t_out: int, mode: int = 0):
    """raw [B,T,C,H,W] int16/f32 counts -> (u8 [B,C,T,H,W], out f32 [B,C,t_out,H,W] with slices 0..T-1 normalised).
    One pass over raw (pv_prepare_stacks_*); H*W must be a multiple of 8."""
    require_cuda(raw, mean, std)
    if raw.dtype not in (torch.int16, torch.float32) or raw.dim() != 5 or not raw.is_contiguous():
        raise TypeError("prepare_stacks: contiguous int16 or float32 [B,T,C,H,W] expected")
    b, t, c, h, w = raw.shape
    u8 = torch.empty((b, c, t, h, w), dtype=torch.uint8, device=raw.device)
    out = torch.empty((b, c, t_out, h, w), dtype=torch.float32, device=raw.device)
    lib = get_lib()
    fn = lib.pv_prepare_stacks_i16 if raw.dtype == torch.int16 else lib.pv_prepare_stacks_f32
    check(fn(ptr(raw), ptr(u8), ptr(out), b, t, c, h * w, t_out, mode, ptr(mean), ptr(std), None, current_stream_ptr()),
          "pv_prepare_stacks")
    return u8, out


def normalise(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, inner: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(x - mean[c]) / std[c] with c = (flat_index // inner) % len(mean); x int16 or f32."""
    require_cuda(x, mean, std)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    lib = get_lib()
    fn = lib.pv_normalise_i16 if x.dtype == torch.int16 else lib.pv_normalise_f32
    if x.dtype not in (torch.int16, torch.float32):
        raise TypeError("normalise: int16 or float32 expected")
    check(fn(ptr(x), ptr(out), x.numel(), inner, mean.numel(), ptr(mean), ptr(std), current_stream_ptr()),
          "pv_normalise")
    return out


# ------------------------------------------------------------------------------------------------
# Conv3D
# ------------------------------------------------------------------------------------------------
def conv_dims(batch, c_in, c_out, t, h, w, padding=(0, 0, 0)) -> Conv3dDims:
    return Conv3dDims(batch, c_in, c_out, t, h, w, padding[0], padding[1], padding[2])


def bf16_cpad(c: int) -> int:
    v = get_lib().pv_bf16_cpad(c)
    if v <= 0:
        raise ValueError(f"bf16 path supports 1..32 channels, got {c}")
    return v


def pack_ncdhw_f32_to_ndhwc_bf16(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    b, c, t, h, w = x.shape
    xp = torch.empty((b, t, h, w, bf16_cpad(c)), dtype=torch.bfloat16, device=x.device)
    check(get_lib().pv_pack_ncdhw_f32_to_ndhwc_bf16(ptr(x), ptr(xp), b, c, t, h, w, current_stream_ptr()),
          "pv_pack_ncdhw_f32_to_ndhwc_bf16")
    return xp


def pack_split2_ncdhw_f32_to_ndhwc_f16(x: torch.Tensor, maxabs_state: Optional[torch.Tensor] = None, cpad32: bool = False):
    """x f32 [B,C,T,H,W] -> (h, l, state): two half-float [B,T,H,W,CPAD] images with x s = h + l (22 significant bits) and
    state = device f32[3] (scratch, s, 1/s), s the power of two that brings max |x| below 2^14.  T*H*W must be a multiple of 4.
    maxabs_state: the state relu_gate_f32(..., want_max=True) produced together with x (its word 0 holds the bits of max |x|):
    the pass that finds the maximum is skipped.  cpad32: images of 32 channels whatever C (conv3d_f32_on_f16x2's operands)."""
    require_cuda(x)
    if x.dtype != torch.float32 or x.dim() != 5 or not x.is_contiguous():
        raise TypeError("pack_split2_ncdhw_f32_to_ndhwc_f16: a contiguous float32 [B,C,T,H,W] tensor is expected")
    b, c, t, h, w = x.shape
    planes = torch.empty((2, b, t, h, w, 32 if cpad32 else bf16_cpad(c)), dtype=torch.float16, device=x.device)
    state = maxabs_state
    have_max = state is not None
    if not have_max:
        state = torch.empty(3, dtype=torch.float32, device=x.device)
    check(get_lib().pv_pack_split2_ncdhw_f32_to_ndhwc_f16(ptr(x), ptr(planes[0]), ptr(planes[1]), ptr(state), int(have_max) | (2 if cpad32 else 0), b, c, t,
                                                          h, w, current_stream_ptr()), "pv_pack_split2_ncdhw_f32_to_ndhwc_f16")
    return planes[0], planes[1], state


def conv3d_bwd_weight_f16(x: torch.Tensor, dy: torch.Tensor, c_in: int, c_out: int, padding=(0, 0, 0), out=None):
    """conv3d_bwd_weight_bf16 on half-float operand images: x [B,T,H,W,CPAD] f16, dy [B,To,Ho,Wo,32] f16.
    out: (dw, db) contiguous f32 tensors of c_out * c_in * 27 / c_out elements to write into (else fresh ones)."""
    require_cuda(x, dy)
    b, t, h, w, cpad = x.shape
    d = conv_dims(b, c_in, c_out, t, h, w, padding)
    need = c_sz(0)
    lib = get_lib()
    check(lib.pv_conv3d_bwd_weight_bf16_workspace_bytes(ctypes.byref(d), ctypes.byref(need)), "wgrad workspace")
    ws = _workspace("wgrad", need.value, x.device)
    if out is None:
        dw = torch.empty((c_out, c_in, 3, 3, 3), dtype=torch.float32, device=x.device)
        db = torch.empty(c_out, dtype=torch.float32, device=x.device)
    else:
        dw, db = out
        if dw.numel() != c_out * c_in * 27 or db.numel() != c_out or not (dw.is_contiguous() and db.is_contiguous()):
            raise ValueError("conv3d_bwd_weight_f16: out = (dw, db) must be contiguous f32 tensors of the gradients' sizes")
    check(lib.pv_conv3d_bwd_weight_f16(ptr(x), ptr(dy), ptr(dw), ptr(db), ctypes.byref(d), ptr(ws), ws.numel(),
                                       current_stream_ptr()), "pv_conv3d_bwd_weight_f16")
    return dw, db


def conv3d_bwd_weight_f32_on_f16x2(x: torch.Tensor, dy: torch.Tensor, padding=(0, 0, 0), dy_maxabs_state=None):
    """Weight and bias gradient of a 3x3x3, stride-1 Conv3d with 32 output channels at f32 accuracy on the F16 matrix cores:
    x f32 [B,Ci<=32,T,H,W] and dy f32 [B,32,To,Ho,Wo] (already multiplied by the ReLU derivative) are each scaled by a power of
    two and split in two half-float terms (22 bits), the weight-gradient kernel runs on the three operand pairs whose product
    is above 2^-22 of the result -- (l,h), (h,l), (h,h) -- and the three f32 results are added in that order, then un-scaled.
    (Round 3's three-term bf16 split -- six launches, 10 bytes of split traffic per element -- was removed in round 6.)
    -> (dw [32,Ci,3,3,3], db [32])."""
    ci = x.shape[1]
    xh, xl, xs = pack_split2_ncdhw_f32_to_ndhwc_f16(x)
    dh, dl, ds = pack_split2_ncdhw_f32_to_ndhwc_f16(dy, maxabs_state=dy_maxabs_state)
    parts = torch.empty((3, 32 * ci * 27), dtype=torch.float32, device=x.device)
    dbp = torch.empty((2, 32), dtype=torch.float32, device=x.device)
    for i, (px, pd) in enumerate(((xl, dh), (xh, dl), (xh, dh))):
        dw_i, db_i = conv3d_bwd_weight_f16(px, pd, ci, 32, padding)
        parts[i].copy_(dw_i.reshape(-1))
        if i >= 1:
            dbp[i - 1].copy_(db_i)       # the bias gradient is the plain sum of dy: one term per plane of dy (l, then h)
    dw = torch.empty((32, ci, 3, 3, 3), dtype=torch.float32, device=x.device)
    check(get_lib().pv_sum_slabs_acc_f32(ptr(parts), ptr(dw), parts.shape[1], 3, 0, current_stream_ptr()), "pv_sum_slabs_acc_f32")
    db = torch.empty(32, dtype=torch.float32, device=x.device)
    check(get_lib().pv_sum_slabs_acc_f32(ptr(dbp), ptr(db), 32, 2, 0, current_stream_ptr()), "pv_sum_slabs_acc_f32")
    dw.mul_(xs[2]).mul_(ds[2])      # exact: powers of two (device scalars: no host round trip)
    db.mul_(ds[2])
    return dw, db


def conv3d_bwd_weight_f32_from_split2(xh, xl, xs, dh, dl, ds, c_in: int, padding=(0, 0, 0)):
    """conv3d_bwd_weight_f32_on_f16x2 on operand images that already exist (the f16x2 forward made x's, the f16x2 data gradient
    made dy's): three launches, no split pass.  -> (dw [32,c_in,3,3,3], db [32])."""
    parts = torch.empty((3, 32 * c_in * 27), dtype=torch.float32, device=xh.device)
    dbp = torch.empty((3, 32), dtype=torch.float32, device=xh.device)      # (row 0: the bias sum of dy's l image against x_l: not a term)
    for i, (px, pd) in enumerate(((xl, dh), (xh, dl), (xh, dh))):
        conv3d_bwd_weight_f16(px, pd, c_in, 32, padding, out=(parts[i], dbp[(i + 2) % 3]))      # db: (dh, dl, dh) -> rows 2, 0, 1: rows 0, 1 = l, h
    dw = torch.empty((32, c_in, 3, 3, 3), dtype=torch.float32, device=xh.device)
    check(get_lib().pv_sum_slabs_acc_f32(ptr(parts), ptr(dw), parts.shape[1], 3, 0, current_stream_ptr()), "pv_sum_slabs_acc_f32")
    db = torch.empty(32, dtype=torch.float32, device=xh.device)
    check(get_lib().pv_sum_slabs_acc_f32(ptr(dbp), ptr(db), 32, 2, 0, current_stream_ptr()), "pv_sum_slabs_acc_f32")
    dw.mul_(xs[2] * ds[2])      # exact: powers of two (device scalars: no host round trip)
    db.mul_(ds[2])
    return dw, db


def conv3d_pack_weight_split2_f16(w: torch.Tensor):
    """w f32 [c_out<=32, c_in<=32, 3,3,3] -> (wp f16 [4, elems]: forward operator (h, l), data-gradient operator (h, l); state f32[3])."""
    require_cuda(w)
    if w.dtype != torch.float32 or w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3) or not w.is_contiguous():
        raise TypeError("conv3d_pack_weight_split2_f16: a contiguous float32 [Co,Ci,3,3,3] weight is expected")
    lib = get_lib()
    n = int(lib.pv_conv3d_split2_weight_elems())
    wp = torch.empty((4, n // 4), dtype=torch.float16, device=w.device)
    state = torch.empty(69, dtype=torch.float32, device=w.device)      # (max bits, s, 1 / s, -, -, 32 + 32 absolute row sums)
    check(lib.pv_conv3d_pack_weight_split2_f16(ptr(w), ptr(wp), ptr(state), w.shape[0], w.shape[1], current_stream_ptr()),
          "pv_conv3d_pack_weight_split2_f16")
    return wp, state


F16X2_SMALL_PRODUCTS_F16 = True      # conv3d_f32_on_f16x2: x_l w_h and x_h w_l leave the matrix cores as half floats (False: f32, the cross-check)


def conv3d_f16x2_covers(batch, c_in, c_out, t, h, w, padding) -> bool:
    """Does the f16x2 form (conv3d_f32_on_f16x2) take this layer?  (64-byte voxels, >= 2 output slices per time chunk, size limits)"""
    d = conv_dims(batch, c_in, c_out, t, h, w, padding)
    return bool(get_lib().pv_conv3d_fwd_f16_f32out_covers(ctypes.byref(d)))


def conv3d_f32_on_f16x2(xh, xl, xs, wp_h, wp_l, ws, c_in: int, c_out: int, padding=(0, 0, 0), bias=None, relu=False, gate_h=None,
                        data_gradient=False, want_planes=True, want_f32=True):
    """One f32 Conv3d (3x3x3, stride 1, 32 output channels of the OPERATOR) from split operand images: xh / xl f16 [B,T,H,W,32]
    with scale state xs, wp_h / wp_l ONE operator's fragments with state ws (data_gradient: the transposed-flipped operator's).
    Three matrix-core launches into partial tensors, then one pass: ordered sum, un-scale, bias, ReLU / gate (gate_h: the h image
    [B,To,Ho,Wo,32] of the gating activation), NCDHW -- and, want_planes, y's own split for the next layer.
    want_f32=False: only the split is written (a consumer that reads operand images).
    -> (y f32 [B,32,To,Ho,Wo] or None, state f32[3] = (bits of max |y|, s_y, 1 / s_y), planes f16 [2,B,To,Ho,Wo,32] or None)."""
    if not (want_planes or want_f32):
        raise ValueError("conv3d_f32_on_f16x2: nothing to write")
    require_cuda(xh, xl)
    b, t, h, w, cpad = xh.shape
    if cpad != 32:
        raise ValueError("conv3d_f32_on_f16x2: 64-byte voxels (17..32 input channels) are expected")
    d = conv_dims(b, c_in, c_out, t, h, w, padding)
    to, ho, wo = t + 2 * padding[0] - 2, h + 2 * padding[1] - 2, w + 2 * padding[2] - 2
    lib = get_lib()
    st = current_stream_ptr()
    if F16X2_SMALL_PRODUCTS_F16:      # the two small products as half floats (2^-12 of the accumulators): half their bytes
        p01 = torch.empty((2, b, to, ho, wo, 32), dtype=torch.float16, device=xh.device)
        parts = torch.empty((1, b, to, ho, wo, 32), dtype=torch.float32, device=xh.device)
        outs = ((p01[0], 1), (p01[1], 1), (parts[0], 0))
    else:
        p01 = None
        parts = torch.empty((3, b, to, ho, wo, 32), dtype=torch.float32, device=xh.device)
        outs = ((parts[0], 0), (parts[1], 0), (parts[2], 0))
    for (px, pw), (out, is16) in zip(((xl, wp_h), (xh, wp_l), (xh, wp_h)), outs):
        check(lib.pv_conv3d_fwd_f16_f32out(ptr(px), ptr(pw), ptr(out), is16, ctypes.byref(d), st), "pv_conv3d_fwd_f16_f32out")
    y = torch.empty((b, 32, to, ho, wo), dtype=torch.float32, device=xh.device) if want_f32 else None
    state = torch.zeros(3, dtype=torch.float32, device=xh.device)
    planes = torch.empty((2, b, to, ho, wo, 32), dtype=torch.float16, device=xh.device) if want_planes else None
    check(lib.pv_sum3_ndhwc_to_ncdhw_f32(ptr(parts), ptr(p01), ptr(xs), ptr(ws), int(bool(data_gradient)), ptr(bias), ptr(gate_h),
                                         ptr(y), ptr(planes[0]) if want_planes else None, ptr(planes[1]) if want_planes else None,
                                         ptr(state), int(bool(relu)), b, to * ho * wo, st), "pv_sum3_ndhwc_to_ncdhw_f32")
    return y, state, planes


def unpack_ndhwc_bf16_to_ncdhw_f32(xp: torch.Tensor, c: int) -> torch.Tensor:
    require_cuda(xp)
    b, t, h, w, cpad = xp.shape
    x = torch.empty((b, c, t, h, w), dtype=torch.float32, device=xp.device)
    check(get_lib().pv_unpack_ndhwc_bf16_to_ncdhw_f32(ptr(xp), ptr(x), b, c, t, h, w, current_stream_ptr()),
          "pv_unpack_ndhwc_bf16_to_ncdhw_f32")
    return x


def repack_gate_ncdhw_to_ndhwc_bf16(dy: torch.Tensor, y_mask: Optional[torch.Tensor]) -> torch.Tensor:
    require_cuda(dy, y_mask)
    b, c, t, h, w = dy.shape
    out = torch.empty((b, t, h, w, 32), dtype=torch.bfloat16, device=dy.device)
    check(get_lib().pv_repack_gate_ncdhw_to_ndhwc_bf16(ptr(dy), ptr(y_mask), ptr(out), b, c, t, h, w,
                                                       current_stream_ptr()), "pv_repack_gate_ncdhw_to_ndhwc_bf16")
    return out


def conv3d_pack_weight_bf16(weight: torch.Tensor, transpose_flip: bool = False) -> torch.Tensor:
    require_cuda(weight)
    co, ci = weight.shape[:2]
    kch = co if transpose_flip else ci
    n = get_lib().pv_conv3d_packed_weight_elems(kch)
    wp = torch.empty(n, dtype=torch.bfloat16, device=weight.device)
    check(get_lib().pv_conv3d_pack_weight_bf16(ptr(weight), ptr(wp), co, ci, int(transpose_flip), current_stream_ptr()),
          "pv_conv3d_pack_weight_bf16")
    return wp


class stage_timing:
    """Context manager around pv_stage_timing_begin / _end: per-stage device milliseconds of the multi-kernel entry points
    (Farnebäck pyramid, advection stages) recorded with HIP events on the launching stream.
        with K.stage_timing() as st: ...pv calls...
        st.stages  ->  {label: (total_ms, occurrences)} in first-seen order"""

    def __init__(self, capacity: int = 64):
        self.capacity, self.stages = capacity, {}

    def __enter__(self):
        check(get_lib().pv_stage_timing_begin(), "pv_stage_timing_begin")
        return self

    def __exit__(self, *exc):
        names = (ctypes.c_char_p * self.capacity)()
        ms = (ctypes.c_float * self.capacity)()
        counts = (ctypes.c_int32 * self.capacity)()
        n = ctypes.c_int32(0)
        check(get_lib().pv_stage_timing_end(names, ms, counts, self.capacity, ctypes.byref(n)), "pv_stage_timing_end")
        self.stages = {names[i].decode(): (float(ms[i]), int(counts[i])) for i in range(n.value)}
        return False


def device_calibration(copy_floats: int = 128 * 1003520, mfma_iters: int = 20000, reps: int = 5) -> dict:
    """What THIS device sustains, measured with HIP events on the current stream (pv_calibrate_*): a plain device copy of
    `copy_floats` floats (default: the size of fc1's weight) and a bare bf16 matrix-instruction loop on every SIMD."""
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.empty(copy_floats, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    wgs = 256 * 2                                   # two 4-wave workgroups per CU: two waves per SIMD
    sink = torch.zeros(wgs, dtype=torch.float32, device=dev)
    lib = get_lib()

    def timed(fn, n):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3       # seconds per call

    t_copy = timed(lambda: check(lib.pv_calibrate_copy_f32(ptr(a), ptr(b), copy_floats, current_stream_ptr()), "pv_calibrate_copy_f32"), reps * 2)
    t_mfma = timed(lambda: check(lib.pv_calibrate_mfma_bf16(ptr(sink), wgs, mfma_iters, current_stream_ptr()), "pv_calibrate_mfma_bf16"), reps)
    flops = wgs * 4 * mfma_iters * 8 * 16384.0
    # hand the gigabyte back to the driver: left in torch's cache it changed where the NEXT allocations of the caller land, and a
    # bench leg that times individual launches read its first (HBM-bound) kernel at 160-200 us instead of 75 (round 6)
    del a, b, sink
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return {"copy_TBps": round(8.0 * copy_floats / t_copy / 1e12, 3), "copy_bytes": 8 * copy_floats,
            "mfma_bf16_TFLOPs": round(flops / t_mfma / 1e12, 1), "mfma_ms": round(t_mfma * 1e3, 3),
            "what": "pv_calibrate_copy_f32 (16 B per lane, read + write bytes) and pv_calibrate_mfma_bf16 (v_mfma_f32_16x16x32_bf16 "
                    "back to back on register operands, two waves per SIMD, non-trivial data), HIP events, this process, this device"}


def relu_mask_shape(b: int, t: int, h: int, w: int):
    """Shape of the 1-bit relu mask of an NDHWC activation [b, t, h, w, 32]: int32 [b, t, hp, wp], the plane padded to
    whole 8 x 32 tiles (pv_relu_mask_dims)."""
    hp, wp = ctypes.c_int32(), ctypes.c_int32()
    check(get_lib().pv_relu_mask_dims(h, w, ctypes.byref(hp), ctypes.byref(wp)), "pv_relu_mask_dims")
    return (b, t, hp.value, wp.value)


def conv3d_fwd_bf16(x: torch.Tensor, gate: Optional[torch.Tensor], wp: torch.Tensor, bias: Optional[torch.Tensor],
                    c_in: int, c_out: int, padding=(0, 0, 0), relu=True, y_ncdhw=False,
                    out_gate: Optional[torch.Tensor] = None, out_gate_mask: Optional[torch.Tensor] = None,
                    want_relu_mask: bool = False):
    """x [B,T,H,W,CPAD(c_in)] bf16 -> y [B,To,Ho,Wo,32] bf16 (or [B,c_out,To,Ho,Wo] if y_ncdhw).
    want_relu_mask: also return the 1-bit relu mask of y, int32 relu_mask_shape(B,To,Ho,Wo) (bit c = y[..., c] > 0):
    (y, mask).
    out_gate_mask: such a mask of `out_gate` (the dgrad epilogue then reads 4 instead of 64 bytes per voxel)."""
    require_cuda(x, gate, wp, bias, out_gate, out_gate_mask)
    b, t, h, w, cpad = x.shape
    if cpad != bf16_cpad(c_in) or x.dtype != torch.bfloat16:
        raise TypeError("conv3d_fwd_bf16: x must be bf16 [B,T,H,W,CPAD(c_in)]")
    d = conv_dims(b, c_in, c_out, t, h, w, padding)
    to, ho, wo = d.out_shape()
    shape = (b, c_out, to, ho, wo) if y_ncdhw else (b, to, ho, wo, 32)
    y = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
    if out_gate is not None and tuple(out_gate.shape) != shape:
        raise TypeError("conv3d_fwd_bf16: out_gate must have the output's NDHWC shape")
    if out_gate_mask is not None and (out_gate is None or out_gate_mask.dtype != torch.int32
                                      or tuple(out_gate_mask.shape) != relu_mask_shape(*shape[:4])
                                      or not out_gate_mask.is_contiguous()):
        raise TypeError("conv3d_fwd_bf16: out_gate_mask must be the contiguous int32 relu mask of out_gate (relu_mask_shape)")
    mask = torch.empty(relu_mask_shape(*shape[:4]), dtype=torch.int32, device=x.device) if want_relu_mask else None
    check(get_lib().pv_conv3d_fwd_bf16(ptr(x), ptr(gate), ptr(wp), ptr(bias), ptr(y), ptr(out_gate), ptr(out_gate_mask),
                                       ptr(mask), ctypes.byref(d), int(relu), int(y_ncdhw), current_stream_ptr()),
          "pv_conv3d_fwd_bf16")
    return (y, mask) if want_relu_mask else y


def conv3d_fwd_bf16_f32in(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], c_out: int, padding=(0, 0, 0),
                          relu=True, want_packed=True, want_relu_mask=False):
    """First layer: x f32 NCDHW [B,C<=16,T,H,W] -> (y bf16 NDHWC [B,To,Ho,Wo,32], xp bf16 NDHWC [B,T,H,W,16] or None
    [, relu mask of y int32 [B,To,Ho,Wo] when want_relu_mask]).
    One pass over the input instead of pack + conv; xp is what conv3d_bwd_weight_bf16 reads."""
    require_cuda(x, wp, bias)
    if x.dtype != torch.float32 or x.dim() != 5 or not x.is_contiguous():
        raise TypeError("conv3d_fwd_bf16_f32in: x must be a contiguous f32 [B,C,T,H,W] tensor")
    b, c_in, t, h, w = x.shape
    if bf16_cpad(c_in) != 16:
        raise TypeError("conv3d_fwd_bf16_f32in: built for c_in <= 16")
    d = conv_dims(b, c_in, c_out, t, h, w, padding)
    to, ho, wo = d.out_shape()
    y = torch.empty((b, to, ho, wo, 32), dtype=torch.bfloat16, device=x.device)
    xp = torch.empty((b, t, h, w, 16), dtype=torch.bfloat16, device=x.device) if want_packed else None
    mask = torch.empty(relu_mask_shape(b, to, ho, wo), dtype=torch.int32, device=x.device) if want_relu_mask else None
    check(get_lib().pv_conv3d_fwd_bf16_f32in(ptr(x), ptr(xp), ptr(wp), ptr(bias), ptr(y), ptr(mask), ctypes.byref(d),
                                             int(relu), current_stream_ptr()), "pv_conv3d_fwd_bf16_f32in")
    return (y, xp, mask) if want_relu_mask else (y, xp)


def conv3d_bwd_weight_bf16(x: torch.Tensor, dy: torch.Tensor, y_mask: Optional[torch.Tensor], c_in: int, c_out: int,
                           padding=(0, 0, 0)):
    """x [B,T,H,W,CPAD] bf16, dy/y_mask [B,To,Ho,Wo,32] bf16 -> (dw f32 [c_out,c_in,3,3,3], db f32 [c_out])."""
    require_cuda(x, dy, y_mask)
    b, t, h, w, cpad = x.shape
    d = conv_dims(b, c_in, c_out, t, h, w, padding)
    need = c_sz(0)
    lib = get_lib()
    check(lib.pv_conv3d_bwd_weight_bf16_workspace_bytes(ctypes.byref(d), ctypes.byref(need)), "wgrad workspace")
    ws = _workspace("wgrad", need.value, x.device)
    dw = torch.empty((c_out, c_in, 3, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty(c_out, dtype=torch.float32, device=x.device)
    check(lib.pv_conv3d_bwd_weight_bf16(ptr(x), ptr(dy), ptr(y_mask), ptr(dw), ptr(db), ctypes.byref(d), ptr(ws),
                                        ws.numel(), current_stream_ptr()), "pv_conv3d_bwd_weight_bf16")
    return dw, db


# ------------------------------------------------------------------------------------------------
# fully connected head, losses, Adam
# ------------------------------------------------------------------------------------------------
def linear_fwd_f32(x, weight, bias, relu=False):
    require_cuda(x, weight, bias)
    m, k = x.shape
    n = weight.shape[0]
    need = c_sz(0)
    lib = get_lib()
    check(lib.pv_linear_workspace_bytes(m, n, k, ctypes.byref(need)), "pv_linear_workspace_bytes")
    ws = _workspace("linear", need.value, x.device)
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib.pv_linear_fwd_f32(ptr(x), ptr(weight), ptr(bias), ptr(y), m, n, k, int(relu), ptr(ws), ws.numel(),
                                current_stream_ptr()), "pv_linear_fwd_f32")
    return y


LINEAR_F32_SKINNY = True      # fc1-sized f32 Linears (<= 32 rows, <= 128 outputs): forward / dx as streams over the weight (False: the split-product GEMM)


def linear_f32_skinny_covers(m: int, n: int, k: int) -> bool:
    return LINEAR_F32_SKINNY and bool(get_lib().pv_linear_f32_skinny_covers(m, n, k))


def linear_fwd_f32_skinny(x, weight, bias, relu=False):
    """y = relu?(x w^T + bias), f32, exact products: x [m <= 32, k], w [n <= 128, k] (linear_f32_skinny.hip)."""
    require_cuda(x, weight, bias)
    m, k = x.shape
    n = weight.shape[0]
    lib = get_lib()
    ws = _workspace("linear_skinny", int(lib.pv_linear_fwd_f32_skinny_workspace_bytes(n)), x.device)
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(lib.pv_linear_fwd_f32_skinny(ptr(x), ptr(weight), ptr(bias), ptr(y), m, n, k, int(relu), ptr(ws), ws.numel(),
                                       current_stream_ptr()), "pv_linear_fwd_f32_skinny")
    return y


def linear_dx_f32_skinny(g, weight):
    """dx = g w, f32, exact products: g [m <= 32, n <= 128], w [n, k]."""
    require_cuda(g, weight)
    m, n = g.shape
    k = weight.shape[1]
    dx = torch.empty((m, k), dtype=torch.float32, device=g.device)
    check(get_lib().pv_linear_dx_f32_skinny(ptr(g), ptr(weight), ptr(dx), m, n, k, current_stream_ptr()), "pv_linear_dx_f32_skinny")
    return dx


def linear_bwd_f32(x, weight, dy, y_mask, need_dx=True):
    require_cuda(x, weight, dy, y_mask)
    m, k = x.shape
    n = weight.shape[0]
    dx = torch.empty((m, k), dtype=torch.float32, device=x.device) if need_dx else None
    dw = torch.empty((n, k), dtype=torch.float32, device=x.device)
    db = torch.empty(n, dtype=torch.float32, device=x.device)
    check(get_lib().pv_linear_bwd_f32(ptr(x), ptr(weight), ptr(dy), ptr(y_mask), ptr(dx), ptr(dw), ptr(db), m, n, k,
                                      current_stream_ptr()), "pv_linear_bwd_f32")
    return dx, dw, db


LINEAR_FWD_MAX_ROWS = 1024     # pv_linear_fwd_bf16: rows of x per call


def linear_fwd_bf16(x_bf16, w_bf16, bias, relu=False):
    require_cuda(x_bf16, w_bf16, bias)
    m, k = x_bf16.shape
    n = w_bf16.shape[0]
    need = c_sz(0)
    lib = get_lib()
    # rows per call: 1024 for the fc1 shapes (the LDS-staged kernel walks them 64 at a time, one stream over the weights each, and
    # one reduce per call), 128 for the others (the register-tiled kernel's four row tiles)
    cap = LINEAR_FWD_MAX_ROWS if (32 <= n <= 128 and n % 16 == 0) else 128
    check(lib.pv_linear_bf16_workspace_bytes(min(m, cap), n, k, ctypes.byref(need)), "pv_linear_bf16_workspace_bytes")
    ws = _workspace("linear_bf16", need.value, x_bf16.device)
    y = torch.empty((m, n), dtype=torch.float32, device=x_bf16.device)
    for r0 in range(0, m, cap):
        rows = min(cap, m - r0)
        check(lib.pv_linear_fwd_bf16(ptr(x_bf16[r0:r0 + rows]), ptr(w_bf16), ptr(bias), ptr(y[r0:r0 + rows]), rows, n, k, int(relu),
                                     ptr(ws), ws.numel(), current_stream_ptr()), "pv_linear_fwd_bf16")
    return y


def linear_bwd_bf16(x_bf16, w_bf16, dy, y_mask, need_dx=True, need_dw=True, gate_dx_by_x=False, need_db=True):
    """gate_dx_by_x: x is a ReLU output; dx leaves multiplied by (x > 0) (the producer's ReLU derivative)."""
    require_cuda(x_bf16, w_bf16, dy, y_mask)
    m, k = x_bf16.shape
    n = w_bf16.shape[0]
    dx = torch.empty((m, k), dtype=torch.bfloat16, device=dy.device) if need_dx else None
    dw = torch.empty((n, k), dtype=torch.float32, device=dy.device) if need_dw else None
    db = torch.empty(n, dtype=torch.float32, device=dy.device) if need_db else None
    check(get_lib().pv_linear_bwd_bf16(ptr(x_bf16), ptr(w_bf16), ptr(dy), ptr(y_mask), ptr(dx), ptr(dw), ptr(db), m, n,
                                       k, int(bool(gate_dx_by_x and need_dx)), current_stream_ptr()), "pv_linear_bwd_bf16")
    return dx, dw, db


def forecast_losses(y_hat: torch.Tensor, y: torch.Tensor, need_grad: bool = True, grad_scale: float = 1.0,
                    per_horizon: bool = False):
    """(mse, nmae, mse_exp, mae_exp) as a device f32[4] and d nmae / d y_hat (base_model.py:98-103).
    y may be any strided 2-D view (e.g. batch.pv.pv_yield[:, -forecast_len:, 0]).
    per_horizon=True additionally returns f32[2, forecast_len] = (mse, mae) per forecast step (base_model.py:121-141),
    computed by the same launch."""
    require_cuda(y_hat)
    if not y.is_cuda or y.dtype != torch.float32 or y.dim() != 2 or y.shape != y_hat.shape:
        raise TypeError("forecast_losses: y must be a float32 CUDA [B, forecast_len] view matching y_hat")
    m, n = y_hat.shape
    out4 = torch.empty(4, dtype=torch.float32, device=y_hat.device)
    grad = torch.empty_like(y_hat) if need_grad else None
    horizons = torch.empty((2, n), dtype=torch.float32, device=y_hat.device) if per_horizon else None
    check(get_lib().pv_forecast_losses_f32(ptr(y_hat), ctypes.c_void_p(y.data_ptr()), y.stride(0), y.stride(1), m, n,
                                           grad_scale, ptr(out4), ptr(grad), ptr(horizons), current_stream_ptr()),
          "pv_forecast_losses_f32")
    return (out4, grad, horizons) if per_horizon else (out4, grad)


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr=5e-4, betas=(0.9, 0.999), eps=1e-8,
              bf16_shadow=None, grad_scale=1.0):
    require_cuda(param, grad, exp_avg, exp_avg_sq, bf16_shadow)
    check(get_lib().pv_adam_step_f32(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(bf16_shadow),
                                     param.numel(), lr, betas[0], betas[1], eps, step, grad_scale,
                                     current_stream_ptr()), "pv_adam_step_f32")


def adam_step_multi(items, step: int, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    """items: list of (param, grad, exp_avg, exp_avg_sq, bf16_shadow | None) sharing hyper-parameters and step count;
    one launch per _lib.PV_ADAM_MAX_TENSORS tensors instead of one per tensor."""
    lib = get_lib()
    for i in range(0, len(items), _lib.PV_ADAM_MAX_TENSORS):
        chunk = items[i:i + _lib.PV_ADAM_MAX_TENSORS]
        arr = (_lib.AdamTensor * len(chunk))()
        for j, (p, g, m, v, sh) in enumerate(chunk):
            require_cuda(p, g, m, v, sh)
            arr[j] = _lib.AdamTensor(ptr(p), ptr(g), ptr(m), ptr(v), ptr(sh), p.numel())
        check(lib.pv_adam_step_multi_f32(arr, len(chunk), lr, betas[0], betas[1], eps, step, grad_scale,
                                         current_stream_ptr()), "pv_adam_step_multi_f32")


def adam_scalars_advance(scalars: torch.Tensor, step: torch.Tensor, lr=5e-4, betas=(0.9, 0.999), eps=1e-8) -> None:
    """step (int32 [1], device) += 1 and scalars (float32 [>= 6], device) = the Adam scalars of that step: what the `_dev`
    forms of the Adam kernels read, so that a captured HIP graph can replay the optimiser step."""
    require_cuda(scalars, step)
    if scalars.dtype != torch.float32 or scalars.numel() < 6 or step.dtype != torch.int32:
        raise TypeError("adam_scalars_advance: float32 [6] scalars and an int32 step on the device")
    check(get_lib().pv_adam_scalars_advance(ptr(scalars), ptr(step), lr, betas[0], betas[1], eps, current_stream_ptr()),
          "pv_adam_scalars_advance")


def adam_step_multi_dev(items, scalars: torch.Tensor, grad_scale=1.0):
    """adam_step_multi with the step's scalars taken from device memory (adam_scalars_advance)."""
    lib = get_lib()
    require_cuda(scalars)
    for i in range(0, len(items), _lib.PV_ADAM_MAX_TENSORS):
        chunk = items[i:i + _lib.PV_ADAM_MAX_TENSORS]
        arr = (_lib.AdamTensor * len(chunk))()
        for j, (p, g, m, v, sh) in enumerate(chunk):
            require_cuda(p, g, m, v, sh)
            arr[j] = _lib.AdamTensor(ptr(p), ptr(g), ptr(m), ptr(v), ptr(sh), p.numel())
        check(lib.pv_adam_step_multi_dev_f32(arr, len(chunk), ptr(scalars), grad_scale, current_stream_ptr()),
              "pv_adam_step_multi_dev_f32")


def conv3d_pack_weights_multi(jobs):
    """jobs: list of (weight f32 [Co,Ci,3,3,3], packed bf16 buffer, transpose_flip); one launch for all of them."""
    lib = get_lib()
    for i in range(0, len(jobs), _lib.PV_PACK_MAX_JOBS):
        chunk = jobs[i:i + _lib.PV_PACK_MAX_JOBS]
        arr = (_lib.PackJob * len(chunk))()
        for j, (w, wp, flip) in enumerate(chunk):
            require_cuda(w, wp)
            arr[j] = _lib.PackJob(ptr(w), ptr(wp), w.shape[0], w.shape[1], int(flip))
        check(lib.pv_conv3d_pack_weights_multi_bf16(arr, len(chunk), current_stream_ptr()),
              "pv_conv3d_pack_weights_multi_bf16")


def linear_wgrad_adam_bf16(x_bf16, dy, y_mask, param, exp_avg, exp_avg_sq, bf16_shadow, step: int, lr=5e-4,
                           betas=(0.9, 0.999), eps=1e-8):
    """fc1: gradient (dy ⊙ (y>0))^T x computed on the fly and applied by Adam in the same pass (no dw tensor)."""
    require_cuda(x_bf16, dy, y_mask, param, exp_avg, exp_avg_sq, bf16_shadow)
    m, k = x_bf16.shape
    n = param.shape[0]
    check(get_lib().pv_linear_wgrad_adam_bf16(ptr(x_bf16), ptr(dy), ptr(y_mask), ptr(param), ptr(exp_avg),
                                              ptr(exp_avg_sq), ptr(bf16_shadow), m, n, k, lr, betas[0], betas[1], eps,
                                              step, current_stream_ptr()), "pv_linear_wgrad_adam_bf16")


def linear_wgrad_adam_f32(x, dy, y_mask, param, exp_avg, exp_avg_sq, step: int, lr=5e-4, betas=(0.9, 0.999), eps=1e-8):
    """The f32 model's fc1: gradient (dy ⊙ (y>0))^T x from the f32 activations (exact products, f32 accumulation in batch order)
    applied by Adam in the same pass over p, m, v -- no gradient tensor (y_mask may be None: dy already gated)."""
    require_cuda(x, dy, y_mask, param, exp_avg, exp_avg_sq)
    if x.dtype != torch.float32 or not x.is_contiguous() or not dy.is_contiguous():
        raise TypeError("linear_wgrad_adam_f32: contiguous float32 x [M,K] and dy [M,N] expected")
    m, k = x.shape
    n = param.shape[0]
    check(get_lib().pv_linear_wgrad_adam_f32(ptr(x), ptr(dy), ptr(y_mask), ptr(param), ptr(exp_avg), ptr(exp_avg_sq), m, n, k, lr,
                                             betas[0], betas[1], eps, step, current_stream_ptr()), "pv_linear_wgrad_adam_f32")


def linear_wgrad_dx_adam_bf16(x_bf16, dy, y_mask, param, exp_avg, exp_avg_sq, bf16_shadow, step: int, lr=5e-4,
                              betas=(0.9, 0.999), eps=1e-8, need_dx=True, need_db=False, gate_dx_by_x=False,
                              moments_tiled=False):
    """fc1's whole backward in ONE pass over the matrix: weight gradient applied by Adam in place (as
    linear_wgrad_adam_bf16, bit-identical) AND dx = (dy ⊙ (y>0)) . W_old (bf16 [M, K]) from the weights it streams.
    gate_dx_by_x: dx is also multiplied by (x > 0), the ReLU derivative of the layer that produced x.
    moments_tiled: exp_avg / exp_avg_sq hold the [K/128][N][128] tile layout (moments_to_tiled)."""
    require_cuda(x_bf16, dy, y_mask, param, exp_avg, exp_avg_sq, bf16_shadow)
    m, k = x_bf16.shape
    n = param.shape[0]
    dx = torch.empty((m, k), dtype=torch.bfloat16, device=dy.device) if need_dx else None
    db = torch.empty(n, dtype=torch.float32, device=dy.device) if need_db else None
    check(get_lib().pv_linear_wgrad_dx_adam_bf16(ptr(x_bf16), ptr(dy), ptr(y_mask), ptr(param), ptr(exp_avg), ptr(exp_avg_sq),
                                                 ptr(bf16_shadow), ptr(dx), ptr(db), m, n, k, lr, betas[0], betas[1], eps,
                                                 step, int(bool(gate_dx_by_x)), int(bool(moments_tiled)),
                                                 current_stream_ptr()),
          "pv_linear_wgrad_dx_adam_bf16")
    return (dx, db) if need_db else dx


def linear_wgrad_dx_adam_dev_bf16(x_bf16, dy, y_mask, param, exp_avg, exp_avg_sq, bf16_shadow, scalars, need_db=False,
                                  gate_dx_by_x=False, moments_tiled=False):
    """linear_wgrad_dx_adam_bf16 with the step's Adam scalars in device memory (adam_scalars_advance)."""
    require_cuda(x_bf16, dy, y_mask, param, exp_avg, exp_avg_sq, bf16_shadow, scalars)
    m, k = x_bf16.shape
    n = param.shape[0]
    dx = torch.empty((m, k), dtype=torch.bfloat16, device=dy.device)
    db = torch.empty(n, dtype=torch.float32, device=dy.device) if need_db else None
    check(get_lib().pv_linear_wgrad_dx_adam_dev_bf16(ptr(x_bf16), ptr(dy), ptr(y_mask), ptr(param), ptr(exp_avg), ptr(exp_avg_sq),
                                                     ptr(bf16_shadow), ptr(dx), ptr(db), m, n, k, ptr(scalars),
                                                     int(bool(gate_dx_by_x)), int(bool(moments_tiled)), current_stream_ptr()),
          "pv_linear_wgrad_dx_adam_dev_bf16")
    return (dx, db) if need_db else dx


def linear_wgrad_dx_adam_tall_bf16(x_cols, g_all, param, exp_avg, exp_avg_sq, bf16_shadow, step: int, lr=5e-4, betas=(0.9, 0.999),
                                   eps=1e-8, grad_scale=1.0, need_dx=True, gate_dx_by_x=False, moments_tiled=False):
    """K-sharded fc1 (pv_linear_wgrad_dx_adam_tall_bf16): this rank's column shard updated by Adam from the gradient over ALL
    rows of the global batch (x_cols [M, K/W] bf16, g_all [M, N] f32, already gated), and dx_cols = g_all . W_old in the same pass."""
    require_cuda(x_cols, g_all, param, exp_avg, exp_avg_sq, bf16_shadow)
    m, k = x_cols.shape
    n = param.shape[0]
    dx = torch.empty((m, k), dtype=torch.bfloat16, device=g_all.device) if need_dx else None
    need = ctypes.c_size_t()
    check(get_lib().pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes(m, ctypes.byref(need)), "pv_linear_wgrad_dx_adam_tall_bf16_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=g_all.device)      # g_all as operand fragments (40 KB per 32 rows)
    check(get_lib().pv_linear_wgrad_dx_adam_tall_bf16(ptr(x_cols), ptr(g_all), ptr(param), ptr(exp_avg), ptr(exp_avg_sq),
                                                      ptr(bf16_shadow), ptr(dx), m, n, k, lr, betas[0], betas[1], eps, step,
                                                      float(grad_scale), int(bool(gate_dx_by_x)), int(bool(moments_tiled)), ptr(ws),
                                                      need.value, current_stream_ptr()),
          "pv_linear_wgrad_dx_adam_tall_bf16")
    return dx


def kshard_one_pass_supported(n: int, k: int) -> bool:
    return n <= 128 and n % 8 == 0 and k % 8 == 0


MOMENT_TILE = 128      # FD_KT of linear_bf16.hip: k-columns per workgroup of the one-pass fc1 backward


def moments_to_tiled(t: torch.Tensor) -> torch.Tensor:
    """Row-major [N, K] -> the tile layout [K/128][N][128] of pv_linear_wgrad_dx_adam_bf16(moments_tiled=1), as a tensor
    that keeps the shape [N, K] (only its bytes are permuted)."""
    n, k = t.shape
    return t.view(n, k // MOMENT_TILE, MOMENT_TILE).permute(1, 0, 2).contiguous().view(n, k)


def moments_to_rows(t: torch.Tensor) -> torch.Tensor:
    """Inverse of moments_to_tiled."""
    n, k = t.shape
    return t.view(k // MOMENT_TILE, n, MOMENT_TILE).permute(1, 0, 2).contiguous().view(n, k)


def fused_dx_update_supported(m: int, n: int, k: int, capturable: bool = False) -> bool:
    """m <= 32: pv_linear_wgrad_dx_adam_bf16 (or its device-scalar twin under a captured graph); more rows: the row-block form
    pv_linear_wgrad_dx_adam_tall_bf16 (host scalars only)."""
    return (m <= 32 or (m <= 4096 and not capturable)) and n <= 128 and n % 8 == 0 and k % 8 == 0


def linear_wgrad_bf16out(x_bf16, dy, y_mask, n: int) -> torch.Tensor:
    """fc1 weight gradient written directly as bf16 [N,K] (data-parallel wire format)."""
    require_cuda(x_bf16, dy, y_mask)
    m, k = x_bf16.shape
    dw = torch.empty((n, k), dtype=torch.bfloat16, device=dy.device)
    check(get_lib().pv_linear_wgrad_bf16out(ptr(x_bf16), ptr(dy), ptr(y_mask), ptr(dw), m, n, k, current_stream_ptr()),
          "pv_linear_wgrad_bf16out")
    return dw


def adam_step_bf16grad(param, grad_bf16, exp_avg, exp_avg_sq, step: int, lr=5e-4, betas=(0.9, 0.999), eps=1e-8,
                       bf16_shadow=None, grad_scale=1.0):
    require_cuda(param, grad_bf16, exp_avg, exp_avg_sq, bf16_shadow)
    check(get_lib().pv_adam_step_bf16grad(ptr(param), ptr(grad_bf16), ptr(exp_avg), ptr(exp_avg_sq), ptr(bf16_shadow),
                                          param.numel(), lr, betas[0], betas[1], eps, step, grad_scale,
                                          current_stream_ptr()), "pv_adam_step_bf16grad")


def scale_bias_relu_f32(x: torch.Tensor, bias: Optional[torch.Tensor] = None, alpha: float = 1.0, relu: bool = False) -> torch.Tensor:
    """act(alpha x + bias) over the rows of x [M, N] (pv_scale_bias_relu_f32)."""
    require_cuda(x, bias)
    x = x.contiguous()
    y = torch.empty_like(x)
    m = x.numel() // x.shape[-1]
    check(get_lib().pv_scale_bias_relu_f32(ptr(x), ptr(bias), ptr(y), m, x.shape[-1], float(alpha), int(bool(relu)),
                                           current_stream_ptr()), "pv_scale_bias_relu_f32")
    return y


def scale_f32(x: torch.Tensor, alpha: float) -> torch.Tensor:
    return scale_bias_relu_f32(x, None, alpha, False)


def clock_watch_launch(n_samples: int, sleep_units: int, stream) -> torch.Tensor:
    """Starts the one-wave clock watcher on `stream` (a torch.cuda.Stream other than the one the kernels of interest run on) and
    returns its sample buffer int64 [n_samples, 2] = (shader cycles, 100 MHz ticks); read it after stream.synchronize()."""
    buf = torch.zeros((n_samples, 2), dtype=torch.int64, device="cuda")
    check(get_lib().pv_clock_watch(ptr(buf), n_samples, sleep_units, ctypes.c_void_p(stream.cuda_stream)), "pv_clock_watch")
    return buf


def engine_clock_under(fn, seconds: float = 0.02) -> dict:
    """The engine clock (MHz) the device holds while `fn` (something that launches kernels on the current stream) loops for
    `seconds`: the one-wave watcher on a side stream, samples taken well inside the loop.  sysfs reads a ~10 ms average and cannot
    see a 70 us kernel; the conv kernels run at ~1.7-1.8 GHz where the idle device and a plain copy read 2.4."""
    import time
    side = torch.cuda.Stream()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = int((seconds + 0.012) * 1e6 / 1.1) + 1000
    buf = clock_watch_launch(n, 40, side)
    time.sleep(0.002)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype(np.float64)
    s = s[s[:, 1] > 0]
    t_us = (s[:, 1] - s[0, 1]) / 100.0
    dt, dr = np.diff(s[:, 0]), np.diff(s[:, 1])
    ok = (dr > 0) & (t_us[1:] > 6000) & (t_us[1:] < 2000 + seconds * 1e6 - 3000)
    mhz = dt[ok] / dr[ok] * 100.0
    if len(mhz) < 50:
        return {"samples": int(len(mhz))}
    return {"median_MHz": round(float(np.median(mhz))), "p10_MHz": round(float(np.percentile(mhz, 10))),
            "p90_MHz": round(float(np.percentile(mhz, 90))), "samples": int(len(mhz))}


def swap01_segments(x: torch.Tensor) -> torch.Tensor:
    """x [n0, n1, seg] contiguous -> [n1, n0, seg] contiguous (pv_swap01_segments; segments of a multiple of 16 bytes): the
    staging copy of an all-to-all whose chunks are column slices."""
    require_cuda(x)
    if x.dim() != 3 or not x.is_contiguous() or (x.shape[2] * x.element_size()) % 16:
        raise TypeError("swap01_segments: a contiguous [n0, n1, seg] tensor with segments of a multiple of 16 bytes")
    out = torch.empty((x.shape[1], x.shape[0], x.shape[2]), dtype=x.dtype, device=x.device)
    check(get_lib().pv_swap01_segments(ptr(x), ptr(out), x.shape[0], x.shape[1], x.shape[2] * x.element_size(), current_stream_ptr()),
          "pv_swap01_segments")
    return out


def cast_f32_to_bf16(x: torch.Tensor) -> torch.Tensor:
    """bf16 (round to nearest even) copy of a contiguous f32 tensor (pv_cast_f32_to_bf16)."""
    require_cuda(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(get_lib().pv_cast_f32_to_bf16(ptr(x), ptr(out), x.numel(), current_stream_ptr()), "pv_cast_f32_to_bf16")
    return out


def embedding_fwd(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    require_cuda(table, ids)
    ids = ids.to(torch.int64).contiguous()
    out = torch.empty((ids.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    check(get_lib().pv_embedding_fwd_f32(ptr(table), ptr(ids), ptr(out), ids.numel(), table.shape[1], table.shape[0],
                                         current_stream_ptr()), "pv_embedding_fwd_f32")
    return out


def embedding_bwd(dout: torch.Tensor, ids: torch.Tensor, n_rows: int) -> torch.Tensor:
    require_cuda(dout, ids)
    ids = ids.to(torch.int64).contiguous()
    dtable = torch.empty((n_rows, dout.shape[1]), dtype=torch.float32, device=dout.device)
    check(get_lib().pv_embedding_bwd_f32(ptr(dout), ptr(ids), ptr(dtable), ids.numel(), dout.shape[1], n_rows,
                                         current_stream_ptr()), "pv_embedding_bwd_f32")
    return dtable


# ---- general Conv3D / MaxPool3d / MSE (optical-flow notebook model, Conv3dMaxPool) -----------------------------------
def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(a) for a in v)


def conv_geom(batch, c_in, c_out, t, h, w, kernel, stride=1, padding=0) -> Conv3dGeom:
    return Conv3dGeom(batch, c_in, c_out, t, h, w, *_triple(kernel), *_triple(stride), *_triple(padding))


def relu_gate_f32(dy: torch.Tensor, y: torch.Tensor, want_max: bool = False) -> torch.Tensor:
    """dy where y > 0, else 0 (the backward of F.relu, model.py:117-120), f32.  want_max: -> (out, state): the pass also
    leaves the bits of the largest |out| in state[0] -- the scale pack_split2_ncdhw_f32_to_ndhwc_f16 needs, found without a pass
    of its own (hand it over as maxabs_state)."""
    require_cuda(dy, y)
    if dy.dtype != torch.float32 or y.dtype != torch.float32 or dy.numel() != y.numel() or dy.numel() % 4:
        raise TypeError("relu_gate_f32: two float32 tensors of equal size (a multiple of 4 elements)")
    dy, y = dy.contiguous(), y.contiguous()
    out = torch.empty_like(dy)
    if want_max:
        state = torch.empty(3, dtype=torch.float32, device=dy.device)
        check(get_lib().pv_relu_gate_max_f32(ptr(dy), ptr(y), ptr(out), dy.numel(), ptr(state), current_stream_ptr()),
              "pv_relu_gate_max_f32")
        return out, state
    check(get_lib().pv_relu_gate_f32(ptr(dy), ptr(y), ptr(out), dy.numel(), current_stream_ptr()), "pv_relu_gate_f32")
    return out


def conv3d_general_fwd_f32(x, weight, bias, stride=1, padding=0, relu=False):
    require_cuda(x, weight, bias)
    b, ci, t, h, w = x.shape
    co = weight.shape[0]
    g = conv_geom(b, ci, co, t, h, w, tuple(weight.shape[2:]), stride, padding)
    y = torch.empty((b, co) + g.out_shape(), dtype=torch.float32, device=x.device)
    check(get_lib().pv_conv3d_general_fwd_f32(ptr(x), ptr(weight), ptr(bias), ptr(y), ctypes.byref(g), int(relu),
                                              current_stream_ptr()), "pv_conv3d_general_fwd_f32")
    return y


def conv3d_general_bwd_data_f32(dy, y_mask, weight, x_shape, stride=1, padding=0, x_mask=None):
    require_cuda(dy, y_mask, weight, x_mask)
    b, ci, t, h, w = x_shape
    g = conv_geom(b, ci, weight.shape[0], t, h, w, tuple(weight.shape[2:]), stride, padding)
    dx = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    check(get_lib().pv_conv3d_general_bwd_data_f32(ptr(dy), ptr(y_mask), ptr(weight), ptr(dx), ptr(x_mask), ctypes.byref(g),
                                                   current_stream_ptr()), "pv_conv3d_general_bwd_data_f32")
    return dx


def conv3d_general_bwd_weight_f32(x, dy, y_mask, weight_shape, stride=1, padding=0, need_bias=True):
    require_cuda(x, dy, y_mask)
    b, ci, t, h, w = x.shape
    g = conv_geom(b, ci, weight_shape[0], t, h, w, tuple(weight_shape[2:]), stride, padding)
    nbytes = ctypes.c_size_t(0)
    check(get_lib().pv_conv3d_general_bwd_weight_workspace_bytes(ctypes.byref(g), ctypes.byref(nbytes)),
          "pv_conv3d_general_bwd_weight_workspace_bytes")
    ws = _workspace("conv3d_general_wgrad", nbytes.value, x.device)
    dw = torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    db = torch.empty((weight_shape[0],), dtype=torch.float32, device=x.device) if need_bias else None
    check(get_lib().pv_conv3d_general_bwd_weight_f32(ptr(x), ptr(dy), ptr(y_mask), ptr(dw), ptr(db), ctypes.byref(g),
                                                     ptr(ws), nbytes.value, current_stream_ptr()),
          "pv_conv3d_general_bwd_weight_f32")
    return dw, db


def maxpool3d_fwd_f32(x, kernel=3, stride=None, padding=0):
    require_cuda(x)
    b, c, t, h, w = x.shape
    g = conv_geom(b, c, c, t, h, w, kernel, kernel if stride is None else stride, padding)
    y = torch.empty((b, c) + g.out_shape(), dtype=torch.float32, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.int32, device=x.device)
    check(get_lib().pv_maxpool3d_fwd_f32(ptr(x), ptr(y), ptr(idx), ctypes.byref(g), current_stream_ptr()),
          "pv_maxpool3d_fwd_f32")
    return y, idx, g


def maxpool3d_bwd_f32(dy, idx, g: Conv3dGeom):
    require_cuda(dy, idx)
    dx = torch.empty((g.batch, g.c_in, g.t_in, g.h_in, g.w_in), dtype=torch.float32, device=dy.device)
    check(get_lib().pv_maxpool3d_bwd_f32(ptr(dy), ptr(idx), ptr(dx), ctypes.byref(g), current_stream_ptr()),
          "pv_maxpool3d_bwd_f32")
    return dx


def mse_loss(y_hat, y, need_grad=True, grad_scale=1.0):
    require_cuda(y_hat, y)
    if y_hat.shape != y.shape:
        raise ValueError(f"mse_loss: shapes differ: {tuple(y_hat.shape)} vs {tuple(y.shape)}")
    out = torch.empty((1,), dtype=torch.float32, device=y_hat.device)
    grad = torch.empty_like(y_hat) if need_grad else None
    check(get_lib().pv_mse_loss_f32(ptr(y_hat), ptr(y), y_hat.numel(), float(grad_scale), ptr(out), ptr(grad),
                                    current_stream_ptr()), "pv_mse_loss_f32")
    return out, grad


# ---- Perceiver path: strided batched GEMM on the f32 matrix cores + row-wise kernels -----------------------------------
def _require_device(*tensors):
    """Like require_cuda but for STRIDED views: the GEMM reads its operands through explicit element strides."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("predict_pv_yield_amd: tensors must live on the MI355X (no CPU path is provided)")


def _gemm_views(a: torch.Tensor, b: torch.Tensor):
    """a [..., M, K], b [..., K, N] with the same (<= 2) leading batch dims (stride-0 broadcast allowed)."""
    if a.dim() != b.dim() or a.dim() < 2 or a.dim() > 4 or a.shape[:-2] != b.shape[:-2] or a.shape[-1] != b.shape[-2]:
        raise ValueError(f"gemm: incompatible shapes {tuple(a.shape)} x {tuple(b.shape)}")
    if a.dtype not in (torch.float32, torch.bfloat16) or b.dtype != torch.float32:      # (a bf16 A: see gemm's bf16_operands)
        raise TypeError("gemm: float32 operands expected")
    batch = tuple(a.shape[:-2])
    while len(batch) < 2:
        a, b, batch = a.unsqueeze(0), b.unsqueeze(0), (1,) + batch
    return a, b, batch


GEMM_BF16_OPERANDS = 1      # PV_GEMM_BF16_OPERANDS of include/pv_yield_hip.h
GEMM_A_IS_BF16 = 2          # PV_GEMM_A_IS_BF16


def gemm(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
         out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, bf16_operands: bool = False) -> torch.Tensor:
    """C = A @ B (+ bias) (+ residual) for arbitrarily STRIDED views (transposes, column slices and per-head permutes are
    free: only the strides change).  `out`: optional view [..., M, N] with unit last stride to write into.
    `residual`: a [M, N] tensor with unit last stride added in the epilogue (2-D products only).
    bf16_operands: both operands rounded once to bf16, one matrix-core product, f32 accumulation (torch.autocast's linear);
    default: the f32-accurate three-term form."""
    _require_device(a, b, out, residual)
    require_cuda(bias)
    a_bf16 = a.dtype == torch.bfloat16      # (with bf16_operands: A may be STORED as bf16 -- read as it is)
    if a_bf16 and not bf16_operands:
        raise ValueError("gemm: a bfloat16 A needs bf16_operands=True")
    lead = tuple(a.shape[:-2])
    a4, b4, batch = _gemm_views(a, b)
    m, k, n = a4.shape[-2], a4.shape[-1], b4.shape[-1]
    if out is None:
        out = torch.empty(lead + (m, n), dtype=torch.float32, device=a.device)
    c4 = out
    while c4.dim() < 4:
        c4 = c4.unsqueeze(0)
    if tuple(c4.shape) != batch + (m, n) or c4.stride(-1) != 1:
        raise ValueError("gemm: bad output view")
    d = _lib.GemmDesc(m, n, k, a4.stride(-2), a4.stride(-1), b4.stride(-2), b4.stride(-1), c4.stride(-2), batch[0], batch[1],
                      a4.stride(0), a4.stride(1), b4.stride(0), b4.stride(1), c4.stride(0), c4.stride(1), 1, 0)
    if residual is not None:
        if batch != (1, 1) or tuple(residual.shape) != (m, n) or residual.stride(-1) != 1 or residual.dtype != torch.float32:
            raise ValueError("gemm: residual must be a float32 [M, N] tensor with unit last stride (2-D products only)")
    check(get_lib().pv_gemm_ex_f32(ptr(a4), ptr(b4), ptr(bias), ptr(residual), residual.stride(0) if residual is not None else 0,
                                   ptr(c4), ctypes.byref(d), int(relu),
                                   (GEMM_BF16_OPERANDS if bf16_operands else 0) | (GEMM_A_IS_BF16 if a_bf16 else 0),
                                   current_stream_ptr()), "pv_gemm_ex_f32")
    return out


def gemm_rows_bf16out_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    """pv_gemm_rows_bf16out_f32's shape rule: one tall row-major A [M, K <= 64] (16-byte aligned), B [K, N]."""
    return (a.dim() == 2 and b.dim() == 2 and a.dtype == torch.float32 and a.stride(1) == 1 and a.shape[1] <= 64
            and a.shape[0] >= 2048 and a.data_ptr() % 16 == 0)


def gemm_rows_bf16out(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, bf16_operands: bool = False) -> torch.Tensor:
    """C (bf16) = A @ B (+ bias) for a tall row-major A with K <= 64: f32-accurate products, ONE rounding in the store (the
    key / value projection of a cross-attention in bf16-operand mode)."""
    _require_device(a, b)
    require_cuda(bias)
    if not gemm_rows_bf16out_supported(a, b):
        raise ValueError("gemm_rows_bf16out: a tall row-major float32 A [M >= 2048, K <= 64] is expected")
    m, k, n = a.shape[0], a.shape[1], b.shape[1]
    out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    d = _lib.GemmDesc(m, n, k, a.stride(0), a.stride(1), b.stride(0), b.stride(1), n, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0)
    check(get_lib().pv_gemm_rows_bf16out_f32(ptr(a), ptr(b), ptr(bias), ptr(out), ctypes.byref(d),
                                             GEMM_BF16_OPERANDS if bf16_operands else 0, current_stream_ptr()),
          "pv_gemm_rows_bf16out_f32")
    return out


# split-K sizing: enough workgroups to fill the chip, but a K chunk long enough to amortise a workgroup's fixed costs
SPLITK_TARGET_WORKGROUPS = 1024
SPLITK_MIN_CHUNK = 256


def gemm_splitk(a: torch.Tensor, b: torch.Tensor, accumulate_into: Optional[torch.Tensor] = None,
                bf16_operands: bool = False) -> torch.Tensor:
    """C[M, N] = A[M, K] @ B[K, N] for small M, N and a huge K (weight gradients): K is cut over workgroups, the partial
    products are summed in index order (deterministic).  accumulate_into: a contiguous [M, N] tensor that receives
    `+= C` (the slab sum adds into it) and is returned."""
    _require_device(a, b, accumulate_into)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0]:
        raise ValueError(f"gemm_splitk: incompatible shapes {tuple(a.shape)} x {tuple(b.shape)}")
    m, k, n = a.shape[0], a.shape[1], b.shape[1]
    if accumulate_into is not None and (tuple(accumulate_into.shape) != (m, n) or not accumulate_into.is_contiguous()):
        raise ValueError("gemm_splitk: accumulate_into must be a contiguous [M, N] tensor")
    tiles = ((m + 127) // 128) * ((n + 63) // 64)
    splits = max(1, min((SPLITK_TARGET_WORKGROUPS + tiles - 1) // tiles, (k + SPLITK_MIN_CHUNK - 1) // SPLITK_MIN_CHUNK, 4096))
    if splits == 1 and accumulate_into is None:
        return gemm(a, b, bf16_operands=bf16_operands)
    slabs = _workspace("gemm_splitk", splits * m * n * 4, a.device)
    d = _lib.GemmDesc(m, n, k, a.stride(0), a.stride(1), b.stride(0), b.stride(1), n, 1, 1, 0, 0, 0, 0, 0, 0, splits, m * n)
    a_bf16 = a.dtype == torch.bfloat16
    if a_bf16 and not bf16_operands:
        raise ValueError("gemm_splitk: a bfloat16 A needs bf16_operands=True")
    check(get_lib().pv_gemm_ex_f32(ptr(a), ptr(b), None, None, 0, ptr(slabs), ctypes.byref(d), 0,
                                   (GEMM_BF16_OPERANDS if bf16_operands else 0) | (GEMM_A_IS_BF16 if a_bf16 else 0),
                                   current_stream_ptr()), "pv_gemm_ex_f32")
    out = accumulate_into if accumulate_into is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    check(get_lib().pv_sum_slabs_acc_f32(ptr(slabs), ptr(out), m * n, splits, int(accumulate_into is not None),
                                         current_stream_ptr()), "pv_sum_slabs_acc_f32")
    return out


def colsum(x: torch.Tensor, accumulate_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Column sums of a contiguous [rows, cols] tensor (the bias gradient of nn.Linear), chunk partials added in order."""
    require_cuda(x, accumulate_into)
    if x.dim() != 2 or x.dtype != torch.float32:
        raise TypeError("colsum: a contiguous float32 [rows, cols] tensor is expected")
    rows, cols = x.shape
    n_ws = get_lib().pv_colsum_workspace_floats(rows, cols)
    ws = _workspace("colsum", n_ws * 4, x.device)
    out = accumulate_into if accumulate_into is not None else torch.empty(cols, dtype=torch.float32, device=x.device)
    check(get_lib().pv_colsum_f32(ptr(x), ptr(out), rows, cols, ptr(ws), int(accumulate_into is not None),
                                  current_stream_ptr()), "pv_colsum_f32")
    return out


def layernorm_fwd(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    require_cuda(x, w, b)
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(get_lib().pv_layernorm_fwd_f32(ptr(x), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, d, eps,
                                         current_stream_ptr()), "pv_layernorm_fwd_f32")
    return y, mean, rstd


def layernorm_bwd(x, w, dy, mean, rstd, need_dx: bool = True, accumulate_into=None, dx_add=None):
    """-> (dx | None, dw, db).  For d % 4 == 0, dw and db are the two halves of ONE [2 d] tensor (one reduction launch).
    accumulate_into: a (dw, db) pair returned by an earlier call, which receives `+=` instead.
    dx_add: a tensor like x added to dx in the kernel's store (the gradient that reaches x past the normalised branch)."""
    require_cuda(x, w, dy, mean, rstd, dx_add)
    if dx_add is not None and (not need_dx or dx_add.shape != x.shape or not dx_add.is_contiguous()):
        raise ValueError("layernorm_bwd: dx_add must be a contiguous tensor like x (and need_dx)")
    d = x.shape[-1]
    rows = x.numel() // d
    nbytes = ctypes.c_size_t(0)
    check(get_lib().pv_layernorm_bwd_workspace_bytes(rows, d, ctypes.byref(nbytes)), "pv_layernorm_bwd_workspace_bytes")
    ws = _workspace("layernorm_bwd", nbytes.value, x.device)
    dx = torch.empty_like(x) if need_dx else None
    if accumulate_into is not None:
        dw, db = accumulate_into
    elif d % 4 == 0:     # halves of one buffer stay 16-byte aligned (the optimizer kernels require it)
        dwdb = torch.empty(2 * d, dtype=torch.float32, device=x.device)
        dw, db = dwdb[:d], dwdb[d:]
    else:
        dw = torch.empty(d, dtype=torch.float32, device=x.device)
        db = torch.empty(d, dtype=torch.float32, device=x.device)
    check(get_lib().pv_layernorm_bwd_f32(ptr(x), ptr(w), ptr(dy), ptr(mean), ptr(rstd), ptr(dx), ptr(dw), ptr(db), rows, d,
                                         ptr(ws), nbytes.value, int(accumulate_into is not None), ptr(dx_add),
                                         current_stream_ptr()), "pv_layernorm_bwd_f32")
    return dx, dw, db


def layernorm_bwd_params_from_proj_supported(dkv16: torch.Tensor, w_kv: torch.Tensor, x: torch.Tensor) -> bool:
    return (dkv16.dtype == torch.bfloat16 and dkv16.is_contiguous() and w_kv.is_contiguous() and x.is_contiguous()
            and x.dtype == torch.float32 and w_kv.dtype == torch.float32 and w_kv.shape[0] in (64, 128)
            and x.shape[-1] == w_kv.shape[1] <= 64 and dkv16.shape[-1] == w_kv.shape[0]
            and dkv16.numel() // dkv16.shape[-1] == x.numel() // x.shape[-1] and dkv16.data_ptr() % 16 == 0)


def layernorm_bwd_params_from_proj(dkv16, w_kv, x, mean, rstd, accumulate_into=None):
    """(dw, db) of the LayerNorm in  x -> LayerNorm -> to_kv  from the bf16 gradient of the projection, d(LayerNorm output)
    never materialised (pv_layernorm_bwd_params_from_proj_bf16).  dw / db as layernorm_bwd returns them."""
    require_cuda(dkv16, w_kv, x, mean, rstd)
    if not layernorm_bwd_params_from_proj_supported(dkv16, w_kv, x):
        raise ValueError("layernorm_bwd_params_from_proj: bf16 gradient rows [rows, 64 | 128], f32 weight [kdim, d <= 64], f32 x")
    d = x.shape[-1]
    rows = x.numel() // d
    nbytes = ctypes.c_size_t(0)
    check(get_lib().pv_layernorm_bwd_params_from_proj_workspace_bytes(rows, d, ctypes.byref(nbytes)),
          "pv_layernorm_bwd_params_from_proj_workspace_bytes")
    ws = _workspace("layernorm_bwd_proj", nbytes.value, x.device)
    if accumulate_into is not None:
        dw, db = accumulate_into
    elif d % 4 == 0:
        dwdb = torch.empty(2 * d, dtype=torch.float32, device=x.device)
        dw, db = dwdb[:d], dwdb[d:]
    else:
        dw = torch.empty(d, dtype=torch.float32, device=x.device)
        db = torch.empty(d, dtype=torch.float32, device=x.device)
    check(get_lib().pv_layernorm_bwd_params_from_proj_bf16(ptr(dkv16), ptr(w_kv), ptr(x), ptr(mean), ptr(rstd), ptr(dw), ptr(db),
                                                           rows, d, w_kv.shape[0], ptr(ws), nbytes.value,
                                                           int(accumulate_into is not None), current_stream_ptr()),
          "pv_layernorm_bwd_params_from_proj_bf16")
    return dw, db


def _context_rows(x: torch.Tensor, x2):
    """(rows, d, d1, period) of a context given as x [..., d] or as x [..., P, d1] + x2 [P, d2] (row r = x row r followed by
    x2 row r % P: image channels + the position features every image shares)."""
    d1 = x.shape[-1]
    rows = x.numel() // d1
    if x2 is None:
        return rows, d1, 0, 0
    return rows, d1 + x2.shape[-1], d1, x2.shape[0]


def context_fwd_supported(x: torch.Tensor, w_kv: torch.Tensor, x2=None) -> bool:
    rows, d, d1, period = _context_rows(x, x2)
    # (period >= 32: pv_context_bwd_bf16's rule -- a forward the backward cannot follow must not be chosen)
    if x2 is not None and not (x2.dim() == 2 and x2.dtype == torch.float32 and x2.is_contiguous() and x.dim() >= 2
                               and x.shape[-2] == period and period >= 32 and d1 % 2 == 0 and x2.data_ptr() % 8 == 0
                               and x2.device == x.device):
        return False
    return (x.dtype == torch.float32 and x.is_contiguous() and w_kv.is_contiguous() and w_kv.dtype == torch.float32
            and w_kv.shape[0] == 128 and d == w_kv.shape[1] <= 48 and d % 2 == 0
            and x.data_ptr() % 16 == 0 and 2048 <= rows < 2 ** 31)


def context_fwd(x, ln_w, ln_b, w_kv, eps: float = 1e-5, x2=None):
    """K | V (bf16) = LayerNorm(context) @ w_kv^T in one pass (pv_context_fwd_bf16) -> (kv16 [..., 128], mean, rstd).
    context = x [..., d], or x [..., P, d1] with x2 [P, d2] appended to every group of P rows (never concatenated in memory)."""
    require_cuda(x, ln_w, ln_b, w_kv, x2)
    if not context_fwd_supported(x, w_kv, x2):
        raise ValueError("context_fwd: contiguous f32 x [..., d <= 48, even] (or x [..., P, d1] + x2 [P, d2]) and w_kv [128, d] expected")
    rows, d, d1, period = _context_rows(x, x2)
    kv16 = torch.empty(x.shape[:-1] + (128,), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(get_lib().pv_context_fwd_bf16(ptr(x), ptr(x2), d1, period, ptr(ln_w), ptr(ln_b), ptr(w_kv), ptr(kv16), ptr(mean),
                                        ptr(rstd), rows, d, 128, float(eps), current_stream_ptr()), "pv_context_fwd_bf16")
    return kv16, mean, rstd


def context_bwd_supported(dkv16: torch.Tensor, w_kv: torch.Tensor, x: torch.Tensor, x2=None) -> bool:
    rows, d, d1, period = _context_rows(x, x2)
    if x2 is not None and not (x2.dim() == 2 and x2.dtype == torch.float32 and x2.is_contiguous() and x.dim() >= 2
                               and x.shape[-2] == period and period >= 32 and x2.device == x.device):
        return False
    return (dkv16.dtype == torch.bfloat16 and dkv16.is_contiguous() and w_kv.is_contiguous() and x.is_contiguous()
            and x.dtype == torch.float32 and w_kv.dtype == torch.float32 and w_kv.shape[0] == 128 and d == w_kv.shape[1] <= 64
            and dkv16.shape[-1] == 128 and dkv16.numel() // 128 == rows < 2 ** 31 and dkv16.data_ptr() % 16 == 0)


def context_bwd(dkv16, w_kv, x, mean, rstd, ln_w, ln_b, accumulate_kv_into=None, accumulate_ln_into=None, x2=None):
    """The backward of  context -> LayerNorm -> to_kv  for a context that takes no gradient, from the bf16 gradient rows of K | V,
    in one pass (pv_context_bwd_bf16): -> (dw_kv [128, d], dln_w, dln_b).  accumulate_*_into: tensors that receive `+=` instead.
    x2: as in context_fwd."""
    require_cuda(dkv16, w_kv, x, mean, rstd, ln_w, ln_b, x2)
    if not context_bwd_supported(dkv16, w_kv, x, x2):
        raise ValueError("context_bwd: bf16 gradient rows [rows, 128], f32 weight [128, d <= 64], f32 context rows")
    rows, d, d1, period = _context_rows(x, x2)
    nbytes = ctypes.c_size_t(0)
    check(get_lib().pv_context_bwd_workspace_bytes(rows, d, ctypes.byref(nbytes)), "pv_context_bwd_workspace_bytes")
    ws = _workspace("context_bwd", nbytes.value, x.device)
    dw = accumulate_kv_into if accumulate_kv_into is not None else torch.empty_like(w_kv)
    if accumulate_ln_into is not None:
        dlw, dlb = accumulate_ln_into
    elif d % 4 == 0:
        both = torch.empty(2 * d, dtype=torch.float32, device=x.device)
        dlw, dlb = both[:d], both[d:]
    else:
        dlw = torch.empty(d, dtype=torch.float32, device=x.device)
        dlb = torch.empty(d, dtype=torch.float32, device=x.device)
    check(get_lib().pv_context_bwd_bf16(ptr(dkv16), ptr(w_kv), ptr(x), ptr(x2), d1, period, ptr(mean), ptr(rstd), ptr(ln_w),
                                        ptr(ln_b), ptr(dw), ptr(dlw), ptr(dlb), rows, d, w_kv.shape[0], ptr(ws), nbytes.value,
                                        int(accumulate_kv_into is not None), int(accumulate_ln_into is not None),
                                        current_stream_ptr()), "pv_context_bwd_bf16")
    return dw, dlw, dlb


def softmax_fwd_(x: torch.Tensor, scale: float) -> torch.Tensor:
    """In place: x <- softmax(scale * x) over the last dimension (contiguous)."""
    require_cuda(x)
    n = x.shape[-1]
    check(get_lib().pv_softmax_fwd_f32(ptr(x), ptr(x), x.numel() // n, n, scale, current_stream_ptr()), "pv_softmax_fwd_f32")
    return x


def softmax_bwd_(p: torch.Tensor, dp: torch.Tensor, scale: float) -> torch.Tensor:
    """In place on dp: dp <- scale * p * (dp - sum(dp * p))."""
    require_cuda(p, dp)
    n = p.shape[-1]
    check(get_lib().pv_softmax_bwd_f32(ptr(p), ptr(dp), ptr(dp), p.numel() // n, n, scale, current_stream_ptr()),
          "pv_softmax_bwd_f32")
    return dp


def geglu_fwd(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    h = x.shape[-1] // 2
    y = torch.empty(x.shape[:-1] + (h,), dtype=torch.float32, device=x.device)
    check(get_lib().pv_geglu_fwd_f32(ptr(x), ptr(y), x.numel() // (2 * h), h, current_stream_ptr()), "pv_geglu_fwd_f32")
    return y


def geglu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    require_cuda(x, dy)
    h = x.shape[-1] // 2
    dx = torch.empty_like(x)
    check(get_lib().pv_geglu_bwd_f32(ptr(x), ptr(dy), ptr(dx), x.numel() // (2 * h), h, current_stream_ptr()), "pv_geglu_bwd_f32")
    return dx


def mean_axis1_fwd(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    b, n, d = x.shape
    y = torch.empty((b, d), dtype=torch.float32, device=x.device)
    check(get_lib().pv_mean_axis1_fwd_f32(ptr(x), ptr(y), b, n, d, current_stream_ptr()), "pv_mean_axis1_fwd_f32")
    return y


def mean_axis1_bwd(dy: torch.Tensor, n: int) -> torch.Tensor:
    require_cuda(dy)
    b, d = dy.shape
    dx = torch.empty((b, n, d), dtype=torch.float32, device=dy.device)
    check(get_lib().pv_mean_axis1_bwd_f32(ptr(dy), ptr(dx), b, n, d, current_stream_ptr()), "pv_mean_axis1_bwd_f32")
    return dx


def gru_seq_fwd(gi: torch.Tensor, h0: Optional[torch.Tensor], w_hh: torch.Tensor, b_hh: torch.Tensor):
    require_cuda(gi, h0, w_hh, b_hh)
    b, t, h3 = gi.shape
    h = h3 // 3
    out = torch.empty((b, t, h), dtype=torch.float32, device=gi.device)
    saved = torch.empty((b, t, 4 * h), dtype=torch.float32, device=gi.device)
    check(get_lib().pv_gru_seq_fwd_f32(ptr(gi), ptr(h0), ptr(w_hh), ptr(b_hh), ptr(out), ptr(saved), b, t, h,
                                       current_stream_ptr()), "pv_gru_seq_fwd_f32")
    return out, saved


def gru_seq_bwd(dout, dh_last, h0, out, saved, w_hh, need_dh0: bool):
    require_cuda(dout, dh_last, h0, out, saved, w_hh)
    b, t, h = out.shape
    dgi = torch.empty((b, t, 3 * h), dtype=torch.float32, device=out.device)
    dh0 = torch.empty((b, h), dtype=torch.float32, device=out.device) if need_dh0 else None
    dw = torch.empty((3 * h, h), dtype=torch.float32, device=out.device)
    db = torch.empty((3 * h,), dtype=torch.float32, device=out.device)
    nbytes = b * (3 * h * h + 3 * h) * 4
    ws = _workspace("gru_bwd", nbytes, out.device)
    check(get_lib().pv_gru_seq_bwd_f32(ptr(dout), ptr(dh_last), ptr(h0), ptr(out), ptr(saved), ptr(w_hh), ptr(dgi), ptr(dh0),
                                       ptr(dw), ptr(db), b, t, h, ptr(ws), nbytes, current_stream_ptr()), "pv_gru_seq_bwd_f32")
    return dgi, dh0, dw, db


def attention_desc(q: torch.Tensor, kv: torch.Tensor, heads: int, scale: float) -> "_lib.AttentionDesc":
    b, n_q, inner = q.shape
    return _lib.AttentionDesc(b, heads, n_q, kv.shape[1], inner // heads, q.stride(0), q.stride(1), kv.stride(0), kv.stride(1),
                              scale)


def attention_fwd(q: torch.Tensor, kv: torch.Tensor, heads: int, scale: float, bf16_operands: bool = False):
    """q [b, i, h*64], kv [b, j, 2*h*64] (k | v halves) contiguous -> (out [b, i, h*64], lse [b, h, i]).
    bf16_operands: the two products take bf16 operands on the bf16 matrix cores (f32 accumulation, f32 softmax)."""
    require_cuda(q, kv)
    b, n_q, inner = q.shape
    d = attention_desc(q, kv, heads, scale)
    out = torch.empty_like(q)
    lse = torch.empty((b, heads, n_q), dtype=torch.float32, device=q.device)
    kv16 = kv.dtype == torch.bfloat16      # K / V stored as bf16 (gemm_rows_bf16out): bf16-operand kernels only
    if kv16 and not bf16_operands:
        raise ValueError("attention_fwd: a bfloat16 kv needs bf16_operands=True")
    v_ptr = ctypes.c_void_p(kv.data_ptr() + inner * kv.element_size())
    if bf16_operands:
        n_ws = get_lib().pv_attention_fwd_workspace_floats(ctypes.byref(d))
        ws = _workspace("attention_fwd", n_ws * 4, q.device) if n_ws else None
        fn = get_lib().pv_attention_fwd_bf16kv if kv16 else get_lib().pv_attention_fwd_bf16
        check(fn(ptr(q), ptr(kv), v_ptr, ptr(out), ptr(lse), ctypes.byref(d), ptr(ws), current_stream_ptr()),
              "pv_attention_fwd_bf16kv" if kv16 else "pv_attention_fwd_bf16")
    else:
        check(get_lib().pv_attention_fwd_f32(ptr(q), ptr(kv), v_ptr, ptr(out), ptr(lse), ctypes.byref(d), current_stream_ptr()),
              "pv_attention_fwd_f32")
    return out, lse


def attention_bwd(q, kv, out, dout, lse, heads: int, scale: float, bf16_operands: bool = False, accumulate_dkv_into=None,
                  dkv_bf16: bool = False):
    """Backward of attention_fwd (n_q <= 128): returns (dq like q, dkv like kv).
    accumulate_dkv_into (bf16 operands only): a tensor like kv that receives `+= dkv` in the kernel's store.
    dkv_bf16 (bf16-stored kv only): dkv is STORED as bf16 too -- for a consumer that rounds it to bf16 anyway."""
    require_cuda(q, kv, out, dout, lse, accumulate_dkv_into)
    b, n_q, inner = q.shape
    d = attention_desc(q, kv, heads, scale)
    dq = torch.empty_like(q)
    kv16 = kv.dtype == torch.bfloat16
    if kv16 and not bf16_operands:
        raise ValueError("attention_bwd: a bfloat16 kv needs bf16_operands=True")
    if dkv_bf16:
        if not kv16 or accumulate_dkv_into is not None:
            raise ValueError("attention_bwd: dkv_bf16 needs a bfloat16 kv and no accumulate_dkv_into")
        dkv = torch.empty(kv.shape, dtype=torch.bfloat16, device=kv.device)
        n_ws = get_lib().pv_attention_bwd_workspace_floats(ctypes.byref(d))
        delta = _workspace("attention_bwd", n_ws * 4, q.device)
        check(get_lib().pv_attention_bwd_bf16kv16(ptr(q), ptr(kv), ctypes.c_void_p(kv.data_ptr() + inner * 2), ptr(out), ptr(dout),
                                                  ptr(lse), ptr(delta), ptr(dq), ptr(dkv),
                                                  ctypes.c_void_p(dkv.data_ptr() + inner * 2), ctypes.byref(d),
                                                  current_stream_ptr()), "pv_attention_bwd_bf16kv16")
        return dq, dkv
    if accumulate_dkv_into is not None:
        if (not bf16_operands or accumulate_dkv_into.shape != kv.shape or not accumulate_dkv_into.is_contiguous()
                or accumulate_dkv_into.dtype != torch.float32):
            raise ValueError("attention_bwd: accumulate_dkv_into needs the bf16-operand kernels and a contiguous float32 tensor "
                             "shaped like kv")
        dkv = accumulate_dkv_into
    else:
        dkv = torch.empty(kv.shape, dtype=torch.float32, device=kv.device)
    n_ws = get_lib().pv_attention_bwd_workspace_floats(ctypes.byref(d))
    delta = _workspace("attention_bwd", n_ws * 4, q.device)
    v_ptr = ctypes.c_void_p(kv.data_ptr() + inner * kv.element_size())
    dv_ptr = ctypes.c_void_p(dkv.data_ptr() + inner * 4)
    if bf16_operands:
        fn = get_lib().pv_attention_bwd_bf16kv if kv16 else get_lib().pv_attention_bwd_bf16
        check(fn(ptr(q), ptr(kv), v_ptr, ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dq), ptr(dkv), dv_ptr, ctypes.byref(d),
                 int(accumulate_dkv_into is not None), current_stream_ptr()),
              "pv_attention_bwd_bf16kv" if kv16 else "pv_attention_bwd_bf16")
    else:
        check(get_lib().pv_attention_bwd_f32(ptr(q), ptr(kv), v_ptr, ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dq), ptr(dkv),
                                             dv_ptr, ctypes.byref(d), current_stream_ptr()), "pv_attention_bwd_f32")
    return dq, dkv
