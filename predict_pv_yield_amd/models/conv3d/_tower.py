"""One Conv3D tower + its first fully-connected layer, shared by model.py (padding 0) and model_sat_nwp.py
(padding (1,0,0), two towers).  Every FLOP runs in the gfx950 kernels behind include/pv_yield_hip.h."""
import logging

import torch

from ... import functional as Fn

_LOG = logging.getLogger(__name__)
_WARNED = set()


def bf16_tower_supported(c_in: int, channels: int, flat_features: int) -> bool:
    """The MFMA path pads channels to 16 or 32 and streams fc1 in 8-element (16-byte) groups.  Asked only by models built
    with precision="bf16": a shape outside it falls to the exact-f32 kernels (several times slower: the f32 matrix cores
    run at 1/16 of the bf16 rate, and shapes they do not cover take register-tiled FMA kernels) -- said once per shape."""
    ok = 16 < channels <= 32 and c_in <= 32 and flat_features % 8 == 0
    if not ok and (c_in, channels, flat_features) not in _WARNED:
        _WARNED.add((c_in, channels, flat_features))
        _LOG.warning("predict_pv_yield_amd: precision='bf16' was requested, but a Conv3D tower with %d input channels, %d conv "
                     "channels and %d flattened features is outside the bf16 matrix-core kernels (conv channels 17..32, input "
                     "channels <= 32, features %% 8 == 0): it runs on the exact-f32 kernels, expect a several times slower step",
                     c_in, channels, flat_features)
    return ok


def conv_tower_fc1(data: torch.Tensor, convs, fc1, c_in: int, channels: int, padding, flat_features: int,
                   use_bf16: bool, fc1_channels_last: bool = False) -> torch.Tensor:
    """relu(fc1(flatten_NCDHW(relu(conv_n(... relu(conv_0(data))))))) — model.py:117-125 / model_sat_nwp.py:188-197,236-246.
    fc1_channels_last (bf16 tower): fc1.weight's columns are stored in (t, h, w, c) order (_fc1_layout.py), so the last layer
    writes NDHWC like every other and fc1's input gradient returns in the order its dgrad reads."""
    if fc1_channels_last and not use_bf16:
        raise RuntimeError("fc1.weight is stored channels-last (bf16 tower, models/conv3d/_fc1_layout.py) but the f32 tower was "
                           "asked for: it flattens NCDHW and would multiply by column-permuted weights")
    if not data.is_cuda:
        raise RuntimeError("predict_pv_yield_amd Conv3D model runs on the MI355X only: move the module and the "
                           "batch to cuda (there is no CPU fallback)")
    batch_size = data.shape[0]
    if use_bf16:
        from ...hip_ops import bf16_cpad
        fused_first = (len(convs) > 1 and data.dtype == torch.float32 and bf16_cpad(c_in) == 16 and padding[2] == 0
                       and not data.requires_grad)
        # training: every layer but the last also emits the 1-bit relu mask of its output; the next layer's dgrad gates
        # dx with it (4 bytes per voxel instead of re-reading the 64-byte bf16 activation)
        masks = Fn.USE_RELU_MASKS and torch.is_grad_enabled() and any(p.requires_grad for c in convs for p in c.parameters())
        mask = None
        if fused_first:
            # the first layer reads the f32 NCDHW input itself and leaves the NDHWC bf16 image for its wgrad behind
            out = Fn.conv3d_first_layer_bf16(data, convs[0].weight, convs[0].bias, tuple(padding), relu=True, dy_pregated=True,
                                             want_relu_mask=masks)
            if masks:
                out, mask = out
            c_in = channels
        else:
            out = Fn.PackInputBF16.apply(data)
        for i, layer in enumerate(convs):
            if fused_first and i == 0:
                continue
            last = i == len(convs) - 1
            # layers >= 1 consume a ReLU output and gate their own dx; every layer but the last is followed by one; the last
            # one's gradient comes from fc1, whose one-pass backward gates it too when it runs ("ask": functional._take_pregated)
            out = Fn.conv3d_relu_bf16(out, layer.weight, layer.bias, c_in, tuple(padding), relu=True,
                                      y_ncdhw=last and not fc1_channels_last, x_is_relu_output=i > 0,
                                      dy_pregated=(not last) or ("ask" if fc1_channels_last else False),
                                      x_relu_mask=mask if i > 0 else None, want_relu_mask=masks and not last)
            mask = None
            if masks and not last:
                out, mask = out
            c_in = channels
        if fc1_channels_last and channels != 32:
            raise RuntimeError("fc1_channels_last needs 32 conv channels (the NDHWC image has no padding channels then)")
        out = out.reshape(batch_size, flat_features)  # NCDHW flatten order, or (t, h, w, c) with fc1_channels_last
        return Fn.linear_bf16(out, fc1.weight, fc1.bias, relu=True, x_is_relu_output=True)   # out = relu(last conv)
    # precision="fp32": f32-accurate arithmetic on the reference layout (exact f32 products forward / dgrad; split 16-bit products
    # for weight gradients and fc1 unless PV_EXACT_F32=1, functional.exact_f32()).  32-channel 3x3x3 layers run on the f32 matrix cores
    # (v_mfma_f32_32x32x2_f32: forward / dgrad conv3d_fwd_mfma_f32<3,3,3>, weight gradient conv3d_wgrad_mfma_f32<3,3,3>),
    # the first layer and odd shapes on the register-tiled FMA kernels; each dgrad gates dx with its producer's ReLU
    out = data.float() if data.dtype != torch.float32 else data
    n = len(convs)
    # every layer receives an already gated gradient: from the next layer's dgrad epilogue, the last one from a streaming
    # gate of fc1's input gradient (sizes that are no multiple of 4: the last layer gates while it stages, on the tiled kernel)
    gate_last = (batch_size * flat_features) % 4 == 0
    for i, layer in enumerate(convs):
        # a layer whose consumer is another layer of the half-float form hands over its operand images, not a float32 tensor
        chain = False
        if i + 1 < n:
            if Fn.is_operand_images(out):
                geo = (out.shape[1], 32) + tuple(out.shape[2:5])
            else:
                geo = (out.shape[0], out.shape[1]) + tuple(out.shape[2:])
            if Fn.conv_f16x2_takes(*geo, layer.weight, (1, 1, 1), padding, out.requires_grad):
                nxt = (geo[0], 32) + tuple(d + 2 * q - 2 for d, q in zip(geo[2:], Fn._triple(padding)))
                chain = Fn.conv_f16x2_takes(*nxt, convs[i + 1].weight, (1, 1, 1), padding)
        out = Fn.conv3d_general_f32(out, layer.weight, layer.bias, stride=(1, 1, 1), padding=tuple(padding), relu=True,
                                    x_is_relu_output=i > 0, dy_pregated=(i + 1 < n) or gate_last, chain_out=chain)
    if gate_last:
        out = Fn.relu_gate_f32(out)
    out = out.reshape(batch_size, flat_features)
    return Fn.linear_f32(out, fc1.weight, fc1.bias, relu=True)
