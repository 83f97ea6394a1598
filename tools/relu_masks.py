"""Diagnostic (tests + tools/val_ablation.py only, never the product path): run the bf16 Conv3D tower with GIVEN ReLU masks.

The bf16 path's conv-weight gradients sit 3-11 % (norm-wise) from the f32 arithmetic's; tests/test_oracle_conv.py shows on the CPU
that 0.5 % of that is operand rounding and the rest is ReLU units within bf16 rounding of zero that flip.  The GPU twin needs the
HIP kernels to run with the f32 run's masks: inside `forced_relu_masks(...)` every forward launch of a ReLU layer (the conv
layers' pv_conv3d_fwd_bf16 / pv_conv3d_fwd_bf16_f32in, fc1's pv_linear_fwd_bf16) has its OUTPUT rewritten in place before
autograd saves it:   y <- mask ? max(y, tiny) : 0.   Every later gate of the backward pass (the layer's own (y > 0), the next
layer's dgrad epilogue (x > 0), fc1's one-pass (x > 0)) reads those signs, so no unit can flip with respect to the given masks;
the forward values move by at most the magnitude of a flipped unit (within bf16 rounding of zero)."""
import torch

from predict_pv_yield_amd import hip_ops as K


class forced_relu_masks:
    def __init__(self, conv_masks, fc1_mask=None):
        """conv_masks: bool tensors [B, 32, T', H', W'] (the reference's NCDHW), one per ReLU conv layer, matched to a launch by
        its output shape; fc1_mask: bool [B, N] or None."""
        self.by_shape = {tuple(m.shape[2:]): m.permute(0, 2, 3, 4, 1).contiguous() for m in conv_masks}      # -> NDHWC
        self.fc1_mask = fc1_mask
        self.forced = 0
        self._saved = {}

    def _force(self, y, mask):
        tiny = torch.tensor(2.0 ** -100, dtype=y.dtype, device=y.device)
        y.copy_(torch.where(mask, torch.maximum(y, tiny), torch.zeros((), dtype=y.dtype, device=y.device)))
        self.forced += 1

    def __enter__(self):
        conv, first, lin = K.conv3d_fwd_bf16, K.conv3d_fwd_bf16_f32in, K.linear_fwd_bf16
        self._saved = {"conv3d_fwd_bf16": conv, "conv3d_fwd_bf16_f32in": first, "linear_fwd_bf16": lin}

        def conv_w(x, gate, wp, bias, c_in, c_out, padding, relu=True, y_ncdhw=False, **kw):
            out = conv(x, gate, wp, bias, c_in, c_out, padding, relu, y_ncdhw, **kw)
            if relu:
                y = out[0] if isinstance(out, tuple) else out
                if y_ncdhw:
                    m = self.by_shape.get(tuple(y.shape[2:]))
                    if m is not None:
                        self._force(y, m.permute(0, 4, 1, 2, 3))
                else:
                    m = self.by_shape.get(tuple(y.shape[1:4]))
                    if m is not None:
                        self._force(y, m)
            return out

        def first_w(x, wp, bias, c_out, padding=(0, 0, 0), relu=True, **kw):
            out = first(x, wp, bias, c_out, padding, relu, **kw)
            if relu:
                y = out[0]
                m = self.by_shape.get(tuple(y.shape[1:4]))
                if m is not None:
                    self._force(y, m)
            return out

        def lin_w(x_bf16, w_bf16, bias, relu=False):
            y = lin(x_bf16, w_bf16, bias, relu)
            if relu and self.fc1_mask is not None and tuple(y.shape) == tuple(self.fc1_mask.shape):
                self._force(y, self.fc1_mask)
            return y

        K.conv3d_fwd_bf16, K.conv3d_fwd_bf16_f32in, K.linear_fwd_bf16 = conv_w, first_w, lin_w
        # the 1-bit masks the forwards write in-kernel would carry the UNFORCED signs: inside this context every dgrad gates by
        # the (rewritten) bf16 activation instead -- bit-identical gradients by tests/test_gpu_conv.py's mask tests
        from predict_pv_yield_amd import functional as Fn
        self._use_masks = Fn.USE_RELU_MASKS
        Fn.USE_RELU_MASKS = False
        return self

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            setattr(K, k, v)
        from predict_pv_yield_amd import functional as Fn
        Fn.USE_RELU_MASKS = self._use_masks
        return False


def f32_relu_masks_torch(model, sat):
    """The f32 arithmetic's ReLU masks of the conv layers and of fc1 for `model`'s CURRENT weights, by torch's own f32 operators
    on the device (MIOpen / rocBLAS: a diagnostic tool may use them; the product path never does).  -> (conv masks, fc1 mask)"""
    import torch.nn.functional as F
    from predict_pv_yield_amd.models.conv3d._fc1_layout import reference_layout
    with torch.no_grad():
        out = sat.float()
        masks = []
        for layer in model._conv_layers():
            out = F.relu(F.conv3d(out, layer.weight, layer.bias))
            masks.append(out > 0)
        z = F.linear(out.reshape(out.shape[0], -1), reference_layout(model.fc1.weight), model.fc1.bias)
    return masks, z > 0
