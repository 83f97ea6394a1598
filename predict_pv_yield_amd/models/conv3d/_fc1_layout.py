"""fc1's weight with its input axis in CHANNELS-LAST order (bf16 Conv3D tower only).

The reference flattens the last conv activation [B, C, T, H, W] and multiplies by fc1.weight [N, C*T*H*W]
(predict_pv_yield/models/conv3d/model.py:122-125).  The bf16 tower keeps activations as [B, T, H, W, C]; handing fc1 that
image as it is -- and taking fc1's input gradient back in the same order -- needs the COLUMNS of fc1.weight in (t, h, w, c)
order.  The alternative (rounds 1-4) transposed 64 MB each way per step: the last layer's NCDHW-writing epilogue and a 41 us
repack of dx in front of its dgrad.

The parameter stays a plain contiguous [N, K] float32 tensor (nothing in the optimiser, the data-parallel exchange or the
kernels changes); only the meaning of its column index differs:   stored[n, s*C + c] == reference[n, c*S + s],  S = T*H*W.
What leaves or enters the process is the reference's layout:
  * Module.state_dict() / load_state_dict(): hooks below (checkpoints are interchangeable with the reference's);
  * HipAdam.state_dict() / load_state_dict() / moments(): the two moment tensors of a marked parameter, likewise;
  * reference_layout(param, tensor): for anyone who reads `model.fc1.weight` (or a gradient of it) directly.
The mark is `param._pv_k_channels = C` (0 / absent: reference order).  The MODULE attribute `_fc1_k_channels` is the truth:
copy.deepcopy / pickle / a converting _apply create fresh Parameter objects without Python attributes, so Model re-applies the
mark in __setstate__, _apply, forward() and configure_optimizers() (Model._mark_fc1_layout), and `Model.precision` is
read-only after construction -- the stored column order cannot come apart from the arithmetic that reads it."""
import torch


def to_channels_last(w: torch.Tensor, channels: int) -> torch.Tensor:
    """reference [N, C*S] -> stored [N, S*C]"""
    n, k = w.shape
    return w.reshape(n, channels, k // channels).transpose(1, 2).reshape(n, k)


def to_reference(w: torch.Tensor, channels: int) -> torch.Tensor:
    """stored [N, S*C] -> reference [N, C*S]"""
    n, k = w.shape
    return w.reshape(n, k // channels, channels).transpose(1, 2).reshape(n, k)


def k_channels(param) -> int:
    return int(getattr(param, "_pv_k_channels", 0) or 0)


def reference_layout(param, tensor=None) -> torch.Tensor:
    """`tensor` (default: the parameter itself; or its gradient, an optimiser moment ...: anything with the parameter's
    shape) in the reference's column order."""
    t = param.detach() if tensor is None else tensor
    c = k_channels(param)
    return to_reference(t, c) if c and t.dim() == 2 else t


def _save_hook(module, state_dict, prefix, local_metadata):
    c = getattr(module, "_fc1_k_channels", 0)
    key = prefix + "fc1.weight"
    if c and key in state_dict and state_dict[key].dim() == 2:
        state_dict[key] = to_reference(state_dict[key].detach(), c)


def _load_pre_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    c = getattr(module, "_fc1_k_channels", 0)
    key = prefix + "fc1.weight"
    if c and key in state_dict and state_dict[key].dim() == 2 and state_dict[key].shape[1] % c == 0:
        state_dict[key] = to_channels_last(state_dict[key], c)


def install(module, channels: int) -> None:
    """Marks `module.fc1.weight` as channels-last along K (its VALUES are permuted accordingly: a freshly initialised
    nn.Linear is i.i.d. per element, but permuting keeps `torch.manual_seed(s); Model(...)` equal to the reference's
    initialisation of the same seed) and installs the two state-dict hooks."""
    module._fc1_k_channels = int(channels)
    with torch.no_grad():
        module.fc1.weight.copy_(to_channels_last(module.fc1.weight.detach().clone(), channels))
    module._register_state_dict_hook(_save_hook)
    module._register_load_state_dict_pre_hook(_load_pre_hook, with_module=True)
    mark(module)


def mark(module) -> None:
    c = getattr(module, "_fc1_k_channels", 0)
    if c:
        module.fc1.weight._pv_k_channels = c
