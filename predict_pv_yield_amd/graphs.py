"""HIP-graph capture of a whole train step: forward + loss + backward + optimiser step recorded once on a capture stream
and replayed per batch -- no Python, autograd or launch work between the ~30 kernels of a step.

What makes the step replayable here:
  * every kernel of the path launches on torch's current stream through the C ABI, so a `torch.cuda.graph` capture records
    them like any other work (workspaces that grow during the capture come from the graph's private pool);
  * `HipAdam(capturable=True)` keeps the step counter and the bias-correction scalars in device memory, advanced by a
    one-thread kernel inside the graph (kernel ARGUMENTS are frozen at capture; device memory is not);
  * the batch is copied into static input tensors before each replay; the loss is read from a static output tensor.
The reference has no counterpart (Lightning drives eager PyTorch); the semantics are those of
`opt.zero_grad(); loss = model.training_step(batch, i); loss.backward(); opt.step()` (base_model.py:91-99, 255-257).
"""
from typing import Any, Dict

import torch


def _map_tensors(obj: Any, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    if hasattr(obj, "__dict__") and not isinstance(obj, type):      # attribute containers: data.batch.BatchML and its sections
        import copy
        out = copy.copy(obj)
        for k, v in vars(obj).items():
            setattr(out, k, _map_tensors(v, fn))
        return out
    return obj


# Python-side marks a loader leaves on a batch tensor and the model reads (optical_flow.AdvectingLoader tags the frames it
# has already advected: Model._satellite_input then takes them as they are).  clone() drops Python attributes, and a static
# batch without the mark would be advected a second time inside the captured step.
_TENSOR_MARKS = ("_pv_advected",)


def _clone_keeping_marks(t: torch.Tensor) -> torch.Tensor:
    c = t.clone()
    for name in _TENSOR_MARKS:
        if hasattr(t, name):
            setattr(c, name, getattr(t, name))
    return c


class GraphBatchMismatch(ValueError):
    """The batch handed to a captured step differs (shape, dtype or mark) from the one it was captured for."""


def _copy_into(dst: Any, src: Any) -> None:
    if torch.is_tensor(dst):
        for name in _TENSOR_MARKS:
            if bool(getattr(dst, name, False)) != bool(getattr(src, name, False)):
                raise GraphBatchMismatch(f"GraphedTrainStep: batch tensor mark {name} = {getattr(src, name, False)!r} differs from the "
                                 f"captured step's ({getattr(dst, name, False)!r}): the captured step was recorded for the other form")
        if dst.shape != src.shape or dst.dtype != src.dtype:
            raise GraphBatchMismatch(f"GraphedTrainStep: batch tensor {tuple(src.shape)} {src.dtype} does not match the captured "
                             f"{tuple(dst.shape)} {dst.dtype} (a graph replays fixed shapes)")
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)
    elif hasattr(dst, "__dict__") and not isinstance(dst, type):
        for k, d in vars(dst).items():
            _copy_into(d, getattr(src, k))


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer, example_batch); loss = step(batch) for every batch of the same shapes.

    `optimizer` must be a HipAdam(capturable=True).  Three eager steps run first on a side stream (they ARE training steps:
    allocator warm-up, lazily built operand copies and workspaces), then one step is captured."""

    def __init__(self, model, optimizer, example_batch: Dict, batch_idx: int = 0, warmup: int = 3):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs HipAdam(capturable=True): the by-value Adam step would be frozen at capture")
        self.model, self.optimizer = model, optimizer
        self.static_batch = _map_tensors(example_batch, _clone_keeping_marks)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step(batch_idx)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        if hasattr(optimizer, "freeze_layout"):
            # from here on the optimiser's state tensors are never replaced: the moments of the large matrix keep the layout the
            # eager steps left (tiled after one fused backward) -- a layout conversion must not be RECORDED into the graph either
            optimizer.freeze_layout(True)
        with torch.cuda.graph(self.graph):
            loss = model.training_step(self.static_batch, batch_idx)
            self._backward(loss)
            optimizer.step()
        self.static_loss = loss.detach()
        self.replays = 0
        # kernel ARGUMENTS are frozen in the graph: the scratch buffers the captured kernels write to must stay where they
        # are (hip_ops._workspace refuses to replace them while this graph lives), and the Adam hyper-parameters that
        # pv_adam_scalars_advance received by value cannot change any more
        from . import hip_ops as K
        self._pins = K.pin_workspaces()
        g = optimizer.param_groups[0]
        self._hyper = (g["lr"], tuple(g["betas"]), g["eps"])

    def close(self) -> None:
        """Releases the graph and the workspaces it pinned."""
        from . import hip_ops as K
        K.unpin_workspaces(getattr(self, "_pins", None))
        self._pins = None
        if getattr(self, "optimizer", None) is not None and hasattr(self.optimizer, "freeze_layout"):
            self.optimizer.freeze_layout(False)
        self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    @staticmethod
    def _backward(loss):
        from .functional import unit_gradient
        loss.backward(unit_gradient(loss))

    def _eager_step(self, batch_idx):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.model.training_step(self.static_batch, batch_idx)
        self._backward(loss)
        self.optimizer.step()
        return loss

    def __call__(self, batch: Dict) -> torch.Tensor:
        if self.graph is None:
            raise RuntimeError("GraphedTrainStep: closed")
        g = self.optimizer.param_groups[0]
        if (g["lr"], tuple(g["betas"]), g["eps"]) != self._hyper:
            raise RuntimeError(f"GraphedTrainStep: lr / betas / eps changed after capture ({self._hyper} -> "
                               f"{(g['lr'], tuple(g['betas']), g['eps'])}); they were frozen as kernel arguments of the "
                               f"captured step -- capture a new GraphedTrainStep after changing them")
        _copy_into(self.static_batch, batch)
        self.graph.replay()
        self.replays += 1
        return self.static_loss
