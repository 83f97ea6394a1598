"""Minimal PyTorch-Lightning-compatible surface (pytorch_lightning is not installed in this image).

Covers exactly what the reference drives (SURVEY.md §8b "Module protocol" / "Trainer/CLI"):
  LightningModule : forward / training_step / validation_step / validation_epoch_end / test_step /
                    configure_optimizers, self.log / self.log_dict(on_step, on_epoch, sync_dist),
                    self.current_epoch, self.logger, self.device, self.trainer   (base_model.py:27-257)
  Trainer         : fit / validate / test / predict, callback_metrics, checkpoint_callback,
                    gpus, min/max_epochs, fast_dev_run, resume_from_checkpoint, precision,
                    limit_*_batches, weights_summary, progress_bar_refresh_rate, profiler
                    (configs/trainer/default.yaml, predict_pv_yield/training.py:63-107)
  callbacks       : ModelCheckpoint, EarlyStopping (configs/callbacks/default.yaml)
  loggers         : CSVLogger (configs/logger/csv.yaml)
Under torch.distributed (one process per GPU, RCCL) the Trainer all-reduces gradients after backward
(predict_pv_yield_amd.distributed) and logged metrics with sync_dist=True.
"""
import csv
import logging
import os
import random
import time
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import nn

log = logging.getLogger(__name__)

# checkpoint dialect written by ModelCheckpoint (the reference pins pytorch-lightning 1.4/1.5, requirements.txt)
PL_COMPAT_VERSION = "1.5.10"


def seed_everything(seed: int, workers: bool = False) -> int:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    return seed


def _to_float(v) -> float:
    if isinstance(v, torch.Tensor):
        return float(v.detach().float().mean().item())
    return float(v)


class Callback:
    def on_fit_start(self, trainer, module): ...
    def on_validation_end(self, trainer, module): ...
    def on_train_epoch_end(self, trainer, module): ...
    def on_fit_end(self, trainer, module): ...


class LightningLoggerBase:
    name = "logger"

    def log_metrics(self, metrics: Dict[str, float], step: int): ...
    def log_hyperparams(self, params): ...
    def finalize(self, status: str = "success"): ...

    @property
    def experiment(self):
        return self


class CSVLogger(LightningLoggerBase):
    """pytorch_lightning.loggers.csv_logs.CSVLogger(save_dir, name)."""

    def __init__(self, save_dir: str = ".", name: str = "csv/", version: Optional[str] = None, prefix: str = ""):
        self.save_dir, self.name, self.version = save_dir, name, version or "version_0"
        self.rows: List[Dict[str, Any]] = []
        self.hparams: Dict[str, Any] = {}

    @property
    def log_dir(self):
        return os.path.join(self.save_dir, self.name, self.version)

    def log_metrics(self, metrics, step):
        self.rows.append({"step": step, **metrics})

    def log_hyperparams(self, params):
        self.hparams.update(dict(params))

    def finalize(self, status="success"):
        if not self.rows:
            return
        os.makedirs(self.log_dir, exist_ok=True)
        keys = sorted({k for r in self.rows for k in r})
        with open(os.path.join(self.log_dir, "metrics.csv"), "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=keys)
            wr.writeheader()
            wr.writerows(self.rows)


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer: Optional["Trainer"] = None
        self._current_epoch = 0
        self._logged: Dict[str, float] = {}

    # -- attributes Lightning provides -------------------------------------------------------
    @property
    def current_epoch(self) -> int:
        return self.trainer.current_epoch if self.trainer is not None else self._current_epoch

    @property
    def global_step(self) -> int:
        return self.trainer.global_step if self.trainer is not None else 0

    @property
    def logger(self):
        return self.trainer.logger if self.trainer is not None else None

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")

    # -- logging -----------------------------------------------------------------------------
    def log(self, name, value, on_step=None, on_epoch=None, sync_dist=False, **kwargs):
        self.log_dict({name: value}, on_step=on_step, on_epoch=on_epoch, sync_dist=sync_dist)

    def log_dict(self, dictionary, on_step=None, on_epoch=None, sync_dist=False, **kwargs):
        if self.trainer is not None:
            self.trainer._record(dictionary, on_step=on_step, on_epoch=on_epoch, sync_dist=sync_dist)
        else:  # direct call outside a Trainer (tests/models/baseline/test_baseline_model_gsp.py:41-58)
            self._logged.update({k: v for k, v in dictionary.items()})

    # -- hooks (overridden by models) -----------------------------------------------------------
    def training_step(self, batch, batch_idx): raise NotImplementedError
    def validation_step(self, batch, batch_idx): ...
    def test_step(self, batch, batch_idx): ...
    def validation_epoch_end(self, outputs): ...
    def predict_step(self, batch, batch_idx): return self(batch)
    def configure_optimizers(self): raise NotImplementedError


class LightningDataModule:
    def prepare_data(self): ...
    def setup(self, stage=None): ...
    def train_dataloader(self): raise NotImplementedError
    def val_dataloader(self): return None
    def test_dataloader(self): return None


class ModelCheckpoint(Callback):
    def __init__(self, monitor=None, save_top_k=1, save_last=False, mode="min", dirpath="checkpoints/",
                 filename="epoch_{epoch:03d}", **kwargs):
        self.monitor, self.save_top_k, self.save_last, self.mode = monitor, save_top_k, save_last, mode
        self.dirpath, self.filename = dirpath, filename
        self.best_model_path, self.best_model_score, self.last_model_path = "", None, ""

    def _save(self, trainer, module, path):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        for o in trainer.optimizers:       # collective: every rank reaches _save
            if hasattr(o, "consolidate_sharded"):
                o.consolidate_sharded()
        if trainer.is_global_zero:
            # the keys pytorch_lightning 1.4/1.5 writes and its resume path indexes without a default
            # (`lr_schedulers`, `callbacks`); `epoch` follows PL's convention: the NEXT epoch to run, restored as-is
            torch.save({"state_dict": module.state_dict(), "epoch": trainer.current_epoch + 1,
                        "global_step": trainer.global_step,
                        "pytorch-lightning_version": PL_COMPAT_VERSION,
                        "callbacks": {}, "lr_schedulers": [],
                        "hyper_parameters": dict(getattr(module, "hparams", {}) or {}),
                        "optimizer_states": [o.state_dict() for o in trainer.optimizers]}, path)

    def on_validation_end(self, trainer, module):
        if trainer.fast_dev_run or trainer.sanity_checking:
            return
        name = self.filename.format(epoch=trainer.current_epoch) + ".ckpt"
        score = trainer.callback_metrics.get(self.monitor) if self.monitor else None
        better = score is not None and (self.best_model_score is None or
                                        (score < self.best_model_score if self.mode == "min" else score > self.best_model_score))
        if self.save_top_k != 0 and (better or self.monitor is None):
            if self.best_model_path and self.save_top_k == 1 and os.path.exists(self.best_model_path) and trainer.is_global_zero:
                os.remove(self.best_model_path)
            self.best_model_path = os.path.join(self.dirpath, name)
            self.best_model_score = score
            self._save(trainer, module, self.best_model_path)
        if self.save_last:
            self.last_model_path = os.path.join(self.dirpath, "last.ckpt")
            self._save(trainer, module, self.last_model_path)


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", patience=3, mode="min", min_delta=0.0, **kwargs):
        self.monitor, self.patience, self.mode, self.min_delta = monitor, patience, mode, min_delta
        self.best, self.wait = None, 0

    def on_validation_end(self, trainer, module):
        if trainer.sanity_checking:
            return
        score = trainer.callback_metrics.get(self.monitor)
        if score is None:
            return
        improved = self.best is None or (score < self.best - self.min_delta if self.mode == "min"
                                         else score > self.best + self.min_delta)
        if improved:
            self.best, self.wait = score, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                trainer.should_stop = True


def _raw_satellite(batch) -> bool:
    """True for a whole batch whose satellite imagery is still raw int16 counts (data/netcdf_dataset.py)."""
    try:
        sat = batch["satellite"]["data"] if isinstance(batch, dict) else batch.satellite.data
    except (KeyError, AttributeError, TypeError):
        return False
    return isinstance(sat, torch.Tensor) and sat.dtype == torch.int16


def _move(batch, device):
    moved = _move_only(batch, device)
    if torch.device(device).type == "cuda" and _raw_satellite(moved):
        # the int16 counts crossed PCIe; normalise on the device (netcdf_dataset.py:96-101 semantics)
        from .data.netcdf_dataset import normalise_satellite_on_device
        moved = normalise_satellite_on_device(moved)
    return moved


def _move_only(batch, device):
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True)
    if hasattr(batch, "to") and not isinstance(batch, (str, bytes)):
        return batch.to(device)
    if isinstance(batch, dict):
        return {k: _move_only(v, device) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(_move_only(v, device) for v in batch)
    return batch


class Trainer:
    def __init__(self, gpus=0, min_epochs=1, max_epochs=1000, fast_dev_run=False, callbacks=None, logger=None,
                 resume_from_checkpoint=None, precision=32, weights_summary=None, progress_bar_refresh_rate=None,
                 profiler=None, limit_train_batches=1.0, limit_val_batches=1.0, limit_test_batches=1.0,
                 num_sanity_val_steps=0, terminate_on_nan=False, accelerator=None, max_steps=None,
                 default_root_dir=None, log_every_n_steps=50, advect_on_side_stream=False, hip_graph=False,
                 large_grad_mode=None, **unused):
        self.gpus, self.min_epochs, self.max_epochs = gpus, min_epochs, max_epochs
        # hip_graph (new, opt-in; also PV_TRAINER_HIP_GRAPH=1): single-process fits with ONE HipAdam optimiser replay the train
        # step as a HIP graph (graphs.GraphedTrainStep) after three eager steps -- same losses, parameters and logged values
        # as the eager loop, without the Python / autograd / launch work of the hundreds of launches of a Perceiver step
        self.hip_graph = bool(hip_graph) or bool(os.environ.get("PV_TRAINER_HIP_GRAPH"))
        # large_grad_mode (new, world_size > 1; also PV_GRAD_SYNC): how fc1's gradient is exchanged -- "sharded" (default:
        # reduce-scatter over rows + all-gather of the operand copy), "ksharded" (fc1's columns dealt over the ranks, activations
        # exchanged instead of gradients: distributed.py), "bf16" / "autograd" (bf16 / f32 all-reduce, what Lightning DDP does)
        self.large_grad_mode = large_grad_mode or os.environ.get("PV_GRAD_SYNC") or "sharded"
        if self.large_grad_mode not in ("sharded", "ksharded", "bf16", "autograd"):
            raise ValueError("Trainer(large_grad_mode=...) must be 'sharded', 'ksharded', 'bf16' or 'autograd'")
        self._graph_step = None
        # config 3 only (Model(future_frames="optical_flow") fed raw int16 counts): advect in the loader wrapper
        # (optical_flow.AdvectingLoader) instead of inside the model's forward; same batches, bit for bit.  The name is
        # historical: rounds 2-3 ran the wrapper's advection on a side HIP stream, which round 4 removed (it bought
        # 3 % at best and cost 50 % on one device; optical_flow.AdvectingLoader's docstring)
        self.advect_on_side_stream = bool(advect_on_side_stream)
        self.fast_dev_run = bool(fast_dev_run)
        self.callbacks: List[Callback] = list(callbacks or [])
        if isinstance(logger, (list, tuple)):
            self.loggers = list(logger)
        elif logger in (None, True, False):
            self.loggers = []
        else:
            self.loggers = [logger]
        self.resume_from_checkpoint = resume_from_checkpoint
        self.precision, self.profiler = precision, profiler
        self.limit = {"train": limit_train_batches, "val": limit_val_batches, "test": limit_test_batches}
        self.max_steps = max_steps
        self.terminate_on_nan = terminate_on_nan
        self.current_epoch, self.global_step = 0, 0
        self._callback_metrics: Dict[str, float] = {}
        self._pending_logs: List[tuple] = []
        self.log_every_n_steps = max(1, int(log_every_n_steps))
        self.should_stop = False
        self.sanity_checking = False
        self.optimizers: List[torch.optim.Optimizer] = []
        self.model: Optional[LightningModule] = None
        self.datamodule = None
        self._epoch_acc: Dict[str, List[float]] = {}
        self._profile: Dict[str, float] = {}
        if not any(isinstance(c, ModelCheckpoint) for c in self.callbacks):
            self.callbacks.append(ModelCheckpoint(save_top_k=0))

    # -- properties the reference reads (training.py:101-107, utils.py:172-177) -------------
    @property
    def logger(self):
        if not self.loggers:
            return None
        return self.loggers[0] if len(self.loggers) == 1 else _LoggerCollection(self.loggers)

    @property
    def checkpoint_callback(self):
        return next(c for c in self.callbacks if isinstance(c, ModelCheckpoint))

    @property
    def world_size(self):
        return torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1

    @property
    def is_global_zero(self):
        return not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0

    # -- metric plumbing ------------------------------------------------------------------------
    # Logged values stay DEVICE tensors until they are needed as floats: `_record` stacks one log_dict call into one
    # small f64 vector (and, for sync_dist, enqueues ONE all-reduce of that vector on the device -- a collective at a
    # fixed program point on every rank, no host wait); `_drain` turns the backlog into floats with one `.tolist()` per
    # entry.  Drained every `log_every_n_steps` train steps, at every epoch flush and whenever `callback_metrics` is
    # read -- so the train step itself never blocks the host between forward and backward.
    @property
    def callback_metrics(self) -> Dict[str, float]:
        self._drain()
        return self._callback_metrics

    def _record(self, dictionary, on_step=None, on_epoch=None, sync_dist=False):
        keys = list(dictionary)
        vals = [dictionary[k] for k in keys]
        tensors = [v for v in vals if isinstance(v, torch.Tensor)]
        dev = next((v.device for v in tensors if v.is_cuda), torch.device("cpu"))
        if tensors:
            vec = torch.stack([(v.detach().to(dtype=torch.float64).mean().to(dev) if isinstance(v, torch.Tensor)
                                else torch.tensor(float(v), dtype=torch.float64, device=dev)) for v in vals])
        else:
            vec = torch.tensor([float(v) for v in vals], dtype=torch.float64)
        if sync_dist and self.world_size > 1:
            if torch.distributed.get_backend() != "nccl":
                vec = vec.cpu()
            elif not vec.is_cuda:
                vec = vec.to(self.device)
            torch.distributed.all_reduce(vec, op=torch.distributed.ReduceOp.SUM)
            vec = vec / self.world_size
        self._pending_logs.append((keys, vec, on_step, on_epoch, self.global_step))

    def _drain(self):
        if not self._pending_logs:
            return
        pending, self._pending_logs = self._pending_logs, []
        for keys, vec, on_step, on_epoch, step in pending:
            step_metrics = {}
            for k, v in zip(keys, vec.tolist()):
                if on_step is not False:
                    self._callback_metrics[f"{k}_step" if on_epoch else k] = v
                    step_metrics[f"{k}_step" if on_epoch else k] = v
                if on_epoch:
                    self._epoch_acc.setdefault(k, []).append(v)
            if step_metrics and self.is_global_zero:
                for lg in self.loggers:
                    lg.log_metrics(step_metrics, step)

    def _flush_epoch(self):
        self._drain()
        out = {}
        for k, vs in self._epoch_acc.items():
            if vs:
                out[f"{k}_epoch"] = float(np.mean(vs))
                out[k] = out[f"{k}_epoch"]
        self._callback_metrics.update(out)
        if out and self.is_global_zero:
            for lg in self.loggers:
                lg.log_metrics({k: v for k, v in out.items() if k.endswith("_epoch")}, self.global_step)
        self._epoch_acc = {}

    # -- loops ---------------------------------------------------------------------------------------
    def _device(self):
        if self.gpus not in (0, None, "0", []) and torch.cuda.is_available():
            from .distributed import local_device_index
            return torch.device("cuda", local_device_index())
        return torch.device("cpu")

    def _limit(self, which, loader):
        if self.fast_dev_run:
            return 1
        lim = self.limit[which]
        try:
            n = len(loader)
        except TypeError:
            n = None
        if isinstance(lim, float) and lim <= 1.0:
            return None if n is None or lim == 1.0 else max(1, int(n * lim))
        return int(lim)

    def _attach(self, model, datamodule=None):
        # `python -m torch.distributed.run --nproc-per-node N run.py ...`: join the process group from
        # torchrun's environment HERE, in this process, before the module touches the GPU.  Without it every rank
        # would see world_size 1 and train the whole dataset on its own, racing on the same checkpoint files.
        from .distributed import init_from_env
        init_from_env()
        self.model = model
        model.trainer = self
        if datamodule is not None:
            self.datamodule = datamodule
        self.device = self._device()   # batches go here even when the module has no parameters (baseline model)
        model.to(self.device)

    def _loader(self, which, dataloaders=None):
        if dataloaders is not None:
            return dataloaders
        dm = self.datamodule
        if dm is None:
            return None
        return {"train": dm.train_dataloader, "val": dm.val_dataloader, "test": dm.test_dataloader}[which]()

    def _to_device(self, batch, model):
        """Batch onto the device; raw int16 satellite counts are normalised there (f-3) -- unless the model advects future
        frames itself (config 3: the optical-flow pipeline works on the COUNTS and normalises inside)."""
        if getattr(model, "future_frames", None) == "optical_flow":
            return _move_only(batch, self.device)
        return _move(batch, self.device)

    def _train_batches(self, loader, model, lim):
        """The epoch's batches on the device; with advect_on_side_stream (and a model / batch pair of config 3) the advection
        pipeline runs in the loader wrapper (optical_flow.AdvectingLoader), on the training stream."""
        def moved():
            for i, batch in enumerate(loader):
                if lim is not None and i >= lim:
                    break
                yield self._to_device(batch, model)

        it = moved()
        if not (self.advect_on_side_stream and getattr(model, "future_frames", None) == "optical_flow"):
            return it
        first = next(it, None)
        if first is None:
            return iter(())
        import itertools
        rest = itertools.chain([first], it)
        sat = first.get("satellite") if isinstance(first, dict) else None
        if not (isinstance(sat, dict) and torch.is_tensor(sat.get("data")) and sat["data"].dtype == torch.int16 and sat["data"].is_cuda):
            return rest
        from .optical_flow import AdvectingLoader
        return iter(AdvectingLoader(rest, n_future=model.forecast_len_5))

    # -- opt-in HIP-graph replay of the train step (hip_graph=True) ---------------------------------------------------------
    GRAPH_EAGER_STEPS = 3      # ordinary steps before the capture: workspaces, optimiser state and allocator pools exist by then

    def _enter_graph_mode(self, model) -> bool:
        """Swaps the model's HipAdam for a capturable twin (step counter and bias corrections in device memory) carrying the
        same state.  False (eager loop) for anything but one HipAdam on one process."""
        self._graph_step, self._graph_logs, self._graph_eager, self._graph_side = None, [], 0, None
        if not self.hip_graph:
            return False
        from .optim import HipAdam
        if self.world_size > 1 or len(self.optimizers) != 1 or not isinstance(self.optimizers[0], HipAdam):
            return False
        old = self.optimizers[0]
        if not old.capturable:
            if len(old.param_groups) != 1:
                return False
            if old.overlap_large_update or old.large_grad_mode not in ("fused", "autograd"):
                return False      # a side-stream update or an exchanged gradient cannot be captured: eager loop
            g = old.param_groups[0]
            # the twin steps exactly like the optimiser it replaces: same treatment of the large matrix, same gradient scale
            twin = HipAdam(g["params"], lr=g["lr"], betas=g["betas"], eps=g["eps"], capturable=True,
                           fuse_large_linear=old.large_grad_mode == "fused")
            twin.grad_scale = old.grad_scale
            if len(old.state):
                twin.load_state_dict(old.state_dict())
            self.optimizers = [twin]
        return True

    def _graph_train_step(self, model, batch, batch_idx) -> bool:
        """One train step in graph mode: True if it ran here (captured replay), False if the caller's eager path must run it
        (the first steps, or a batch whose shapes differ from the captured one)."""
        from .graphs import GraphBatchMismatch, GraphedTrainStep
        if self._graph_step is None:
            if self._graph_eager < self.GRAPH_EAGER_STEPS:
                # ordinary steps, but on a side stream: the parameters' gradient-accumulation nodes are created by the first
                # backward and remember its stream -- on the default stream they would make the capture wait for it, which a
                # capturing stream must not do
                self._graph_eager += 1
                if self._graph_side is None:
                    self._graph_side = torch.cuda.Stream()
                cur = torch.cuda.current_stream()
                self._graph_side.wait_stream(cur)
                with torch.cuda.stream(self._graph_side):
                    opt = self.optimizers[0]
                    opt.zero_grad(set_to_none=True)
                    loss = self._timed("training_step", model.training_step, batch, batch_idx)
                    # as in the eager loop: a fused optimiser steps fc1 inside backward, so the loss is tested BEFORE it
                    if self.terminate_on_nan and not torch.isfinite(loss.detach()).all():
                        raise ValueError("loss is NaN or inf")
                    self._timed("backward", self._backward, loss)
                    self._timed("optimizer_step", opt.step)
                cur.wait_stream(self._graph_side)
                return True
            n0 = len(self._pending_logs)
            self._graph_step = GraphedTrainStep(model, self.optimizers[0], batch, batch_idx=batch_idx, warmup=0)
            # what training_step logged during the capture are the graph's own output tensors: each replay refreshes them
            self._graph_logs = self._pending_logs[n0:]
            del self._pending_logs[n0:]
        try:
            loss = self._timed("training_step", self._graph_step, batch)
        except GraphBatchMismatch:
            # other shapes (a ragged last batch): this one runs eagerly.  The graph is released first: while it lives the
            # optimiser's layout is frozen, and an eager step of a shape the one-pass fc1 backward does not take would have to
            # convert the tiled moments in place (HipAdam._moments_rows raises); the next matching batch captures again
            self._leave_graph_mode()
            return False
        for keys, vec, on_step, on_epoch, _ in self._graph_logs:
            self._pending_logs.append((keys, vec.clone(), on_step, on_epoch, self.global_step))
        if self.terminate_on_nan and not torch.isfinite(loss).all():
            raise ValueError("loss is NaN or inf")
        return True

    def _leave_graph_mode(self) -> None:
        if self._graph_step is not None:
            self._graph_step.close()
            self._graph_step = None

    @staticmethod
    def _backward(loss):
        """loss.backward() with a cached unit root gradient on the device (no per-step fill / multiply launches)."""
        if loss.is_cuda and loss.dim() == 0:
            from .functional import unit_gradient
            loss.backward(unit_gradient(loss))
        else:
            loss.backward()

    def _timed(self, key, fn, *a, **kw):
        if self.profiler is None:
            return fn(*a, **kw)
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        self._profile[key] = self._profile.get(key, 0.0) + time.perf_counter() - t0
        return out

    def _eval_loop(self, which, loader, step_fn, epoch_end_fn=None):
        if loader is None:
            return
        model = self.model
        model.eval()
        outputs = []
        lim = self._limit(which if which != "validate" else "val", loader)
        with torch.no_grad():
            for i, batch in enumerate(loader):
                if lim is not None and i >= lim:
                    break
                outputs.append(self._timed(f"{which}_step", step_fn, self._to_device(batch, model), i))
        if epoch_end_fn is not None:
            epoch_end_fn(outputs)
        self._flush_epoch()

    def fit(self, model, train_dataloaders=None, val_dataloaders=None, datamodule=None, train_dataloader=None):
        if isinstance(train_dataloaders, LightningDataModule) or (train_dataloaders is not None and hasattr(train_dataloaders, "train_dataloader")):
            datamodule, train_dataloaders = train_dataloaders, None
        if train_dataloader is not None:
            train_dataloaders = train_dataloader
        self._attach(model, datamodule)
        if datamodule is not None:
            datamodule.prepare_data()
            datamodule.setup("fit")
        opt = model.configure_optimizers()
        if isinstance(opt, dict):
            opt = opt["optimizer"]
        self.optimizers = list(opt) if isinstance(opt, (list, tuple)) else [opt]
        if self.resume_from_checkpoint:
            ckpt = torch.load(self.resume_from_checkpoint, map_location="cpu")
            model.load_state_dict(ckpt["state_dict"])
            for o, s in zip(self.optimizers, ckpt.get("optimizer_states", [])):
                o.load_state_dict(s)
            self.current_epoch = ckpt.get("epoch", 0)      # PL convention: the stored value is the next epoch to run
            self.global_step = ckpt.get("global_step", 0)
        if self.world_size > 1:
            from .distributed import broadcast_parameters
            broadcast_parameters(model)
            for o in self.optimizers:   # summing all-reduce; HipAdam folds the 1/world_size into its update
                if hasattr(o, "grad_scale"):
                    o.grad_scale = 1.0 / self.world_size
                if hasattr(o, "set_large_grad_mode"):
                    # fc1's gradient travels in bf16 and is reduce-scattered: each rank steps its own rows of the
                    # matrix and the bf16 operand copy is all-gathered (falls back to a bf16 all-reduce when the rows
                    # do not divide over the ranks)
                    o.set_large_grad_mode(self.large_grad_mode)
            # the in-backward exchange leaves SUMS; only optimisers that fold 1/world into their update (HipAdam) can
            # consume them.  Any other optimiser gets the plain averaged all-reduce after backward (no hooks: a hook
            # would have summed the large gradients already and they would be reduced twice)
            self._fold_mean = all(hasattr(o, "grad_scale") for o in self.optimizers)
            if self._fold_mean:
                from .distributed import OverlappedGradSync
                self._grad_sync = OverlappedGradSync(model)
            else:
                self._grad_sync = None
                for o in self.optimizers:
                    if hasattr(o, "set_large_grad_mode"):
                        o.set_large_grad_mode("autograd")
        graph_mode = self._enter_graph_mode(model)
        for cb in self.callbacks:
            cb.on_fit_start(self, model)
        max_epochs = 1 if self.fast_dev_run else self.max_epochs
        while self.current_epoch < max_epochs and not self.should_stop:
            model.train()
            loader = self._loader("train", train_dataloaders)
            lim = self._limit("train", loader)
            for i, batch in enumerate(self._train_batches(loader, model, lim)):
                if graph_mode and self._graph_train_step(model, batch, i):
                    self.global_step += 1
                    if self.global_step % self.log_every_n_steps == 0:
                        self._drain()
                    if self.max_steps and self.global_step >= self.max_steps:
                        self.should_stop = True
                        break
                    continue
                for o in self.optimizers:
                    o.zero_grad(set_to_none=True)
                loss = self._timed("training_step", model.training_step, batch, i)
                # terminate_on_nan: with a fused optimiser (HipAdam "fused": fc1 is stepped inside backward) the loss is
                # tested BEFORE backward, so a non-finite step leaves weights and moments untouched (one host wait, only
                # in this opt-in mode); otherwise after backward has been queued
                nan_first = self.terminate_on_nan and any(getattr(o, "large_grad_mode", None) == "fused"
                                                          for o in self.optimizers)
                if nan_first and not torch.isfinite(loss.detach()).all():
                    raise ValueError("loss is NaN or inf")
                self._timed("backward", self._backward, loss)
                if self.terminate_on_nan and not nan_first and not torch.isfinite(loss.detach()).all():
                    raise ValueError("loss is NaN or inf")
                if self.world_size > 1:
                    from .distributed import all_reduce_gradients
                    if self._grad_sync is not None:
                        self._timed("grad_all_reduce", self._grad_sync.finish)   # large grads were launched in backward
                    else:
                        self._timed("grad_all_reduce", all_reduce_gradients, model, True)
                for o in self.optimizers:
                    self._timed("optimizer_step", o.step)
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0:
                    self._drain()
                if self.max_steps and self.global_step >= self.max_steps:
                    self.should_stop = True
                    break
            self._flush_epoch()
            for cb in self.callbacks:
                cb.on_train_epoch_end(self, model)
            vloader = self._loader("val", val_dataloaders)
            if vloader is not None:
                self._eval_loop("val", vloader, model.validation_step, model.validation_epoch_end)
                for cb in self.callbacks:
                    cb.on_validation_end(self, model)
            self.current_epoch += 1
            if self.current_epoch < self.min_epochs:
                self.should_stop = False
        self._leave_graph_mode()
        for o in self.optimizers:          # sharded large-parameter update: make every rank's f32 copy complete
            if hasattr(o, "consolidate_sharded"):
                o.consolidate_sharded()
        if getattr(self, "_grad_sync", None) is not None:
            self._grad_sync.remove()
            self._grad_sync = None
        for cb in self.callbacks:
            cb.on_fit_end(self, model)
        self._finish()

    def validate(self, model=None, dataloaders=None, datamodule=None, val_dataloaders=None):
        model = model or self.model
        self._attach(model, datamodule)
        if datamodule is not None:
            datamodule.prepare_data()
            datamodule.setup("validate")
        loader = self._loader("val", dataloaders if dataloaders is not None else val_dataloaders)
        self._eval_loop("val", loader, model.validation_step, model.validation_epoch_end)
        self._finish()
        return [dict(self.callback_metrics)]

    def test(self, model=None, dataloaders=None, datamodule=None, test_dataloaders=None):
        model = model or self.model
        self._attach(model, datamodule)
        loader = self._loader("test", dataloaders if dataloaders is not None else test_dataloaders)
        self._eval_loop("test", loader, model.test_step)
        self._finish()
        return [dict(self.callback_metrics)]

    def predict(self, model=None, dataloaders=None, datamodule=None):
        model = model or self.model
        self._attach(model, datamodule)
        loader = self._loader("test", dataloaders)
        model.eval()
        outs = []
        with torch.no_grad():
            for i, batch in enumerate(loader):
                if self.fast_dev_run and i >= 1:
                    break
                outs.append(model.predict_step(self._to_device(batch, model), i))
        return outs

    def _finish(self):
        if self.is_global_zero:
            for lg in self.loggers:
                lg.finalize("success")
        if self.profiler is not None and self._profile and self.is_global_zero:
            log.info("profiler (simple): " + ", ".join(f"{k}={v:.3f}s" for k, v in sorted(self._profile.items())))


class _LoggerCollection(LightningLoggerBase):
    def __init__(self, loggers):
        self._loggers = loggers

    def __getitem__(self, i):
        return self._loggers[i]

    @property
    def experiment(self):
        return [lg.experiment for lg in self._loggers]

    def log_metrics(self, metrics, step):
        for lg in self._loggers:
            lg.log_metrics(metrics, step)

    def log_hyperparams(self, params):
        for lg in self._loggers:
            lg.log_hyperparams(params)
