"""Mirror of predict_pv_yield/utils.py:16-195: load_config, rank-zero logger, extras, print_config,
log_hyperparameters, finish (same names and argument meaning; rich / omegaconf are optional here)."""
import logging
import os
import warnings
from typing import List, Sequence

import yaml

import predict_pv_yield_amd
from .hydra_lite import DictConfig, to_yaml


def _repo_root() -> str:
    return os.path.normpath(os.path.join(os.path.dirname(predict_pv_yield_amd.__file__), os.pardir))


def load_config(config_file):
    """yaml file -> constructor kwargs: the mapping with Hydra's `_target_` entry removed, so tests can call
    `Model(**load_config("tests/configs/model/conv3d.yaml"))` (contract of predict_pv_yield/utils.py:16-32).
    Relative paths resolve against the repository root, like the reference's."""
    location = config_file if os.path.isabs(config_file) else os.path.join(_repo_root(), config_file)
    with open(location, "r") as stream:
        mapping = yaml.safe_load(stream)
    return {key: value for key, value in mapping.items() if key != "_target_"}


def _is_rank_zero() -> bool:
    return int(os.environ.get("RANK", "0")) == 0


def get_logger(name=__name__, level=logging.INFO) -> logging.Logger:
    """Multi-GPU-friendly python logger: only rank 0 emits (utils.py:35-46)."""
    logger = logging.getLogger(name)
    logger.setLevel(level)
    for lvl in ("debug", "info", "warning", "error", "exception", "fatal", "critical"):
        fn = getattr(logger, lvl)
        setattr(logger, lvl, (lambda f: (lambda *a, **k: f(*a, **k) if _is_rank_zero() else None))(fn))
    return logger


# what a debugger-friendly run switches off (section, key, replacement) -- predict_pv_yield/utils.py:77-85
_FAST_DEV_RUN_OVERRIDES = (("trainer", "gpus", 0), ("datamodule", "pin_memory", False), ("datamodule", "num_workers", 0))


def extras(config: DictConfig) -> None:
    """Optional conveniences driven by the main config, applied in place (contract of utils.py:49-88):
    `ignore_warnings` silences python warnings; `debug` turns on `trainer.fast_dev_run`; and a fast_dev_run drops GPUs,
    pinned memory and loader workers (whatever is currently enabled)."""
    log = get_logger()
    if config.get("ignore_warnings"):
        log.info("Disabling python warnings! <config.ignore_warnings=True>")
        warnings.filterwarnings("ignore")
    if config.get("debug"):
        log.info("Running in debug mode! <config.debug=True>")
        config.trainer.fast_dev_run = True
    if not config.trainer.get("fast_dev_run"):
        return
    log.info("Forcing debugger friendly configuration! <config.trainer.fast_dev_run=True>")
    for section, key, off in _FAST_DEV_RUN_OVERRIDES:
        if (section, key) == ("trainer", "gpus"):
            # the reference drops to the CPU for a debug run (utils.py:80-81); the models of this build have no CPU path,
            # so with an MI355X present the one-batch run stays on it
            import torch
            if torch.cuda.is_available():
                log.info("fast_dev_run: keeping trainer.gpus (the HIP models have no CPU path)")
                continue
        if config[section].get(key):
            config[section][key] = off


def print_config(config: DictConfig,
                 fields: Sequence[str] = ("trainer", "model", "datamodule", "callbacks", "logger", "seed"),
                 resolve: bool = True) -> None:
    """utils.py:91-129: print the config tree and write config_tree.txt."""
    if not _is_rank_zero():
        return
    text = []
    for f in fields:
        section = config.get(f)
        text.append(f"{f}:\n" + (to_yaml(section) if isinstance(section, dict) else f"  {section}\n"))
    out = "CONFIG\n" + "".join(text)
    try:
        import rich.syntax
        import rich.tree
        tree = rich.tree.Tree("CONFIG")
        for f in fields:
            section = config.get(f)
            branch = tree.add(f)
            branch.add(rich.syntax.Syntax(to_yaml(section) if isinstance(section, dict) else str(section), "yaml"))
        rich.print(tree)
    except Exception:
        print(out)
    with open("config_tree.txt", "w") as fp:
        fp.write(out)


def empty(*args, **kwargs):
    pass


def _parameter_census(model) -> dict:
    counts = {"total": 0, "trainable": 0, "not_trainable": 0}
    for p in (model.parameters() if hasattr(model, "parameters") else ()):
        n = p.numel()
        counts["total"] += n
        counts["trainable" if p.requires_grad else "not_trainable"] += n
    return {f"model/params_{k}": v for k, v in counts.items()}


def log_hyperparameters(config: DictConfig, model, datamodule, trainer, callbacks: List, logger: List) -> None:
    """Rank 0 sends the trainer / model / datamodule (and seed / callbacks when present) config sections plus the
    model's parameter counts to the trainer's loggers, once: afterwards `log_hyperparams` is made a no-op so the
    trainer does not log the module's own hparams again (contract of utils.py:136-177)."""
    if not _is_rank_zero():
        return
    hparams = {name: config[name] for name in ("trainer", "model", "datamodule")}
    hparams.update({name: config[name] for name in ("seed", "callbacks") if name in config})
    hparams.update(_parameter_census(model))
    target = trainer.logger
    if target is not None:
        target.log_hyperparams(hparams)
        target.log_hyperparams = empty


def finish(config: DictConfig, model, datamodule, trainer, callbacks: List, logger: List) -> None:
    """utils.py:180-195: make sure every logger closed properly."""
    for lg in logger:
        lg.finalize("success")
