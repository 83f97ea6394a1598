"""Mirror of predict_pv_yield/utils.py:16-195: load_config, rank-zero logger, extras, print_config,
log_hyperparameters, finish (same names and argument meaning; rich / omegaconf are optional here)."""
import logging
import os
import warnings
from typing import List, Sequence

import yaml

import predict_pv_yield_amd
from .hydra_lite import DictConfig, to_yaml


def load_config(config_file):
    """Open a yaml configuration file and drop the `_target_` line (utils.py:16-32); paths are relative to the
    repository root like the reference (`configs/model/conv3d.yaml`, `tests/configs/model/conv3d.yaml`)."""
    path = os.path.dirname(predict_pv_yield_amd.__file__)
    full = config_file if os.path.isabs(config_file) else f"{path}/../{config_file}"
    with open(full, "r") as cfg:
        config = yaml.load(cfg, Loader=yaml.FullLoader)
    if "_target_" in config.keys():
        config.pop("_target_")  # This is only for Hydra
    return config


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


def extras(config: DictConfig) -> None:
    """utils.py:49-88: ignore_warnings, debug -> fast_dev_run, fast_dev_run -> debug-friendly settings."""
    log = get_logger()
    if config.get("ignore_warnings"):
        log.info("Disabling python warnings! <config.ignore_warnings=True>")
        warnings.filterwarnings("ignore")
    if config.get("debug"):
        log.info("Running in debug mode! <config.debug=True>")
        config.trainer.fast_dev_run = True
    if config.trainer.get("fast_dev_run"):
        log.info("Forcing debugger friendly configuration! <config.trainer.fast_dev_run=True>")
        if config.trainer.get("gpus"):
            config.trainer.gpus = 0
        if config.datamodule.get("pin_memory"):
            config.datamodule.pin_memory = False
        if config.datamodule.get("num_workers"):
            config.datamodule.num_workers = 0


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


def log_hyperparameters(config: DictConfig, model, datamodule, trainer, callbacks: List, logger: List) -> None:
    """utils.py:136-177: send config sections and parameter counts to all loggers."""
    if not _is_rank_zero():
        return
    hparams = {"trainer": config["trainer"], "model": config["model"], "datamodule": config["datamodule"]}
    if "seed" in config:
        hparams["seed"] = config["seed"]
    if "callbacks" in config:
        hparams["callbacks"] = config["callbacks"]
    params = list(model.parameters()) if hasattr(model, "parameters") else []
    hparams["model/params_total"] = sum(p.numel() for p in params)
    hparams["model/params_trainable"] = sum(p.numel() for p in params if p.requires_grad)
    hparams["model/params_not_trainable"] = sum(p.numel() for p in params if not p.requires_grad)
    if trainer.logger is not None:
        trainer.logger.log_hyperparams(hparams)


def finish(config: DictConfig, model, datamodule, trainer, callbacks: List, logger: List) -> None:
    """utils.py:180-195: make sure every logger closed properly."""
    for lg in logger:
        lg.finalize("success")
