"""`train(config)` — the entry point `run.py` and the reference's `tests/test_training.py` drive.

Behaviour contract (predict_pv_yield/training.py:22-107): seed everything when `config.seed` is present; build the
datamodule, the model, every callback and logger section that names a `_target_`, then the trainer (receiving those
callbacks / loggers); log hyper-parameters; `validate` when the config carries `validate_only`, else `fit`; `test`
afterwards when `test_after_training` is set and this is not a `fast_dev_run`; close the loggers; return
`trainer.callback_metrics[config.optimized_metric]` when that key is configured (for hyper-parameter search), else None.

Organised here as a small table-driven builder (`_build_section` / `_build_group`) plus three phases
(`_assemble`, `_run`, `_wrap_up`) instead of one linear script; under `torch.distributed.run` the process group is
joined first, before anything is instantiated, so that the datamodule shards and the model's parameter broadcast see
the real world size.
"""
from typing import Any, Dict, List, Optional

import torch

from . import distributed, utils
from .hydra_lite import DictConfig, instantiate
from .lightning import seed_everything

log = utils.get_logger(__name__)

torch.set_default_dtype(torch.float32)


def _build_section(config: DictConfig, section: str, **overrides) -> Any:
    """Instantiate `config[section]` (a `_target_` mapping)."""
    node = config[section]
    log.info(f"Instantiating {section} <{node._target_}>")
    return instantiate(node, **overrides)


def _build_group(config: DictConfig, group: str, singular: str) -> List[Any]:
    """Instantiate every child of `config[group]` that names a `_target_` (children without one are settings only)."""
    built = []
    for child in (config.get(group) or {}).values():
        if child and "_target_" in child:
            log.info(f"Instantiating {singular} <{child._target_}>")
            built.append(instantiate(child))
    return built


def _assemble(config: DictConfig) -> Dict[str, Any]:
    parts: Dict[str, Any] = {"config": config}
    parts["datamodule"] = _build_section(config, "datamodule")
    parts["model"] = _build_section(config, "model")
    parts["callbacks"] = _build_group(config, "callbacks", "callback")
    parts["logger"] = _build_group(config, "logger", "logger")
    parts["trainer"] = _build_section(config, "trainer", callbacks=parts["callbacks"], logger=parts["logger"],
                                      _convert_="partial")
    return parts


def _run(parts: Dict[str, Any]) -> None:
    config, trainer = parts["config"], parts["trainer"]
    log.info("Starting training!")
    stage = trainer.validate if "validate_only" in config else trainer.fit
    stage(model=parts["model"], datamodule=parts["datamodule"])
    if config.get("test_after_training") and not config.trainer.get("fast_dev_run"):
        log.info("Starting testing!")
        trainer.test()


def _wrap_up(parts: Dict[str, Any]) -> Optional[float]:
    config, trainer = parts["config"], parts["trainer"]
    log.info("Finalizing!")
    utils.finish(**parts)
    log.info(f"Best checkpoint path:\n{trainer.checkpoint_callback.best_model_path}")
    wanted = config.get("optimized_metric")
    return trainer.callback_metrics[wanted] if wanted else None


def train(config: DictConfig) -> Optional[float]:
    """Run the configured pipeline; returns the optimised metric (or None)."""
    distributed.init_from_env()      # no-op for a single process; must precede any GPU use and any sharding decision
    if "seed" in config:
        seed_everything(config.seed, workers=True)
    parts = _assemble(config)
    log.info("Logging hyperparameters!")
    utils.log_hyperparameters(**parts)
    _run(parts)
    return _wrap_up(parts)
