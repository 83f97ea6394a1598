"""train(config) — mirror of predict_pv_yield/training.py:22-107: seed, instantiate datamodule / model /
callbacks / loggers / trainer from `_target_` configs, fit (or validate when `validate_only` is set), test,
return the optimised metric."""
from typing import List, Optional

import torch

from . import utils
from .hydra_lite import DictConfig, instantiate
from .lightning import Callback, LightningDataModule, LightningLoggerBase, LightningModule, Trainer, seed_everything

log = utils.get_logger(__name__)

torch.set_default_dtype(torch.float32)


def train(config: DictConfig) -> Optional[float]:
    if "seed" in config:
        seed_everything(config.seed, workers=True)

    log.info(f"Instantiating datamodule <{config.datamodule._target_}>")
    datamodule: LightningDataModule = instantiate(config.datamodule)

    log.info(f"Instantiating model <{config.model._target_}>")
    model: LightningModule = instantiate(config.model)

    callbacks: List[Callback] = []
    if "callbacks" in config and config.callbacks:
        for _, cb_conf in config.callbacks.items():
            if cb_conf and "_target_" in cb_conf:
                log.info(f"Instantiating callback <{cb_conf._target_}>")
                callbacks.append(instantiate(cb_conf))

    logger: List[LightningLoggerBase] = []
    if "logger" in config and config.logger:
        for _, lg_conf in config.logger.items():
            if lg_conf and "_target_" in lg_conf:
                log.info(f"Instantiating logger <{lg_conf._target_}>")
                logger.append(instantiate(lg_conf))

    log.info(f"Instantiating trainer <{config.trainer._target_}>")
    trainer: Trainer = instantiate(config.trainer, callbacks=callbacks, logger=logger, _convert_="partial")

    log.info("Logging hyperparameters!")
    utils.log_hyperparameters(config=config, model=model, datamodule=datamodule, trainer=trainer,
                              callbacks=callbacks, logger=logger)

    log.info("Starting training!")
    if "validate_only" in config:
        trainer.validate(model=model, datamodule=datamodule)
    else:
        trainer.fit(model=model, datamodule=datamodule)

    if config.get("test_after_training") and not config.trainer.get("fast_dev_run"):
        log.info("Starting testing!")
        trainer.test()

    log.info("Finalizing!")
    utils.finish(config=config, model=model, datamodule=datamodule, trainer=trainer, callbacks=callbacks,
                 logger=logger)
    log.info(f"Best checkpoint path:\n{trainer.checkpoint_callback.best_model_path}")

    optimized_metric = config.get("optimized_metric")
    if optimized_metric:
        return trainer.callback_metrics[optimized_metric]
