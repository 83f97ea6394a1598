"""CPU: host logic around the hot path -- config composition, _target_ instantiation (own tree AND the reference's
yaml tree through the alias table), load_config, model construction / attribute arithmetic / state_dict names,
the Trainer shim's loops, callbacks and loggers, the fake datamodule.  No kernels run here."""
import os

import numpy as np

import pytest
import torch
from torch import nn

from predict_pv_yield_amd import hydra_lite as H
from predict_pv_yield_amd import lightning as pl
from predict_pv_yield_amd.data.batch import BatchML
from predict_pv_yield_amd.data.dataloader import NetCDFDataModule
from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
from predict_pv_yield_amd.utils import extras, load_config, print_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CONFIGS = "/root/reference/configs"


def test_compose_defaults_and_experiment_overlay():
    cfg = H.compose(os.path.join(ROOT, "configs"), "config", ["experiment=example_simple", "trainer.fast_dev_run=true",
                                                               "trainer.gpus=0", "+extra.key=3"])
    assert cfg.model._target_.endswith("baseline.last_value.Model")       # `override /model: baseline.yaml`
    assert cfg.trainer.fast_dev_run is True and cfg.trainer.gpus == 0
    assert cfg.validate_only == "1" and cfg.seed == 518 and cfg.extra.key == 3
    assert cfg.datamodule.n_train_data == 2                                 # experiment body merged over the group
    assert cfg.work_dir == os.getcwd() and cfg.data_dir == os.getcwd() + "/data/"
    assert "${" not in cfg.hydra.run.dir and cfg.hydra.run.dir.startswith("logs/runs/")
    cfg2 = H.compose(os.path.join(ROOT, "configs"), "config", ["model=conv3d_optical_flow", "~print_config"])
    assert cfg2.model.future_frames == "optical_flow" and "print_config" not in cfg2


def test_command_line_group_choice_beats_experiment():
    cfg = H.compose(os.path.join(ROOT, "configs"), "config", ["experiment=example_simple", "model=conv3d"])
    assert cfg.model._target_.endswith("conv3d.model.Model")


def test_oc_env_and_missing_env(monkeypatch, tmp_path):
    (tmp_path / "config.yaml").write_text("a: ${oc.env:PV_TEST_VAR}\nb: ${oc.env:PV_MISSING,fallback}\nc: ${a}/x\n")
    monkeypatch.setenv("PV_TEST_VAR", "hello")
    cfg = H.compose(str(tmp_path), "config")
    assert cfg.a == "hello" and cfg.b == "fallback" and cfg.c == "hello/x"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference tree not present (GPU box)")
def test_reference_config_tree_drives_this_package():
    """The reference's own yaml files compose and instantiate: `_target_` paths resolve through the alias table."""
    cfg = H.compose(REF_CONFIGS, "config", ["logger=csv", "experiment=conv3d", "datamodule.fake_data=true",
                                            "datamodule.data_path=tests/configs/dataset", "trainer.fast_dev_run=true",
                                            "model.fc1_output_features=4"])
    assert cfg.model._target_ == "predict_pv_yield.models.conv3d.model.Model" and cfg.seed == 518
    model = H.instantiate(cfg.model)
    from predict_pv_yield_amd.models.conv3d.model import Model
    assert isinstance(model, Model) and model.number_of_conv3d_layers == 6
    callbacks = [H.instantiate(c) for c in cfg.callbacks.values()]
    trainer = H.instantiate(cfg.trainer, callbacks=callbacks, logger=[H.instantiate(cfg.logger.csv)], _convert_="partial")
    assert isinstance(trainer, pl.Trainer) and trainer.fast_dev_run
    assert trainer.checkpoint_callback.monitor == "MSE/Validation_epoch"
    dm = H.instantiate(cfg.datamodule)
    assert isinstance(dm, NetCDFDataModule)


def test_load_config_strips_target_and_builds_models():
    from predict_pv_yield_amd.models.conv3d.model import Model
    cfg = load_config("tests/configs/model/conv3d.yaml")
    assert "_target_" not in cfg
    m = Model(**cfg)
    assert m.forecast_len_5 == 12 and m.history_len_5 == 12 and m.forecast_len == 12
    assert m.cnn_output_size == 32 * 8 * 8 * 17
    assert list(m.state_dict())[:4] == ["sat_conv0.weight", "sat_conv0.bias", "conv3d_1.weight", "conv3d_1.bias"]
    assert [k for k in m.state_dict() if k.startswith("fc")] == ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
                                                                  "fc3.weight", "fc3.bias", "fc4.weight", "fc4.bias"]
    prod = load_config("configs/model/conv3d.yaml")
    prod["fc1_output_features"] = 2
    m2 = Model(**prod)
    assert m2.cnn_output_size == 1003520 and m2.forecast_len == 6      # T = 18: 12 observed + 6 forecast frames


def test_model_defaults_match_reference_signature():
    import inspect
    from predict_pv_yield_amd.models.conv3d.model import Model
    sig = inspect.signature(Model.__init__)
    ref_defaults = dict(include_pv_yield=True, include_nwp=True, forecast_minutes=30, history_minutes=60,
                        number_of_conv3d_layers=4, conv3d_channels=32, image_size_pixels=64, number_sat_channels=12,
                        fc1_output_features=128, fc2_output_features=128, fc3_output_features=64,
                        output_variable="pv_yield")   # predict_pv_yield/models/conv3d/model.py:18-32
    for k, v in ref_defaults.items():
        assert sig.parameters[k].default == v, k
    assert list(sig.parameters)[1:13] == list(ref_defaults)


def test_extras_and_print_config(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cfg = H.compose(os.path.join(ROOT, "configs"), "config", ["debug=true", "datamodule.num_workers=4",
                                                               "datamodule.pin_memory=true"])
    extras(cfg)
    assert cfg.trainer.fast_dev_run is True and cfg.trainer.gpus == 0
    assert cfg.datamodule.num_workers == 0 and cfg.datamodule.pin_memory is False
    print_config(cfg, resolve=True)
    assert os.path.exists("config_tree.txt")


class _Toy(pl.LightningModule):
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(4, 1)
        self.val_calls = 0

    def forward(self, x):
        return self.lin(x["x"]).squeeze(-1)

    def training_step(self, batch, batch_idx):
        loss = ((self(batch) - batch["y"]) ** 2).mean()
        self.log_dict({"MSE/Train": loss}, on_step=True, on_epoch=True, sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx):
        self.val_calls += 1
        loss = ((self(batch) - batch["y"]) ** 2).mean()
        self.log_dict({"MSE/Validation": loss}, on_step=True, on_epoch=True, sync_dist=True)
        return loss

    def test_step(self, batch, batch_idx):
        self.log_dict({"MSE/Test": ((self(batch) - batch["y"]) ** 2).mean()}, on_step=True, on_epoch=True)

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=0.05)


def _toy_loader(n=8):
    g = torch.Generator().manual_seed(0)
    w = torch.tensor([1.0, -2.0, 0.5, 3.0])
    items = []
    for _ in range(n):
        x = torch.randn(16, 4, generator=g)
        items.append({"x": x, "y": x @ w})
    return torch.utils.data.DataLoader(items, batch_size=None)


def test_trainer_fit_validate_checkpoint_earlystop(tmp_path):
    model = _Toy()
    ckpt = pl.ModelCheckpoint(monitor="MSE/Validation_epoch", save_top_k=1, save_last=True, dirpath=str(tmp_path / "ck"))
    stop = pl.EarlyStopping(monitor="MSE/Validation_epoch", patience=2)
    logger = pl.CSVLogger(save_dir=str(tmp_path), name="csv/")
    tr = pl.Trainer(gpus=0, max_epochs=6, callbacks=[ckpt, stop], logger=[logger], profiler="simple")
    tr.fit(model, _toy_loader(), _toy_loader(3))
    assert tr.callback_metrics["MSE/Validation_epoch"] < 8.0   # started near 14; 6 epochs of Adam(0.05)
    assert "MSE/Train_epoch" in tr.callback_metrics and "MSE/Train_step" in tr.callback_metrics
    assert os.path.exists(ckpt.best_model_path) and os.path.exists(ckpt.last_model_path)
    assert os.path.exists(os.path.join(logger.log_dir, "metrics.csv"))
    state = torch.load(ckpt.last_model_path)
    assert set(state["state_dict"]) == {"lin.weight", "lin.bias"}
    # resume
    model2 = _Toy()
    tr2 = pl.Trainer(gpus=0, max_epochs=7, resume_from_checkpoint=ckpt.last_model_path)
    tr2.fit(model2, _toy_loader(), _toy_loader(3))
    assert tr2.current_epoch == 7
    out = tr2.test(model2, _toy_loader(2))
    assert "MSE/Test_epoch" in out[0]
    preds = tr2.predict(model2, _toy_loader(2))
    assert len(preds) == 2 and preds[0].shape == (16,)


def test_fast_dev_run_runs_one_batch():
    model = _Toy()
    tr = pl.Trainer(gpus=0, fast_dev_run=True)
    tr.fit(model, _toy_loader(5), _toy_loader(5))
    assert tr.global_step == 1 and model.val_calls == 1


def test_log_dict_outside_trainer_is_tolerated():
    m = _Toy()
    m.log_dict({"a": torch.tensor(1.0)})     # tests/models/baseline/test_baseline_model_gsp.py:41-58 calls steps directly
    assert m.current_epoch == 0 and m.logger is None


def test_fake_dataset_and_datamodule_contract():
    cfg = FakeDataConfiguration(batch_size=2, history_minutes=60, forecast_minutes=60, satellite_image_size_pixels=16)
    ds = FakeDataset(cfg, length=2)
    x = next(iter(torch.utils.data.DataLoader(ds, batch_size=None)))
    b = BatchML(**x)
    assert b.satellite.data.shape == (2, 11, 25, 16, 16)
    assert b.pv.pv_yield.shape == (2, 25, 128) and b.gsp.gsp_yield.shape == (2, 5, 32)
    assert b["pv_yield"] is b.pv.pv_yield and b["nwp"] is b.nwp.data
    assert torch.equal(FakeDataset(cfg, 2)[1]["satellite"]["data"], ds[1]["satellite"]["data"])   # reproducible
    dm = NetCDFDataModule(fake_data=True, data_path=os.path.join(ROOT, "tests/configs/dataset"), n_train_data=3, n_val_data=2)
    assert dm.configuration.satellite_image_size_pixels == 16 and dm.configuration.batch_size == 4
    assert len(list(dm.train_dataloader())) == 3 and len(list(dm.val_dataloader())) == 2
    # fake_data=False reads whole-batch files from <data_path>/train|test and fails loudly when they are missing
    real = NetCDFDataModule(fake_data=False, data_path=os.path.join(ROOT, "tests/configs/dataset"), n_train_data=1)
    with pytest.raises(FileNotFoundError):
        next(iter(real.train_dataloader()))


def test_baseline_model_forward_is_persistence():
    from predict_pv_yield_amd.models.baseline.last_value import Model
    m = Model(forecast_minutes=120, history_minutes=30, output_variable="gsp_yield")
    cfg = FakeDataConfiguration(batch_size=3, history_minutes=30, forecast_minutes=120)
    b = FakeDataset(cfg, 1)[0]
    y = m(b)
    assert y.shape == (3, m.forecast_len_30) == (3, 4)
    assert torch.equal(y[:, 0], b["gsp"]["gsp_yield"][:, -5, 0]) and torch.equal(y[:, 0], y[:, 3])


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name,module,cls", [
    ("conv3d_sat_nwp", "predict_pv_yield_amd.models.conv3d.model_sat_nwp", "Model"),
    ("conv3d_nwp", "predict_pv_yield_amd.models.conv3d.model_nwp", "Model"),
    ("perceiver", "predict_pv_yield_amd.models.perceiver.perceiver", "PerceiverModel"),
    ("perceiver_sat_nwp", "predict_pv_yield_amd.models.perceiver.perceiver_nwp_sat", "Model"),
    ("perceiver_conv3d_sat_nwp", "predict_pv_yield_amd.models.perceiver.perceiver_conv3d_nwp_sat", "Model")])
def test_reference_model_yamls_instantiate_here(name, module, cls):
    """Every model yaml of the reference (configs/model/*.yaml) resolves, through the alias table, to the class of this
    package with the reference's keyword arguments; our own configs/model/<name>.yaml names the same class directly."""
    import importlib
    ref_cfg = H._load_yaml(os.path.join(REF_CONFIGS, "model", name + ".yaml"))
    model = H.instantiate(ref_cfg)
    want = getattr(importlib.import_module(module), cls)
    assert type(model) is want
    ours = H._load_yaml(os.path.join(ROOT, "configs", "model", name + ".yaml"))
    assert ours["_target_"] == f"{module}.{cls}"
    assert {k: v for k, v in ours.items() if k != "_target_"} == {k: v for k, v in ref_cfg.items() if k != "_target_"}


def test_new_model_signatures_match_reference():
    import inspect
    from predict_pv_yield_amd.models.conv3d.model_sat_nwp import Model as SatNwp
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel
    from predict_pv_yield_amd.models.perceiver.perceiver_conv3d_nwp_sat import Model as PConv
    sig = inspect.signature(SatNwp.__init__)          # model_sat_nwp.py:18-37
    assert list(sig.parameters)[1:18] == ["include_pv_or_gsp_yield_history", "include_nwp", "forecast_minutes", "history_minutes",
                                          "number_of_conv3d_layers", "conv3d_channels", "image_size_pixels",
                                          "nwp_image_size_pixels", "number_sat_channels", "number_nwp_channels",
                                          "fc1_output_features", "fc2_output_features", "fc3_output_features",
                                          "output_variable", "embedding_dem", "include_pv_yield_history",
                                          "include_future_satellite"]
    assert sig.parameters["embedding_dem"].default == 16 and sig.parameters["number_nwp_channels"].default == 10
    sig = inspect.signature(PerceiverModel.__init__)  # perceiver.py:46-56
    assert [sig.parameters[k].default for k in ("history_minutes", "forecast_minutes", "batch_size", "num_latents", "latent_dim",
                                                "embedding_dem", "output_variable")] == [30, 120, 32, 128, 64, 16, "pv_yield"]
    sig = inspect.signature(PConv.__init__)           # perceiver_conv3d_nwp_sat.py:64-76
    assert sig.parameters["conv3d_channels"].default == 16 and sig.parameters["use_future_satellite_images"].default is True
    m = PerceiverModel(history_minutes=3, forecast_minutes=3)           # tests/models/perceiver/test_perceiver.py:9
    assert m.total_seq_length == 1 and len(m.perceiver.layers) == 1


def test_nwp_only_model_mirrors_the_reference_signature_and_parameter_names():
    """models/conv3d/model_nwp.py: constructor keywords / defaults of model_nwp.py:18-35, the registration order of its
    layers (state_dict keys = the oracle's, which is pinned on the reference module), the class name quirk, and no CPU path."""
    import inspect
    from oracle import conv3d_oracle as co
    from predict_pv_yield_amd.models.conv3d.model_nwp import Model
    sig = inspect.signature(Model.__init__)
    assert list(sig.parameters)[1:16] == ["include_pv_or_gsp_yield_history", "include_nwp", "forecast_minutes", "history_minutes",
                                          "number_of_conv3d_layers", "conv3d_channels", "nwp_image_size_pixels",
                                          "number_nwp_channels", "fc1_output_features", "fc2_output_features",
                                          "fc3_output_features", "output_variable", "embedding_dem",
                                          "include_pv_yield_history", "include_future_satellite"]
    assert [sig.parameters[k].default for k in ("forecast_minutes", "history_minutes", "nwp_image_size_pixels",
                                                "number_nwp_channels", "output_variable", "embedding_dem")] == [30, 60, 64, 10, "gsp_yield", 16]
    kw = dict(forecast_minutes=120, history_minutes=30, number_of_conv3d_layers=2, nwp_image_size_pixels=8, number_nwp_channels=3,
              fc1_output_features=8, fc3_output_features=8)
    m, o = Model(**kw), co.OracleConv3dNwpModel(**kw)
    assert Model.name == "conv3d_sat_nwp"                      # model_nwp.py:16 keeps the sat+nwp name
    assert list(m.state_dict()) == list(o.state_dict())
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in o.state_dict().values()]
    assert m.nwp_cnn_output_size == o.nwp_cnn_output_size == 32 * 4 * 4 * 4 and m.forecast_len == o.forecast_len == 4
    with pytest.raises(RuntimeError, match="MI355X"):
        m({"nwp": {"data": torch.zeros(1, 3, 4, 8, 8)}})


# ---- step after the path: validation results CSV + per-horizon metrics, on the host with the
# ---- parameter-free baseline, exactly as the reference's own tests drive them ------------------------------------------
def _gsp_baseline_and_loader(n_batches=3, batch_size=4):
    from predict_pv_yield_amd.models.baseline.last_value import Model
    cfg = FakeDataConfiguration(batch_size=batch_size, history_minutes=30, forecast_minutes=120, satellite_image_size_pixels=8)
    model = Model(forecast_minutes=cfg.forecast_minutes, history_minutes=cfg.history_minutes, output_variable="gsp_yield")
    loader = torch.utils.data.DataLoader(FakeDataset(cfg, length=n_batches), batch_size=None)
    return model, loader, cfg


def test_baseline_trainer_validation_writes_the_results_csv(tmp_path):
    """reference tests/models/baseline/test_baseline_model_gsp.py:81-111: validate -> `{results_file_name}_0.csv` with
    len(loader) * batch_size * forecast_len_30 rows and the five named columns."""
    import pandas as pd
    from oracle import conv3d_oracle as co
    model, loader, cfg = _gsp_baseline_and_loader()
    trainer = pl.Trainer(gpus=0, max_epochs=1)
    model.results_file_name = f"{tmp_path}/temp"
    _ = trainer.validate(model, loader)
    results_df = pd.read_csv(f"{model.results_file_name}_0.csv")
    assert len(results_df) == len(loader) * cfg.batch_size * model.forecast_len_30
    for column in ("t0_datetime_utc", "target_datetime_utc", "gsp_id", "actual_gsp_pv_outturn_mw", "forecast_gsp_pv_outturn_mw"):
        assert column in results_df.keys()
    assert len(results_df.keys()) == 5
    # content: every row against the oracle's restatement of make_validation_results
    want = []
    for item in loader:
        b = BatchML(**item)
        want += co.validation_results_rows(model(b).numpy(), b.gsp.gsp_yield.numpy(), b.gsp.gsp_capacity.numpy(),
                                           b.gsp.gsp_id.numpy(), b.metadata.t0_datetime_utc.numpy(), model.forecast_len_30)
    assert list(results_df["gsp_id"]) == [r[2] for r in want]
    assert np.allclose(results_df["actual_gsp_pv_outturn_mw"], [r[3] for r in want], rtol=1e-6)
    assert np.allclose(results_df["forecast_gsp_pv_outturn_mw"], [r[4] for r in want], rtol=1e-6)
    assert list(pd.to_datetime(results_df["target_datetime_utc"])) == [pd.Timestamp(r[1]) for r in want]
    # a second model object starts with its own, empty table (results_dfs used to be a shared class attribute)
    other, _, _ = _gsp_baseline_and_loader()
    assert other.results_dfs == [] and other.results_dfs is not model.results_dfs


def test_baseline_trainer_test_and_direct_step_calls():
    """reference test_baseline_model_gsp.py:41-78: steps called directly (no trainer) and trainer.test()."""
    model, loader, _ = _gsp_baseline_and_loader(2)
    batch = next(iter(loader))
    model.validation_step(batch, 0)
    model.training_step(batch, 0)
    model.test_step(batch, 0)
    out = pl.Trainer(gpus=0, max_epochs=1).test(model, loader)
    assert np.isfinite(out[0]["NMAE/Test_epoch"])


def test_per_horizon_metrics_kat_mae_overwrites_mse():
    """base_model.py:121-141: `MSE_forecast_horizon_{i}/{tag}` carries the MAE of horizon i (the MAE dictionary reuses
    the MSE key names and is merged last), for i < forecast_len_30 only."""
    from oracle import conv3d_oracle as co
    model, loader, _ = _gsp_baseline_and_loader(1, batch_size=3)
    batch = BatchML(**next(iter(loader)))
    # hand-made case: errors of 1, 2, 3, 4 at the four horizons for every example
    y_hat = model(batch)
    batch.gsp.gsp_yield[:, -4:, 0] = y_hat - torch.tensor([1.0, 2.0, 3.0, 4.0])
    model.validation_step(batch, 0)
    logged = {k: float(v) for k, v in model._logged.items()}
    for i, err in enumerate([1.0, 2.0, 3.0, 4.0]):
        assert logged[f"MSE_forecast_horizon_{i}/Validation"] == pytest.approx(err)         # the MAE, not err ** 2
    assert f"MSE_forecast_horizon_4/Validation" not in logged
    assert logged["NMAE/Validation"] == pytest.approx(2.5) and logged["MSE/Validation"] == pytest.approx(7.5)
    want = co.logged_horizon_metrics(model(batch), batch.gsp.gsp_yield[:, -4:, 0], 4, "Validation")
    assert {k: v for k, v in logged.items() if "horizon" in k} == pytest.approx(want)
    # pv_yield output with 30 forecast minutes: six 5-minute steps but forecast_len_30 == 1 -> only horizon 0 is logged
    from predict_pv_yield_amd.models.baseline.last_value import Model
    pv_model = Model(forecast_minutes=30, history_minutes=60, output_variable="pv_yield")
    cfg = FakeDataConfiguration(batch_size=2, history_minutes=60, forecast_minutes=30, satellite_image_size_pixels=8)
    pv_model.test_step(FakeDataset(cfg, 1)[0], 0)
    assert [k for k in pv_model._logged if "horizon" in k] == ["MSE_forecast_horizon_0/Test"]


def test_trainable_models_still_refuse_the_cpu():
    from predict_pv_yield_amd.models.base_model import BaseModel

    class Tiny(BaseModel):
        history_minutes, forecast_minutes = 30, 30

        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(1, 1)

    with pytest.raises(RuntimeError, match="MI355X"):
        Tiny()._losses(torch.zeros(2, 6), torch.zeros(2, 6))


def test_config1_exp001_shaped_cpu_plumbing_run(tmp_path, monkeypatch):
    """BASELINE.json configs[0]: experiments/001-shaped synthetic data (19 five-minute steps, HRV only, 128 x 128, NWP
    [32,10,19,2,2], PV [32,19,128]) through run.py's surface on the HOST, world_size 1: it runs and the loss is finite."""
    from predict_pv_yield_amd.training import train
    from predict_pv_yield_amd.utils import extras
    monkeypatch.chdir(tmp_path)
    cfg = H.compose(os.path.join(ROOT, "configs"), "config",
                    ["experiment=exp001_plumbing", f"datamodule.data_path={ROOT}/configs/dataset/exp001",
                     "optimized_metric=NMAE/Validation_epoch"])
    extras(cfg)
    assert cfg.trainer.gpus == 0
    dm = H.instantiate(cfg.datamodule)
    batch = BatchML(**next(iter(dm.val_dataloader())))
    # experiments/001_...py:225-251,288-293: history_len 6 + forecast_len 12 + 1 = 19 steps; the first 7 frames are the
    # Conv2d's time-as-channel input
    assert batch.satellite.data.shape == (32, 1, 19, 128, 128) and batch.satellite.data[:, :, :7].shape[2] == 7
    assert batch.nwp.data.shape[:2] == (32, 10) and batch.nwp.data.shape[-2:] == (2, 2)
    assert batch.pv.pv_yield.shape == (32, 19, 128)
    score = train(cfg)
    assert np.isfinite(score) and 0.0 < score < 1.0
    assert os.path.exists("results_epoch_0.csv")


def test_every_experiment_names_an_existing_dataset_configuration():
    """Each shipped experiment composes, and the dataset configuration it points at exists (a missing one silently fell back
    to default shapes once)."""
    from predict_pv_yield_amd import hydra_lite as H
    from tests.test_gpu_training import SHIPPED_EXPERIMENTS
    found = sorted(f[:-5] for f in os.listdir(os.path.join(ROOT, "configs", "experiment")) if f.endswith(".yaml"))
    assert found == SHIPPED_EXPERIMENTS
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        for name in found:
            cfg = H.compose(os.path.join(ROOT, "configs"), "config", [f"experiment={name}"])
            path = cfg.datamodule.get("data_path")
            if path is not None:
                assert os.path.isabs(path) and os.path.exists(os.path.join(path, "configuration.yaml")), (name, path)
    finally:
        os.chdir(cwd)


def test_graph_helper_maps_and_refills_attribute_container_batches():
    """graphs.GraphedTrainStep keeps a static copy of the example batch and copies every new batch into it: that must reach the
    tensors of a BatchML (sections as attribute namespaces), not only those of dicts -- a container it does not look into would
    make the graph replay the captured batch for ever."""
    import torch
    from predict_pv_yield_amd.data.batch import BatchML
    from predict_pv_yield_amd.graphs import _copy_into, _map_tensors
    a = BatchML(satellite={"data": torch.arange(6.).reshape(2, 3)}, gsp={"gsp_yield": torch.ones(2, 4), "note": "x"})
    static = _map_tensors(a, lambda t: t.clone())
    assert isinstance(static, BatchML) and static.satellite.data is not a.satellite.data
    assert torch.equal(static.satellite.data, a.satellite.data) and static.gsp.note == "x"
    b = BatchML(satellite={"data": torch.full((2, 3), 7.)}, gsp={"gsp_yield": torch.zeros(2, 4), "note": "x"})
    kept = static.satellite.data
    _copy_into(static, b)
    assert static.satellite.data is kept and torch.equal(kept, b.satellite.data)
    assert torch.equal(static["gsp_yield"], b["gsp_yield"])
    nested = {"k": [torch.zeros(2), (torch.ones(1),)]}
    st = _map_tensors(nested, lambda t: t.clone())
    _copy_into(st, {"k": [torch.full((2,), 3.), (torch.full((1,), 5.),)]})
    assert st["k"][0].tolist() == [3.0, 3.0] and st["k"][1][0].item() == 5.0
