"""GPU parity of the Perceiver module (models/perceiver/perceiver_core.py) against the torch-CPU restatement of
perceiver_pytorch.Perceiver (oracle/perceiver_oracle.py; the package itself is absent and unpinned: parity with it is
unpinned, parity between the HIP path and the restatement is what these tests hold)."""
import pytest
import torch

from oracle import perceiver_oracle as po

pytestmark = pytest.mark.gpu

KW = dict(input_channels=11, input_axis=2, num_freq_bands=6, max_freq=10, depth=4, num_latents=32, latent_dim=64,
          num_classes=24, weight_tie_layers=True)


def _pair(kw, device, seed=0):
    from predict_pv_yield_amd.models.perceiver.perceiver_core import Perceiver
    torch.manual_seed(seed)
    oracle = po.OraclePerceiver(**kw)
    model = Perceiver(**kw)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    return oracle, model.to(device)


def test_position_features_match():
    from predict_pv_yield_amd.models.perceiver.perceiver_core import fourier_position_features
    a = fourier_position_features((5, 7), 10.0, 6)
    b = po.position_encoding((5, 7), 10.0, 6)
    assert a.shape == (5, 7, 26) and torch.equal(a, b)


def test_weight_tying_structure(device):
    oracle, model = _pair(KW, device)
    assert model.layers[1][0] is model.layers[3][0] and model.layers[0][0] is not model.layers[1][0]
    n_unique = sum(p.numel() for p in model.parameters())
    assert n_unique == sum(p.numel() for p in oracle.parameters())
    assert any(k.startswith("layers.3.0.fn.to_kv") for k in model.state_dict())       # tied blocks appear under every layer


@pytest.mark.parametrize("tie", [True, False])
def test_forward_and_gradients_vs_oracle(tie, device):
    kw = dict(KW, weight_tie_layers=tie, depth=3)
    oracle, model = _pair(kw, device, seed=1)
    g = torch.Generator().manual_seed(2)
    data = torch.randn(3, 12, 10, 11, generator=g)
    ref = oracle(data)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    out = model(data.to(device))
    assert out.shape == (3, 24)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=2e-4, atol=2e-4)
    out.backward(dy.to(device))
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        scale = q.grad.abs().max().item() + 1e-12
        assert (p.grad.cpu() - q.grad).abs().max().item() <= 2e-3 * scale + 1e-6, k


def test_reference_shape_forward(device):
    """perceiver.py:70-80 as configured by the reference: 11 channels, 64 x 64, 128 latents x 64, 512 classes, tied."""
    kw = dict(input_channels=11, input_axis=2, num_freq_bands=6, max_freq=10, depth=3, num_latents=128, latent_dim=64,
              num_classes=512, weight_tie_layers=True)
    oracle, model = _pair(kw, device, seed=3)
    data = torch.randn(2, 64, 64, 11, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = oracle(data)
        out = model(data.to(device))
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=2e-4)


def test_on_cpu_fails_loudly():
    from predict_pv_yield_amd.models.perceiver.perceiver_core import Perceiver
    with pytest.raises(RuntimeError, match="MI355X"):
        Perceiver(**KW)(torch.zeros(1, 4, 4, 11))


# ---- PerceiverModel (predict_pv_yield/models/perceiver/perceiver.py:42-200) --------------------------------------------
def _model_pair(device, **kw):
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel
    torch.manual_seed(5)
    oracle = po.OraclePerceiverModel(**kw)
    model = PerceiverModel(**kw)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    return oracle, model.to(device)


def _model_batch(b, t5, t30, px, seed=6):
    g = torch.Generator().manual_seed(seed)
    return dict(sat=torch.randn(b, 11, t5, px, px, generator=g), nwp=torch.randn(b, 10, 3, 64, 64, generator=g),
                pv=torch.rand(b, t5, 128, generator=g), gsp=torch.rand(b, t30, 32, generator=g),
                row=torch.randint(0, 940, (b, 128), generator=g), gsp_id=torch.randint(1, 339, (b, 32), generator=g))


@pytest.mark.parametrize("output_variable", ["pv_yield", "gsp_yield"])
def test_perceiver_model_train_step_vs_oracle(output_variable, device):
    kw = dict(history_minutes=10, forecast_minutes=30 if output_variable == "pv_yield" else 60, batch_size=2, num_latents=16,
              latent_dim=64, embedding_dem=16, output_variable=output_variable)
    oracle, model = _model_pair(device, **kw)
    t5 = kw["history_minutes"] // 5 + kw["forecast_minutes"] // 5 + 1
    t30 = kw["history_minutes"] // 30 + kw["forecast_minutes"] // 30 + 1
    d = _model_batch(2, t5, t30, 8)
    yld, ids = (d["pv"], d["row"]) if output_variable == "pv_yield" else (d["gsp"], d["gsp_id"])
    y_ref = oracle(d["sat"], d["nwp"], yld, ids)
    assert y_ref.shape == (2, oracle.forecast_len)
    target = yld[:2, -oracle.forecast_len:, 0]
    loss_ref = (y_ref - target).abs().mean()
    loss_ref.backward()
    batch = {"satellite": {"data": d["sat"].to(device)}, "nwp": {"data": d["nwp"].to(device)},
             "pv": {"pv_yield": d["pv"].to(device), "pv_system_row_number": d["row"].to(device)},
             "gsp": {"gsp_yield": d["gsp"].to(device), "gsp_id": d["gsp_id"].to(device)}}
    y = model(batch)
    torch.testing.assert_close(y.detach().cpu(), y_ref.detach(), rtol=1e-3, atol=1e-4)
    loss = model.training_step(batch, 0)
    assert abs(float(loss.detach()) - float(loss_ref)) <= 1e-4 * abs(float(loss_ref)) + 1e-6
    loss.backward()
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale = q.grad.abs().max().item() + 1e-12
        assert (p.grad.cpu() - q.grad).abs().max().item() <= 5e-3 * scale + 1e-6, k
    opt = model.configure_optimizers()
    opt.step()                                  # all parameters step through the multi-tensor Adam


def test_perceiver_model_reference_test_shapes(device):
    """tests/models/perceiver/test_perceiver.py: init with 3/3 minutes; forward with 30/60 minutes on 16 px satellite,
    64 px NWP, batch 2 -> [2, 12]."""
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, make_fake_batch
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel, params
    _ = PerceiverModel(history_minutes=3, forecast_minutes=3, nwp_channels=params["nwp_channels"])
    model = PerceiverModel(history_minutes=30, forecast_minutes=60, nwp_channels=params["nwp_channels"],
                           embedding_dem=2048).to(device)
    cfg = FakeDataConfiguration(batch_size=2, history_minutes=30, forecast_minutes=60, satellite_image_size_pixels=16,
                                nwp_image_size_pixels=64)
    x = make_fake_batch(cfg, torch.Generator().manual_seed(1))
    y = model(x.to(device))
    assert len(y.shape) == 2 and y.shape[0] == 2 and y.shape[1] == 60 // 5


# ---- the two other Perceiver models (perceiver_nwp_sat.py, perceiver_conv3d_nwp_sat.py) --------------------------------
def _variant_step(oracle, model, d, output_variable, device, tol=5e-3):
    yld = d["pv"] if output_variable == "pv_yield" else d["gsp"]
    y_ref = oracle(d["sat"], d["nwp"], yld, d["row"])
    loss_ref = (y_ref - yld[:2, -oracle.forecast_len:, 0]).abs().mean()
    loss_ref.backward()
    batch = {"satellite": {"data": d["sat"].to(device)}, "nwp": {"data": d["nwp"].to(device)},
             "pv": {"pv_yield": d["pv"].to(device), "pv_system_row_number": d["row"].to(device)},
             "gsp": {"gsp_yield": d["gsp"].to(device), "gsp_id": d["gsp_id"].to(device)}}
    y = model(batch)
    torch.testing.assert_close(y.detach().cpu(), y_ref.detach(), rtol=1e-3, atol=1e-4)
    loss = model.training_step(batch, 0)
    assert abs(float(loss.detach()) - float(loss_ref)) <= 1e-4 * abs(float(loss_ref)) + 1e-6
    loss.backward()
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        if q.grad is None:
            continue
        scale = q.grad.abs().max().item() + 1e-12
        assert (p.grad.cpu() - q.grad).abs().max().item() <= tol * scale + 1e-6, k
    model.configure_optimizers().step()


@pytest.mark.parametrize("output_variable,embedding_dem", [("pv_yield", 16), ("gsp_yield", 0)])
def test_perceiver_nwp_sat_model_vs_oracle(output_variable, embedding_dem, device):
    from predict_pv_yield_amd.models.perceiver.perceiver_nwp_sat import Model
    kw = dict(history_minutes=10, forecast_minutes=60 if output_variable == "gsp_yield" else 20, batch_size=2, num_latents=16,
              latent_dim=64, embedding_dem=embedding_dem, output_variable=output_variable)
    torch.manual_seed(7)
    oracle = po.OraclePerceiverNwpSatModel(**kw)
    model = Model(**kw)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    model.to(device)
    t5 = kw["history_minutes"] // 5 + kw["forecast_minutes"] // 5 + 1
    t30 = kw["history_minutes"] // 30 + kw["forecast_minutes"] // 30 + 1
    d = _model_batch(2, t5, t30, 8)
    d["nwp"] = torch.randn(2, 10, 2, 8, 8, generator=torch.Generator().manual_seed(9))   # same pixel size as the satellite
    _variant_step(oracle, model, d, output_variable, device)


@pytest.mark.parametrize("use_future", [True, False])
def test_perceiver_conv3d_nwp_sat_model_vs_oracle(use_future, device):
    from predict_pv_yield_amd.models.perceiver.perceiver_conv3d_nwp_sat import Model
    kw = dict(history_minutes=10, forecast_minutes=20, batch_size=2, num_latents=12, latent_dim=24, embedding_dem=0,
              output_variable="pv_yield", conv3d_channels=8, use_future_satellite_images=use_future)
    torch.manual_seed(8)
    oracle = po.OraclePerceiverConv3dNwpSatModel(**kw)
    model = Model(**kw)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    model.to(device)
    d = _model_batch(2, 7, 2, 12)
    d["nwp"] = torch.randn(2, 10, 7, 12, 12, generator=torch.Generator().manual_seed(10))
    sat_before = d["sat"].clone()
    _variant_step(oracle, model, d, "pv_yield", device)
    assert torch.equal(d["sat"], sat_before)                 # the caller's batch is not modified


def test_perceiver_model_trainer_fit(device):
    """The Lightning-shim Trainer drives PerceiverModel like any other module: fit one epoch, predict."""
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel
    torch.manual_seed(11)
    model = PerceiverModel(history_minutes=10, forecast_minutes=20, batch_size=2, num_latents=16, latent_dim=64)
    ds = FakeDataset(FakeDataConfiguration(batch_size=2, history_minutes=10, forecast_minutes=20, satellite_image_size_pixels=8,
                                           nwp_image_size_pixels=64), length=2)
    loader = torch.utils.data.DataLoader(ds, batch_size=None)
    trainer = pl.Trainer(gpus=1, max_epochs=1)
    trainer.fit(model, loader)
    out = trainer.predict(model, loader)
    assert len(out) == 2 and out[0].shape == (2, model.forecast_len_5)
    assert all(torch.isfinite(p).all() for p in model.parameters())


@pytest.mark.parametrize("operand_dtype", ["f32", "bf16"])
def test_perceiver_model_trainer_in_hip_graph_mode_equals_the_eager_trainer(device, operand_dtype):
    """Trainer(hip_graph=True) on the weight-tied PerceiverModel (gradients of tied parameters accumulate into kept buffers, the
    shared context's gradient collects over the layers): parameters and metrics of the eager fit, bit for bit."""
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.data.fake import FakeDataConfiguration, FakeDataset
    from predict_pv_yield_amd.models.perceiver.perceiver import PerceiverModel
    results = []
    for graph in (False, True):
        torch.manual_seed(12)
        model = PerceiverModel(history_minutes=10, forecast_minutes=20, batch_size=2, num_latents=16, latent_dim=64,
                               operand_dtype=operand_dtype)
        ds = FakeDataset(FakeDataConfiguration(batch_size=2, history_minutes=10, forecast_minutes=20, satellite_image_size_pixels=8,
                                               nwp_image_size_pixels=64), length=6)
        loader = torch.utils.data.DataLoader(ds, batch_size=None)
        trainer = pl.Trainer(gpus=1, max_epochs=1, hip_graph=graph, log_every_n_steps=1)
        trainer.fit(model, loader)
        results.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, dict(trainer.callback_metrics)))
    (p0, m0), (p1, m1) = results
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    assert m0.keys() == m1.keys() and all(m0[k] == m1[k] for k in m0), (m0, m1)
