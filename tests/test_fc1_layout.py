"""fc1.weight of the bf16 Conv3D model is STORED channels-last along its input axis (models/conv3d/_fc1_layout.py); whatever
enters or leaves the process -- Module.state_dict(), HipAdam.state_dict(), HipAdam.moments() -- is the reference's
[N, C*T*H*W] column order (predict_pv_yield/models/conv3d/model.py:122-125).  Host logic only: runs without a GPU."""
import copy

import torch

from oracle import conv3d_oracle as co
from predict_pv_yield_amd.models.conv3d import _fc1_layout as L
from predict_pv_yield_amd.models.conv3d.model import Model
from predict_pv_yield_amd.optim import HipAdam

KW = dict(include_pv_yield=False, include_nwp=False, forecast_minutes=60, history_minutes=60, number_of_conv3d_layers=4,
          conv3d_channels=32, image_size_pixels=16, number_sat_channels=11, fc1_output_features=16, fc2_output_features=16,
          fc3_output_features=16)


def test_permutation_is_the_flatten_order_change():
    n, c, t, h, w = 3, 32, 2, 3, 4
    act = torch.randn(5, c, t, h, w)
    ref_w = torch.randn(n, c * t * h * w)
    y_ref = act.reshape(5, -1) @ ref_w.t()                                  # the reference: flatten NCDHW
    y_cl = act.permute(0, 2, 3, 4, 1).reshape(5, -1) @ L.to_channels_last(ref_w, c).t()   # the tower's image, stored weight
    torch.testing.assert_close(y_cl, y_ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(L.to_reference(L.to_channels_last(ref_w, c), c), ref_w)


def test_state_dict_speaks_the_reference_layout():
    torch.manual_seed(518)
    oracle = co.OracleConv3dModel(**KW)
    torch.manual_seed(518)
    model = Model(**KW, precision="bf16")
    assert model._fc1_k_channels == 32 and L.k_channels(model.fc1.weight) == 32
    sd, ref = model.state_dict(), oracle.state_dict()
    assert list(sd) == list(ref)
    for k in ref:        # same seed, same initial values, same shapes -- through the hook for fc1.weight
        assert sd[k].shape == ref[k].shape and torch.equal(sd[k], ref[k]), k
    assert torch.equal(model.fc1.weight.detach(), L.to_channels_last(ref["fc1.weight"], 32))
    assert not torch.equal(model.fc1.weight.detach(), ref["fc1.weight"])
    # load: reference layout in, stored layout inside, reference layout out again; the caller's dict is left alone
    other = {k: torch.randn_like(v) for k, v in ref.items()}
    keep = {k: v.clone() for k, v in other.items()}
    model.load_state_dict(other)
    assert all(torch.equal(other[k], keep[k]) for k in keep)
    assert torch.equal(model.fc1.weight.detach(), L.to_channels_last(keep["fc1.weight"], 32))
    assert torch.equal(model.state_dict()["fc1.weight"], keep["fc1.weight"])
    assert torch.equal(L.reference_layout(model.fc1.weight), keep["fc1.weight"])
    # a deep copy keeps the layout (module attribute + hooks) AND the parameter mark (Model.__setstate__)
    twin = copy.deepcopy(model)
    assert torch.equal(twin.state_dict()["fc1.weight"], keep["fc1.weight"])
    assert L.k_channels(twin.fc1.weight) == 32
    # the f32 model keeps the reference's order in memory
    plain = Model(**KW, precision="fp32")
    plain.load_state_dict(keep)
    assert plain._fc1_k_channels == 0 and torch.equal(plain.fc1.weight.detach(), keep["fc1.weight"])


def test_optimizer_state_speaks_the_reference_layout():
    model = Model(**KW, precision="bf16")
    opt = model.configure_optimizers()
    assert isinstance(opt, HipAdam)
    p = model.fc1.weight
    st = opt._init_state(p)
    st["exp_avg"].copy_(torch.randn_like(p))
    st["exp_avg_sq"].copy_(torch.rand_like(p))
    stored = (st["exp_avg"].clone(), st["exp_avg_sq"].clone())
    sd = opt.state_dict()
    idx = [i for i, q in enumerate(opt._params_in_order()) if q is p][0]
    assert torch.equal(sd["state"][idx]["exp_avg"], L.to_reference(stored[0], 32))
    assert torch.equal(sd["state"][idx]["exp_avg_sq"], L.to_reference(stored[1], 32))
    assert torch.equal(opt.state[p]["exp_avg"], stored[0])             # the live state was not touched
    m, v = opt.moments(p)
    assert torch.equal(m, L.to_reference(stored[0], 32)) and torch.equal(v, L.to_reference(stored[1], 32))
    # a second optimiser resumed from that dict holds the stored layout again
    model2 = Model(**KW, precision="bf16")
    opt2 = model2.configure_optimizers()
    full = {"state": {i: {"step": torch.tensor(1.0), "exp_avg": torch.zeros_like(q), "exp_avg_sq": torch.zeros_like(q)}
                      for i, q in enumerate(opt2._params_in_order())}, "param_groups": sd["param_groups"]}
    full["state"][idx] = copy.deepcopy(sd["state"][idx])
    full["state"][idx]["step"] = torch.tensor(1.0)
    opt2.load_state_dict(full)
    assert torch.equal(opt2.state[model2.fc1.weight]["exp_avg"], stored[0])
    assert torch.equal(opt2.state[model2.fc1.weight]["exp_avg_sq"], stored[1])


def test_a_deep_copied_optimizer_still_knows_the_layout():
    """copy.deepcopy(optimizer) creates fresh Parameter objects without Python attributes; the layout mark also travels on the
    state tensor."""
    model = Model(**KW, precision="bf16")
    opt = model.configure_optimizers()
    p = model.fc1.weight
    st = opt._init_state(p)
    st["exp_avg"].copy_(torch.randn_like(p))
    ref = opt.moments(p)[0].clone()
    twin = copy.deepcopy(opt)
    q = [t for g in twin.param_groups for t in g["params"] if t.shape == p.shape][0]
    assert not hasattr(q, "_pv_k_channels")
    assert torch.equal(twin.moments(q)[0], ref)


def test_the_layout_mark_follows_the_module_not_the_parameter_object():
    """whatever creates fresh Parameter objects (deepcopy, pickle, a converting _apply) drops their Python attributes;
    the module re-applies the mark, so HipAdam(copied_model.parameters()) -- built WITHOUT configure_optimizers() -- still
    writes and reads checkpoints in the reference's column order."""
    import pickle
    torch.manual_seed(3)
    model = Model(**KW, precision="bf16")
    ref_w = model.state_dict()["fc1.weight"].clone()
    for twin in (copy.deepcopy(model), pickle.loads(pickle.dumps(model))):
        assert L.k_channels(twin.fc1.weight) == 32
        assert torch.equal(twin.state_dict()["fc1.weight"], ref_w)
        opt = HipAdam(twin.parameters(), lr=5e-4)                 # not through configure_optimizers()
        p = twin.fc1.weight
        st = opt._init_state(p)
        st["exp_avg"].copy_(torch.randn_like(p))
        st["exp_avg_sq"].copy_(torch.rand_like(p))
        stored = st["exp_avg"].clone()
        idx = [i for i, q in enumerate(opt._params_in_order()) if q is p][0]
        sd = opt.state_dict()
        assert torch.equal(sd["state"][idx]["exp_avg"], L.to_reference(stored, 32))      # reference order leaves
        opt2 = HipAdam(copy.deepcopy(twin).parameters(), lr=5e-4)
        full = {"state": {i: {"step": torch.tensor(1.0), "exp_avg": torch.zeros_like(q), "exp_avg_sq": torch.zeros_like(q)}
                          for i, q in enumerate(opt2._params_in_order())}, "param_groups": sd["param_groups"]}
        full["state"][idx] = copy.deepcopy(sd["state"][idx])
        opt2.load_state_dict(full)                                                       # ... and enters
        assert torch.equal(opt2.state[opt2._params_in_order()[idx]]["exp_avg"], stored)
    # an _apply that REPLACES the parameters (torch.__future__ overwrite mode, as .to(device) does for some backends)
    torch.__future__.set_overwrite_module_params_on_conversion(True)
    try:
        old = model.fc1.weight
        model.double()
        assert model.fc1.weight is not old and L.k_channels(model.fc1.weight) == 32
        model.float()
        assert L.k_channels(model.fc1.weight) == 32
    finally:
        torch.__future__.set_overwrite_module_params_on_conversion(False)
    # a mark removed by hand comes back at the next forward (it raises for the CPU tensor, but after marking)
    del model.fc1.weight._pv_k_channels
    x = {"satellite": {"data": torch.zeros(1, 11, 25, 16, 16)}}
    try:
        model(x)
    except Exception:
        pass
    assert L.k_channels(model.fc1.weight) == 32


def test_precision_is_fixed_at_construction():
    import pytest
    from predict_pv_yield_amd.models.conv3d._tower import conv_tower_fc1
    model = Model(**KW, precision="bf16")
    assert model.precision == "bf16"
    with pytest.raises(AttributeError, match="fixed at construction"):
        model.precision = "fp32"
    # and the tower itself refuses the combination the setter used to allow: channels-last weights under the f32 arithmetic
    with pytest.raises(RuntimeError, match="channels-last"):
        conv_tower_fc1(torch.zeros(1, 11, 25, 16, 16), model._conv_layers(), model.fc1, 11, 32, (0, 0, 0), model.cnn_output_size,
                       use_bf16=False, fc1_channels_last=True)
