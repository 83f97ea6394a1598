"""pv_dense_chain_{fwd,bwd}_f32: the small head (fc2 -> fc3 -> fc4, predict_pv_yield/models/conv3d/model.py:126,151-156) as one
launch each way, against torch on the CPU in float64 (rtol 1e-5: exact f32 products, f32 accumulation in a fixed order) and
against the per-layer kernels it replaces (pv_linear_{fwd,bwd}_f32)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (m, k0, [(n, relu), ...])
    (32, 128, [(128, True), (64, True), (6, False)]),      # the benched head
    (32, 128, [(128, True), (64, True), (24, False)]),     # forecast_minutes = 120
    (4, 16, [(16, True), (16, True), (12, False)]),        # tests/configs/model/conv3d.yaml of the reference
    (7, 100, [(33, True), (97, False)]),
    (1, 5, [(3, True)]),
    (32, 128, [(128, True), (128, True), (128, True)]),
    (19, 31, [(65, False), (2, True), (128, False)]),
]


def _make(m, k0, spec, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k0, generator=g).clamp_min(0)             # the head's input is a ReLU output
    layers, k = [], k0
    for n, relu in spec:
        layers.append((torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g) * 0.1, relu))
        k = n
    dy = torch.randn(m, k, generator=g)
    return x, layers, dy


def _reference(x, layers, dy):
    x = x.double().requires_grad_(True)
    ws = [w.double().requires_grad_(True) for w, _, _ in layers]
    bs = [b.double().requires_grad_(True) for _, b, _ in layers]
    h, ys = x, []
    for w, b, (_, _, relu) in zip(ws, bs, layers):
        h = h @ w.t() + b
        if relu:
            h = torch.relu(h)
        ys.append(h)
    h.backward(dy.double())
    return [y.detach() for y in ys], x.grad, [w.grad for w in ws], [b.grad for b in bs]


def _close(a, b, what, rtol=1e-5):
    a, b = a.detach().double().cpu(), b.double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-30
    assert err <= rtol * scale + 1e-12, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("case", range(len(SHAPES)))
def test_chain_against_float64(device, case):
    from predict_pv_yield_amd import hip_ops as K
    m, k0, spec = SHAPES[case]
    x, layers, dy = _make(m, k0, spec, seed=100 + case)
    ys_ref, dx_ref, dw_ref, db_ref = _reference(x, layers, dy)
    xd = x.to(device)
    ld = [(w.to(device), b.to(device), r) for w, b, r in layers]
    assert K.dense_chain_supported(xd, ld)
    ys = K.dense_chain_fwd(xd, ld)
    for i, (y, yr) in enumerate(zip(ys, ys_ref)):
        _close(y, yr, f"y[{i}]")
    dx, dws, dbs = K.dense_chain_bwd(xd, [w for w, _, _ in ld], [r for _, _, r in ld], ys, dy.to(device))
    _close(dx, dx_ref, "dx")
    for i in range(len(layers)):
        _close(dws[i], dw_ref[i], f"dw[{i}]")
        _close(dbs[i], db_ref[i], f"db[{i}]")
    # dx not wanted: the weight gradients do not change
    _, dws2, dbs2 = K.dense_chain_bwd(xd, [w for w, _, _ in ld], [r for _, _, r in ld], ys, dy.to(device), need_dx=False)
    assert all(torch.equal(a, b) for a, b in zip(dws + dbs, dws2 + dbs2))
    # deterministic: a second launch gives the same bits
    ys3 = K.dense_chain_fwd(xd, ld)
    assert all(torch.equal(a, b) for a, b in zip(ys, ys3))


def test_chain_against_the_per_layer_kernels_through_autograd(device):
    """functional.dense_chain_f32 (one launch each way) vs functional.linear_f32 three times: same values to f32 rounding, and
    the weight gradients -- whose contraction runs over the batch rows in the per-layer kernel's order -- bit for bit when the
    incoming gradients agree bit for bit (last layer)."""
    from predict_pv_yield_amd import functional as Fn
    m, k0, spec = SHAPES[0]
    x, layers, dy = _make(m, k0, spec, seed=7)
    outs = {}
    for fused in (True, False):
        Fn.DENSE_CHAIN = fused
        try:
            xd = x.to(device).requires_grad_(True)
            ld = [(w.to(device).requires_grad_(True), b.to(device).requires_grad_(True), r) for w, b, r in layers]
            y = Fn.dense_chain_f32(xd, ld)
            y.backward(dy.to(device))
            outs[fused] = (y.detach(), xd.grad, [w.grad for w, _, _ in ld], [b.grad for _, b, _ in ld])
        finally:
            Fn.DENSE_CHAIN = True
    (ya, dxa, dwa, dba), (yb, dxb, dwb, dbb) = outs[True], outs[False]
    _close(ya, yb.cpu(), "y", rtol=2e-6)
    _close(dxa, dxb.cpu(), "dx", rtol=2e-6)
    for i in range(3):
        _close(dwa[i], dwb[i].cpu(), f"dw[{i}]", rtol=2e-6)
        _close(dba[i], dbb[i].cpu(), f"db[{i}]", rtol=2e-6)


def test_shapes_outside_the_chain_take_the_per_layer_path(device):
    from predict_pv_yield_amd import functional as Fn
    from predict_pv_yield_amd import hip_ops as K
    g = torch.Generator().manual_seed(3)
    x = torch.randn(40, 64, generator=g).to(device)          # 40 rows: more than one block
    ld = [(torch.randn(16, 64, generator=g).to(device), torch.zeros(16, device=device), True)]
    assert not K.dense_chain_supported(x, ld)
    y = Fn.dense_chain_f32(x, ld)
    ref = torch.relu(x.cpu().double() @ ld[0][0].cpu().double().t())
    _close(y, ref, "fallback y")
