"""GPU: BASELINE configs[4] — the LitModel of experiments/003_perceiver_processes_single_sat_image_then_rnn.py (Perceiver
per satellite image, then GRU encoder / decoder) at its stated size: 128 x 128 pixel images (a 16 384-position context per
image) x 12 channels + NWP, and the 16-bit attention products the experiment trains with (precision=16).

Oracle: oracle/perceiver_oracle.py (torch-CPU restatement of perceiver_pytorch, absent and unpinned upstream: parity with
the package itself is unpinned; the reference's own tests pin shapes only)."""
import numpy as np
import pytest
import torch

from oracle import perceiver_oracle as po

pytestmark = pytest.mark.gpu


def _mods():
    from predict_pv_yield_amd import hip_ops as K
    from predict_pv_yield_amd import perceiver_functional as PF
    return K, PF


def _rel(a, b):
    return (a - b).norm().item() / (b.norm().item() + 1e-30)


def _attention_reference(q, kv, h, scale, round_bf16):
    b, nq, nk = q.shape[0], q.shape[1], kv.shape[1]
    r = (lambda t: po._RoundBF16.apply(t)) if round_bf16 else (lambda t: t)
    qh = r(q.view(b, nq, h, 64).permute(0, 2, 1, 3))
    kh = r(kv[..., :h * 64].reshape(b, nk, h, 64).permute(0, 2, 1, 3))
    vh = r(kv[..., h * 64:].reshape(b, nk, h, 64).permute(0, 2, 1, 3))
    sim = (qh @ kh.transpose(-1, -2)) * scale
    return (r(sim.softmax(dim=-1)) @ vh).permute(0, 2, 1, 3).reshape(b, nq, h * 64), sim


@pytest.mark.parametrize("b,h,nq,nk", [(2, 1, 128, 16384), (2, 8, 128, 128), (2, 2, 40, 70), (1, 1, 97, 33), (1, 1, 100, 2100)])
def test_bf16_attention_forward_and_backward(b, h, nq, nk, device):
    """pv_attention_fwd_bf16 / pv_attention_bwd_bf16 against torch on the CPU: tight against the reference that rounds the
    same operands to bf16 (q, k, v, p in forward AND backward), loose against the exact-f32 attention."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(b * nq + nk)
    q = torch.randn(b, nq, h * 64, generator=g, requires_grad=True)
    kv = torch.randn(b, nk, 2 * h * 64, generator=g, requires_grad=True)
    scale = 0.125
    ref, sim = _attention_reference(q, kv, h, scale, round_bf16=True)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    qd, kvd = q.detach().to(device), kv.detach().to(device)
    out, lse = K.attention_fwd(qd, kvd, h, scale, bf16_operands=True)
    assert _rel(out.cpu(), ref.detach()) <= 4e-3, _rel(out.cpu(), ref.detach())
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(sim.detach(), dim=-1), rtol=1e-4, atol=1e-4)
    dq, dkv = K.attention_bwd(qd, kvd, out, dout.to(device), lse, h, scale, bf16_operands=True)
    assert _rel(dq.cpu(), q.grad) <= 1.5e-2, _rel(dq.cpu(), q.grad)
    assert _rel(dkv.cpu(), kv.grad) <= 1.5e-2, _rel(dkv.cpu(), kv.grad)
    # against exact f32: bf16 operand rounding, a few 1e-3 norm-wise
    q2, kv2 = q.detach().clone().requires_grad_(True), kv.detach().clone().requires_grad_(True)
    ref32, _ = _attention_reference(q2, kv2, h, scale, round_bf16=False)
    ref32.backward(dout)
    assert _rel(out.cpu(), ref32.detach()) <= 1.5e-2
    assert _rel(dq.cpu(), q2.grad) <= 3e-2 and _rel(dkv.cpu(), kv2.grad) <= 3e-2
    # and the f32 kernels of the same call agree with f32 tightly (the switch really selects different kernels)
    out32, _ = K.attention_fwd(qd, kvd, h, scale)
    assert _rel(out32.cpu(), ref32.detach()) <= 1e-5 and not torch.equal(out32, out)


@pytest.mark.parametrize("b,h,nq,nk", [(2, 1, 128, 16384), (2, 2, 40, 72), (1, 1, 100, 2104)])
def test_attention_with_keys_and_values_stored_as_bf16(b, h, nq, nk, device):
    """pv_attention_*_bf16kv read K / V that are bf16 in memory: the same values the f32-K/V kernels round to on the way into
    LDS, so output, log-sum-exp, dq and dkv are identical bits; and pv_gemm_rows_bf16out_f32 (the projection that writes them)
    equals the f32 product rounded to nearest even."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(nk + nq)
    q = torch.randn(b, nq, h * 64, generator=g).to(device)
    kv = torch.randn(b, nk, 2 * h * 64, generator=g).to(device)
    dout = torch.randn(b, nq, h * 64, generator=g).to(device)
    kv16 = kv.to(torch.bfloat16)
    out, lse = K.attention_fwd(q, kv, h, 0.125, bf16_operands=True)
    out16, lse16 = K.attention_fwd(q, kv16, h, 0.125, bf16_operands=True)
    assert torch.equal(out, out16) and torch.equal(lse, lse16)
    dq, dkv = K.attention_bwd(q, kv, out, dout, lse, h, 0.125, bf16_operands=True)
    dq16, dkv16 = K.attention_bwd(q, kv16, out16, dout, lse16, h, 0.125, bf16_operands=True)
    assert dkv16.dtype == torch.float32 and torch.equal(dq, dq16) and torch.equal(dkv, dkv16)
    acc = torch.ones_like(dkv)
    K.attention_bwd(q, kv16, out16, dout, lse16, h, 0.125, bf16_operands=True, accumulate_dkv_into=acc)
    torch.testing.assert_close(acc, dkv + 1.0, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError, match="bf16_operands"):
        K.attention_fwd(q, kv16, h, 0.125)
    x = torch.randn(b * nk, 38, generator=g).to(device)
    w = (torch.randn(2 * h * 64, 38, generator=g) * 0.2).to(device)
    if x.shape[0] >= 2048:
        assert torch.equal(K.gemm_rows_bf16out(x, w.t()), K.gemm(x, w.t()).to(torch.bfloat16))


def test_bf16_stored_key_value_gradients_feed_the_bf16_operand_products_unchanged(device):
    """dK / dV stored as bf16 by the attention backward (pv_attention_bwd_bf16kv16) are the nearest-even roundings of the f32
    ones, and a bf16-operand GEMM that reads them as they are (PV_GEMM_A_IS_BF16) gives the bits it gives on the f32 ones."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(9)
    b, h, nq, nk = 2, 1, 128, 4096
    q = torch.randn(b, nq, 64, generator=g).to(device)
    kv16 = torch.randn(b, nk, 128, generator=g).to(device).to(torch.bfloat16)
    dout = torch.randn(b, nq, 64, generator=g).to(device)
    out, lse = K.attention_fwd(q, kv16, h, 0.125, bf16_operands=True)
    dq, dkv = K.attention_bwd(q, kv16, out, dout, lse, h, 0.125, bf16_operands=True)
    dq16, dkv16 = K.attention_bwd(q, kv16, out, dout, lse, h, 0.125, bf16_operands=True, dkv_bf16=True)
    assert dkv16.dtype == torch.bfloat16 and torch.equal(dq, dq16) and torch.equal(dkv16, dkv.to(torch.bfloat16))
    w = (torch.randn(128, 38, generator=g) * 0.2).to(device)
    x = torch.randn(b * nk, 38, generator=g).to(device)
    g32, g16 = dkv.view(-1, 128), dkv16.view(-1, 128)
    assert torch.equal(K.gemm(g16, w, bf16_operands=True), K.gemm(g32, w, bf16_operands=True))
    assert torch.equal(K.gemm_splitk(g16.t(), x, bf16_operands=True), K.gemm_splitk(g32.t(), x, bf16_operands=True))
    with pytest.raises(ValueError, match="bf16_operands"):
        K.gemm(g16, w)


@pytest.mark.parametrize("rows,d,kdim", [(70_000, 38, 128), (4097, 38, 128), (33, 64, 64), (100_003, 20, 128)])
def test_layernorm_parameter_gradients_straight_from_the_projection_gradient(device, rows, d, kdim):
    """pv_layernorm_bwd_params_from_proj_bf16 against the two kernels it replaces (d ctx = dKV W as a bf16-operand GEMM, then the
    LayerNorm backward's two column sums): the same products and the same f32 d ctx, summed in another (fixed) order."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.5).to(device)
    lw = (1 + 0.1 * torch.randn(d, generator=g)).to(device)
    lb = (0.1 * torch.randn(d, generator=g)).to(device)
    w = (torch.randn(kdim, d, generator=g) * 0.2).to(device)
    g16 = torch.randn(rows, kdim, generator=g).to(device).to(torch.bfloat16)
    _, mean, rstd = K.layernorm_fwd(x, lw, lb, 1e-5)
    dctx = K.gemm(g16, w, bf16_operands=True)
    _, dw_ref, db_ref = K.layernorm_bwd(x, lw, dctx, mean, rstd, need_dx=False)
    assert K.layernorm_bwd_params_from_proj_supported(g16, w, x)
    dw, db = K.layernorm_bwd_params_from_proj(g16, w, x, mean, rstd)
    # f64 reference of the same sums: both forms must sit within f32 summation error of it
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    dw64, db64 = (dctx.double() * xh).sum(0), dctx.double().sum(0)
    scale_w = (dctx.double() * xh).abs().sum(0).max().item()
    scale_b = dctx.double().abs().sum(0).max().item()
    assert (dw.double() - dw64).abs().max().item() <= 2e-6 * scale_w
    assert (db.double() - db64).abs().max().item() <= 2e-6 * scale_b
    assert (dw - dw_ref).abs().max().item() <= 4e-6 * scale_w and (db - db_ref).abs().max().item() <= 4e-6 * scale_b
    # accumulate_into adds
    acc = (dw.clone(), db.clone())
    K.layernorm_bwd_params_from_proj(g16, w, x, mean, rstd, accumulate_into=acc)
    assert torch.allclose(acc[0], 2 * dw, rtol=1e-6, atol=1e-6 * scale_w) and torch.allclose(acc[1], 2 * db, rtol=1e-6, atol=1e-6 * scale_b)
    # run to run: a fixed summation order
    dw2, db2 = K.layernorm_bwd_params_from_proj(g16, w, x, mean, rstd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("rows,d", [(70_000, 38), (4097, 38), (31, 64), (100_003, 20)])
def test_context_backward_in_one_pass(device, rows, d):
    """pv_context_bwd_bf16 (dW_kv, d gamma, d beta from one pass over dK | dV and x) against the weight-gradient GEMM on the
    LayerNorm kernel's output and the LayerNorm-parameter kernel: the same bf16 operands (ctx re-formed in LDS with the LayerNorm
    kernel's expression), f32 sums in another fixed order."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(rows * 3 + d)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.5).to(device)
    lw = (1 + 0.1 * torch.randn(d, generator=g)).to(device)
    lb = (0.1 * torch.randn(d, generator=g)).to(device)
    w = (torch.randn(128, d, generator=g) * 0.2).to(device)
    g16 = torch.randn(rows, 128, generator=g).to(device).to(torch.bfloat16)
    ctx, mean, rstd = K.layernorm_fwd(x, lw, lb, 1e-5)
    dw_ref = K.gemm_splitk(g16.t(), ctx, bf16_operands=True)
    dlw_ref, dlb_ref = K.layernorm_bwd_params_from_proj(g16, w, x, mean, rstd)
    assert K.context_bwd_supported(g16, w, x)
    dw, dlw, dlb = K.context_bwd(g16, w, x, mean, rstd, lw, lb)
    # f64 sums of the same bf16-rounded operands
    dw64 = g16.double().t() @ ctx.to(torch.bfloat16).double()
    scale = (g16.double().abs().t() @ ctx.to(torch.bfloat16).double().abs()).max().item()
    assert dw.shape == (128, d)
    assert (dw.double() - dw64).abs().max().item() <= 2e-6 * scale
    assert (dw_ref.double() - dw64).abs().max().item() <= 2e-6 * scale
    sw, sb = dlw_ref.abs().max().item() + 1e-6, dlb_ref.abs().max().item() + 1e-6
    bound = 3e-5 * (rows ** 0.5)      # sums of `rows` signed terms of order 1, f32 partial sums in two different orders
    assert (dlw - dlw_ref).abs().max().item() <= bound * 1e-2 * max(1.0, sw) and (dlb - dlb_ref).abs().max().item() <= bound * 1e-2 * max(1.0, sb)
    dw2, dlw2, dlb2 = K.context_bwd(g16, w, x, mean, rstd, lw, lb)
    assert torch.equal(dw, dw2) and torch.equal(dlw, dlw2) and torch.equal(dlb, dlb2)
    acc_w, acc_ln = dw.clone(), (dlw.clone(), dlb.clone())
    K.context_bwd(g16, w, x, mean, rstd, lw, lb, accumulate_kv_into=acc_w, accumulate_ln_into=acc_ln)
    assert torch.allclose(acc_w, 2 * dw, rtol=1e-6, atol=1e-6 * scale)
    assert torch.allclose(acc_ln[0], 2 * dlw, rtol=1e-5, atol=1e-5 * sw) and torch.allclose(acc_ln[1], 2 * dlb, rtol=1e-5, atol=1e-5 * sb)


def test_exp003_with_the_context_backward_in_one_pass(device):
    """perceiver_functional.ONE_PASS_CONTEXT_BACKWARD (the default): same output, loss and gradients as with the weight-gradient
    GEMM + LayerNorm-parameter kernel, except to_kv's weight and norm_context's parameters of the cross-attention blocks, which
    are the same sums in another order."""
    from predict_pv_yield_amd import perceiver_functional as PF
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    batch = make_fake_exp003_batch(2, 64, torch.Generator().manual_seed(7))
    results = []
    for one_pass in (True, False):
        PF.ONE_PASS_CONTEXT_BACKWARD = one_pass
        try:
            _, model = _pair(device, "bf16", seed=3)
            y = model(_to(batch, device))
            loss = model.training_step(_to(batch, device), 0)
            loss.backward()
            results.append((y.detach(), loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            PF.ONE_PASS_CONTEXT_BACKWARD = True
    (y1, l1, g1), (y0, l0, g0) = results
    assert torch.equal(y1, y0) and torch.equal(l1, l0) and g1.keys() == g0.keys()
    n_close = 0
    for name in g1:
        if "norm_context" in name or ("to_kv" in name and not torch.equal(g1[name], g0[name])):
            n_close += 1
            scale = g0[name].abs().max().item()
            assert (g1[name] - g0[name]).abs().max().item() <= 1e-4 * scale + 1e-9, name
        else:
            assert torch.equal(g1[name], g0[name]), name
    assert 4 <= n_close <= 6      # norm_context weight + bias and to_kv of the two cross-attention blocks


@pytest.mark.parametrize("shape", [(5, 16384, 38), (70_001, 38), (3, 16384, 38), (4096, 48), (4099, 20)])
def test_context_forward_in_one_pass_against_the_layernorm_and_projection_kernels(device, shape):
    """pv_context_fwd_bf16 (x -> LayerNorm -> to_kv, K | V stored as bf16, the normalised context only in registers) against
    pv_layernorm_fwd_f32 + pv_gemm_rows_bf16out_f32 with bf16 operands.  Where the LayerNorm runs its row-per-thread kernel
    (>= 65 536 short rows: the cross-attention contexts) the statistics are the same serial sums, the operands the same roundings,
    the matrix instructions in the same order: identical bits.  Elsewhere the LayerNorm kernel sums a row across a wave: the
    statistics agree to rounding and K | V to a bf16 unit in the last place on a few values."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(sum(shape))
    d = shape[-1]
    x = (torch.randn(*shape, generator=g) * 3 + 1).to(device)
    lw = (1 + 0.1 * torch.randn(d, generator=g)).to(device)
    lb = (0.1 * torch.randn(d, generator=g)).to(device)
    w = (torch.randn(128, d, generator=g) * 0.2).to(device)
    ctx, mean_ref, rstd_ref = K.layernorm_fwd(x, lw, lb, 1e-5)
    kv_ref = K.gemm_rows_bf16out(ctx.view(-1, d), w.t(), bf16_operands=True)
    assert K.context_fwd_supported(x, w)
    kv16, mean, rstd = K.context_fwd(x, lw, lb, w, 1e-5)
    assert kv16.shape == shape[:-1] + (128,) and kv16.dtype == torch.bfloat16
    rows = x.numel() // d
    if rows >= 65536 and d % 8 != 0:
        assert torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
        assert torch.equal(kv16.view(-1, 128), kv_ref)
    else:
        assert torch.allclose(mean, mean_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rstd, rstd_ref, rtol=1e-5)
        a, b_ = kv16.view(-1, 128).float(), kv_ref.float()
        assert (a - b_).abs().max().item() <= 2.0 ** -6 * b_.abs().max().item()
        assert (a != b_).float().mean().item() < 0.02


def test_exp003_with_the_context_forward_in_one_pass(device):
    """perceiver_functional.ONE_PASS_CONTEXT_FORWARD (the default) changes no bit of output, loss or any gradient."""
    from predict_pv_yield_amd import perceiver_functional as PF
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    batch = make_fake_exp003_batch(2, 64, torch.Generator().manual_seed(8))
    results = []
    for one_pass in (True, False):
        PF.ONE_PASS_CONTEXT_FORWARD = one_pass
        try:
            _, model = _pair(device, "bf16", seed=4)
            y = model(_to(batch, device))
            loss = model.training_step(_to(batch, device), 0)
            loss.backward()
            results.append([y.detach(), loss.detach()] + [p.grad.clone() for p in model.parameters() if p.grad is not None])
        finally:
            PF.ONE_PASS_CONTEXT_FORWARD = True
    assert len(results[0]) == len(results[1]) > 10
    for a, b_ in zip(*results):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("b,p,d1,d2", [(5, 16384, 12, 26), (3, 4099, 2, 36), (70, 1000, 30, 8)])
def test_context_kernels_read_channels_and_position_features_from_two_tensors(device, b, p, d1, d2):
    """pv_context_fwd_bf16 / pv_context_bwd_bf16 with the context rows given as channels [b, P, d1] + position features [P, d2]:
    the bits they give on the concatenated [b, P, d1 + d2] tensor (which Perceiver.forward would build with torch.cat)."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(b * p + d1)
    d = d1 + d2
    chans = (torch.randn(b, p, d1, generator=g) * 3 + 1).to(device)
    pos = torch.randn(p, d2, generator=g).to(device)
    full = torch.cat((chans, pos.unsqueeze(0).expand(b, p, d2)), dim=-1).contiguous()
    lw = (1 + 0.1 * torch.randn(d, generator=g)).to(device)
    lb = (0.1 * torch.randn(d, generator=g)).to(device)
    w = (torch.randn(128, d, generator=g) * 0.2).to(device)
    assert K.context_fwd_supported(chans, w, pos) and K.context_fwd_supported(full, w)
    kv_a, mean_a, rstd_a = K.context_fwd(chans, lw, lb, w, 1e-5, x2=pos)
    kv_b, mean_b, rstd_b = K.context_fwd(full, lw, lb, w, 1e-5)
    assert torch.equal(kv_a, kv_b) and torch.equal(mean_a, mean_b) and torch.equal(rstd_a, rstd_b)
    g16 = torch.randn(b * p, 128, generator=g).to(device).to(torch.bfloat16)
    got = K.context_bwd(g16, w, chans, mean_a, rstd_a, lw, lb, x2=pos)
    ref = K.context_bwd(g16, w, full, mean_a, rstd_a, lw, lb)
    for a_, b_ in zip(got, ref):
        assert torch.equal(a_, b_)


def test_exp003_without_the_concatenated_context(device):
    """perceiver_core.SPLIT_CONTEXT (the default where every cross-attention block runs the one-pass context kernels): the
    [b, positions, channels + fourier] tensor is not built; output, loss and every gradient keep their bits."""
    from predict_pv_yield_amd.models.perceiver import perceiver_core
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    batch = make_fake_exp003_batch(2, 64, torch.Generator().manual_seed(9))
    results = []
    for split in (True, False):
        perceiver_core.SPLIT_CONTEXT = split
        try:
            _, model = _pair(device, "bf16", seed=5)
            y = model(_to(batch, device))
            loss = model.training_step(_to(batch, device), 0)
            loss.backward()
            results.append([y.detach(), loss.detach()] + [p.grad.clone() for p in model.parameters() if p.grad is not None])
        finally:
            perceiver_core.SPLIT_CONTEXT = True
    assert len(results[0]) == len(results[1]) > 10
    for a, b_ in zip(*results):
        assert torch.equal(a, b_)


def test_exp003_with_the_context_norm_inside_the_attention_node(device):
    """perceiver_core.NORM_CONTEXT_IN_THE_ATTENTION_NODE (the default for a context one layer consumes): output and loss are
    the same bits as with norm_context as its own node, every gradient but norm_context's too; norm_context's weight / bias
    gradients are the same sums in another order."""
    from predict_pv_yield_amd.models.perceiver import perceiver_core
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    batch = make_fake_exp003_batch(2, 64, torch.Generator().manual_seed(6))
    results = []
    for inside in (True, False):
        perceiver_core.NORM_CONTEXT_IN_THE_ATTENTION_NODE = inside
        try:
            _, model = _pair(device, "bf16", seed=2)
            y = model(_to(batch, device))
            loss = model.training_step(_to(batch, device), 0)
            loss.backward()
            results.append((y.detach(), loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            perceiver_core.NORM_CONTEXT_IN_THE_ATTENTION_NODE = True
    (y1, l1, g1), (y0, l0, g0) = results
    assert torch.equal(y1, y0) and torch.equal(l1, l0) and g1.keys() == g0.keys() and len(g1) > 10
    n_norm = 0
    for name in g1:
        if "to_kv" in name and not torch.equal(g1[name], g0[name]):     # (one-pass context backward: to_kv's sum has another order too)
            assert (g1[name] - g0[name]).abs().max().item() <= 1e-4 * g0[name].abs().max().item() + 1e-9, name
        elif "norm_context" in name:
            n_norm += 1
            scale = g0[name].abs().max().item()
            assert (g1[name] - g0[name]).abs().max().item() <= 1e-4 * scale + 1e-9, name
        else:
            assert torch.equal(g1[name], g0[name]), name
    assert n_norm == 4      # two cross-attention blocks, weight and bias each


def test_exp003_is_the_same_model_with_keys_and_values_stored_as_bf16(device):
    """operand_dtype="bf16": storing the projected context as bf16 (perceiver_core.KV_STORED_AS_BF16, the default) changes where
    K / V are rounded, not to what: output, loss and every gradient equal those of the f32-stored form bit for bit."""
    from predict_pv_yield_amd.models.perceiver import perceiver_core
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    batch = make_fake_exp003_batch(2, 64, torch.Generator().manual_seed(5))
    results = []
    for stored in (True, False):
        perceiver_core.KV_STORED_AS_BF16 = stored
        # (norm_context as its own node on both sides: inside the attention node its two sums have another order)
        perceiver_core.NORM_CONTEXT_IN_THE_ATTENTION_NODE = False
        try:
            _, model = _pair(device, "bf16", seed=1)
            y = model(_to(batch, device))
            loss = model.training_step(_to(batch, device), 0)
            loss.backward()
            results.append([y.detach(), loss.detach()] + [p.grad.clone() for p in model.parameters() if p.grad is not None])
        finally:
            perceiver_core.KV_STORED_AS_BF16 = True
            perceiver_core.NORM_CONTEXT_IN_THE_ATTENTION_NODE = True
    assert len(results[0]) == len(results[1]) > 10
    for a, b_ in zip(*results):
        assert torch.equal(a, b_)


def _pair(device, operand_dtype, seed=0):
    from predict_pv_yield_amd.models.perceiver.exp003 import LitModel
    torch.manual_seed(seed)
    oracle = po.OracleExp003LitModel()
    model = LitModel(operand_dtype=operand_dtype)
    assert list(model.state_dict().keys()) == list(oracle.state_dict().keys())
    model.load_state_dict(oracle.state_dict())
    return oracle, model.to(device)


def _to(batch, device):
    return {k: v.to(device) for k, v in batch.items()}


def test_exp003_train_step_replays_as_a_hip_graph(device):
    """graphs.GraphedTrainStep on experiments/003's LitModel (operand_dtype="bf16": the one-pass context kernels, the tied-gradient
    bookkeeping, ~600 launches per step): the replayed step gives the eager step's losses bit for bit over several batches."""
    from predict_pv_yield_amd.graphs import GraphedTrainStep
    from predict_pv_yield_amd.models.perceiver.exp003 import LitModel, make_fake_exp003_batch
    from predict_pv_yield_amd.optim import HipAdam
    batches = [_to(make_fake_exp003_batch(2, 64, torch.Generator().manual_seed(s)), device) for s in range(3)]

    def make(capturable):
        torch.manual_seed(11)
        model = LitModel(operand_dtype="bf16").to(device)
        return model, HipAdam(model.parameters(), lr=0.0005, capturable=capturable)

    model_e, opt_e = make(False)
    model_g, opt_g = make(True)
    step = GraphedTrainStep(model_g, opt_g, batches[0], warmup=2)
    try:
        for _ in range(2):      # the helper's warm-up steps are training steps: the same two on the eager side
            opt_e.zero_grad(set_to_none=True)
            model_e.training_step(batches[0], 0).backward()
            opt_e.step()
        for i in range(5):
            opt_e.zero_grad(set_to_none=True)
            loss = model_e.training_step(batches[i % 3], 0)
            loss.backward()
            opt_e.step()
            assert float(step(batches[i % 3])) == float(loss), f"step {i}"
        for p, q in zip(model_g.parameters(), model_e.parameters()):
            assert torch.equal(p, q)
    finally:
        step.close()


def test_exp003_structure_and_batch_contract():
    from predict_pv_yield_amd.models.perceiver import exp003
    m = exp003.LitModel()
    assert exp003.TOTAL_SEQ_LEN == 19 and exp003.NWP_SIZE == 40
    assert len(m.perceiver.layers) == 2 and m.perceiver.layers[0][0] is not m.perceiver.layers[1][0]     # depth 2, NOT tied
    assert m.perceiver.layers[0][0].fn.to_kv.weight.shape == (128, 38)                                   # 12 ch + 26 Fourier
    assert m.encoder_rnn.input_size == 8 + 4 + 1 + 40 and m.decoder_rnn.input_size == 8 + 4 + 40 and m.encoder_rnn.num_layers == 2
    assert m.pv_system_id_embedding.num_embeddings == 940
    b = exp003.make_fake_exp003_batch(3, 16, torch.Generator().manual_seed(0))
    assert b["sat_data"].shape == (3, 19, 16, 16, 12) and b["nwp"].shape == (3, 10, 19, 2, 2) and b["pv_yield"].shape == (3, 19)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(b)


@pytest.mark.parametrize("operand_dtype", ["f32", "bf16"])
def test_exp003_at_128px_forward_loss_and_gradients(device, operand_dtype):
    """The stated config: 128 x 128 pixels => every one of the B * 19 images is a 16 384-position context."""
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    oracle, model = _pair(device, operand_dtype)
    batch = make_fake_exp003_batch(2, 128, torch.Generator().manual_seed(3))
    po.EMULATE_BF16_ATTENTION = po.EMULATE_BF16_LINEAR = operand_dtype == "bf16"
    try:
        y_ref = oracle(batch)
        mse_ref, nmae_ref = oracle.losses(batch)
        nmae_ref.backward()
    finally:
        po.EMULATE_BF16_ATTENTION = po.EMULATE_BF16_LINEAR = False
    y = model(_to(batch, device))
    assert y.shape == (2, 12)
    tol = 2e-4 if operand_dtype == "f32" else 3e-3
    assert (y.detach().cpu() - y_ref.detach()).abs().max().item() <= tol, (y.detach().cpu() - y_ref.detach()).abs().max().item()
    loss = model.training_step(_to(batch, device), 0)
    assert abs(float(loss) - float(nmae_ref)) <= tol * max(1.0, abs(float(nmae_ref)))
    assert abs(float(model._logged["MSE/Train"]) - float(mse_ref)) <= 2 * tol
    loss.backward()
    worst = 0.0
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        assert p.grad is not None, k
        if q.grad.norm().item() < 1e-12:
            continue
        rel = _rel(p.grad.cpu(), q.grad)
        worst = max(worst, rel)
        assert rel <= (1e-4 if operand_dtype == "f32" else 5e-3), (k, rel)     # measured 9e-6 / 9e-4
    print(f"[exp003 {operand_dtype}] worst gradient rel = {worst:.3e}")


def test_exp003_train_steps_and_trainer(device, tmp_path, monkeypatch):
    """Three Adam steps follow the oracle (f32 operands), and the experiment's trainer call runs through the shim
    (precision=16 selects nothing by itself: the model's operand_dtype does) on 64-pixel images."""
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.models.perceiver.exp003 import FakeExp003Dataset, LitModel, make_fake_exp003_batch
    oracle, model = _pair(device, "f32", seed=1)
    batch = make_fake_exp003_batch(2, 32, torch.Generator().manual_seed(5))
    ref_opt = torch.optim.Adam(oracle.parameters(), lr=0.0005)
    opt = model.configure_optimizers()
    ref_losses, losses = [], []
    for _ in range(3):
        ref_opt.zero_grad()
        _, nmae = oracle.losses(batch)
        nmae.backward()
        ref_opt.step()
        ref_losses.append(float(nmae))
        opt.zero_grad()
        loss = model.training_step(_to(batch, device), 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-3)
    monkeypatch.chdir(tmp_path)
    m2 = LitModel(operand_dtype="bf16")
    loader = torch.utils.data.DataLoader(FakeExp003Dataset(batch_size=2, image_size_pixels=64, length=2), batch_size=None)
    trainer = pl.Trainer(gpus=1, max_epochs=1, precision=16)
    trainer.fit(m2, loader, loader)
    assert np.isfinite(trainer.callback_metrics["NMAE/Train_epoch"]) and np.isfinite(trainer.callback_metrics["NMAE/Validation_epoch"])


def test_trainer_replays_the_train_step_as_a_hip_graph(device, tmp_path, monkeypatch):
    """Trainer(hip_graph=True): three eager steps, then the captured step replayed per batch (two epochs, validation in between):
    the parameters, the optimiser's step count and every logged metric equal the eager Trainer's."""
    from predict_pv_yield_amd import lightning as pl
    from predict_pv_yield_amd.models.perceiver.exp003 import FakeExp003Dataset, LitModel
    monkeypatch.chdir(tmp_path)
    results = []
    for graph in (False, True):
        torch.manual_seed(21)
        model = LitModel(operand_dtype="bf16")
        train = torch.utils.data.DataLoader(FakeExp003Dataset(batch_size=2, image_size_pixels=64, length=6), batch_size=None)
        val = torch.utils.data.DataLoader(FakeExp003Dataset(batch_size=2, image_size_pixels=64, length=2, seed=99), batch_size=None)
        trainer = pl.Trainer(gpus=1, max_epochs=2, precision=16, hip_graph=graph, log_every_n_steps=1)
        trainer.fit(model, train, val)
        assert (trainer._graph_step is None)      # released at the end of fit
        opt = trainer.optimizers[0]
        steps = {int(st["step"].item()) for st in opt.state_dict()["state"].values()}
        results.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, dict(trainer.callback_metrics), steps,
                        getattr(opt, "capturable", False)))
    (p0, m0, s0, c0), (p1, m1, s1, c1) = results
    assert not c0 and c1 and s0 == s1 == {12}
    assert p0.keys() == p1.keys()
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    assert m0.keys() == m1.keys() and len(m0) >= 2
    for k in m0:
        assert m0[k] == m1[k], (k, m0[k], m1[k])


def test_exp003_config_composes_and_trains(device, tmp_path, monkeypatch):
    import os
    from predict_pv_yield_amd import hydra_lite as H
    from predict_pv_yield_amd.training import train
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    cfg = H.compose(os.path.join(root, "configs"), "config",
                    ["experiment=exp003_perceiver", "datamodule.batch_size=2", "datamodule.image_size_pixels=32",
                     "datamodule.n_train_data=2", "datamodule.n_val_data=1", "trainer.max_epochs=1",
                     "optimized_metric=NMAE/Validation_epoch"])
    assert cfg.model._target_.endswith("exp003.LitModel") and cfg.trainer.precision == 16
    assert np.isfinite(train(cfg))


def test_no_grad_forward_does_not_register_untied_gradients(device):
    """a validation / sanity-check forward (torch.no_grad) must not count as an application of a parameter --
    otherwise every once-applied parameter looks tied from the first validation pass on, its gradient is kept in _TIED and
    AccumulateGrad has to copy it.  exp-003's perceiver is NOT weight-tied: nothing may be registered, ever."""
    from predict_pv_yield_amd import perceiver_functional as PF
    from predict_pv_yield_amd.models.perceiver.exp003 import make_fake_exp003_batch
    _, model = _pair(device, "f32")
    batch = _to(make_fake_exp003_batch(2, 32, torch.Generator().manual_seed(5)), device)
    seen = []
    keep = PF._tied_keep
    PF._tied_keep = lambda key, grad: (seen.append(key), keep(key, grad))[1]
    try:
        for _ in range(2):
            with torch.no_grad():
                model(batch)                                   # validation-style forward: builds no graph
            model.zero_grad(set_to_none=True)
            model.training_step(batch, 0).backward()
            assert all(k is None for k in seen), "a once-applied parameter was registered as tied"
            assert not PF._TIED
            assert all(getattr(p, "_pv_uses", 0) == 0 for p in model.parameters()), "use counts must be consumed by the backward"
    finally:
        PF._tied_keep = keep
