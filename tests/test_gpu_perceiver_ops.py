"""GPU parity of the Perceiver-path kernels (strided batched f32-MFMA GEMM, LayerNorm, scaled softmax, GEGLU, mean) and of
their autograd bindings against torch CPU ops on seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mods():
    from predict_pv_yield_amd import hip_ops as K
    from predict_pv_yield_amd import perceiver_functional as PF
    return K, PF


@pytest.mark.parametrize("shape", [((), 5, 37, 7), ((), 300, 70, 130), ((3,), 128, 64, 200), ((2, 4), 33, 17, 65),
                                   ((1, 8), 128, 64, 128), ((), 1, 1000, 40)])
def test_gemm_strided_views(shape, device):
    K, _ = _mods()
    batch, m, k, n = shape
    g = torch.Generator().manual_seed(m * n + k)
    a = torch.randn(batch + (m, k), generator=g)
    b = torch.randn(batch + (k, n), generator=g)
    bias = torch.randn(n, generator=g)
    ad, bd = a.to(device), b.to(device)
    ref = a @ b
    tol = dict(rtol=1e-4, atol=1e-4 * np.sqrt(k))
    torch.testing.assert_close(K.gemm(ad, bd).cpu(), ref, **tol)
    torch.testing.assert_close(K.gemm(ad, bd, bias=bias.to(device), relu=True).cpu(), F.relu(ref + bias), **tol)
    # transposed operands as views (no copies): a stored [.., k, m], b stored [.., n, k]
    at = a.transpose(-1, -2).contiguous().to(device).transpose(-1, -2)
    bt = b.transpose(-1, -2).contiguous().to(device).transpose(-1, -2)
    torch.testing.assert_close(K.gemm(at, bt).cpu(), ref, **tol)
    torch.testing.assert_close(K.gemm(ad, bt).cpu(), ref, **tol)
    # column-sliced operand (k / v halves of a kv projection) and output written into a column slice
    wide = torch.randn(batch + (k, 2 * n), generator=g).to(device)
    out = torch.zeros(batch + (m, 2 * n), device=device)
    K.gemm(ad, wide[..., n:], out=out[..., :n])
    torch.testing.assert_close(out[..., :n].cpu(), a @ wide[..., n:].cpu(), **tol)
    assert float(out[..., n:].abs().max()) == 0.0


@pytest.mark.parametrize("off,ld_pad", [(0, 0), (1, 0), (2, 3), (4, 4), (0, 1)])
def test_gemm_quad_staging_falls_back_on_misaligned_views(off, ld_pad, device):
    """gemm_bf16x3's staging takes one 16-byte load per four elements only for whole tiles with a 16-byte aligned base, a
    unit stride and the other stride a multiple of 4; views that break any of these (column offset, padded leading
    dimension) must take the per-element path and give the same product.  256 x 192 outputs = whole tiles only, K = 96."""
    K, _ = _mods()
    m, k, n = 256, 96, 192
    g = torch.Generator().manual_seed(off * 10 + ld_pad)
    a_store = torch.randn(m, k + off + ld_pad, generator=g)
    b_store = torch.randn(k, n + off + ld_pad, generator=g)
    a, b = a_store[:, off:off + k], b_store[:, off:off + n]
    ref = (a.double() @ b.double()).float()
    ad, bd = a_store.to(device)[:, off:off + k], b_store.to(device)[:, off:off + n]
    tol = dict(rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(K.gemm(ad, bd).cpu(), ref, **tol)
    # the same operands stored transposed (m / n contiguous): the MC layouts of both operands
    at = a_store.t().contiguous().to(device)[off:off + k, :].t()
    bt = b_store.t().contiguous().to(device)[off:off + n, :].t()
    torch.testing.assert_close(K.gemm(at, bt).cpu(), ref, **tol)
    torch.testing.assert_close(K.gemm(ad, bt).cpu(), ref, **tol)
    torch.testing.assert_close(K.gemm(at, bd).cpu(), ref, **tol)


def test_gemm_splitk_weight_gradient_shape(device):
    K, _ = _mods()
    g = torch.Generator().manual_seed(4)
    dy, x = torch.randn(70000, 48, generator=g), torch.randn(70000, 37, generator=g)
    got = K.gemm_splitk(dy.to(device).t(), x.to(device)).cpu()
    ref = (dy.double().t() @ x.double()).float()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=2e-2)
    assert torch.equal(got, K.gemm_splitk(dy.to(device).t(), x.to(device)).cpu())     # deterministic


def test_gemm_broadcast_batch(device):
    K, _ = _mods()
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(20, 16, generator=g)
    w = torch.randn(3, 16, 24, generator=g)
    got = K.gemm(lat.to(device).unsqueeze(0).expand(3, 20, 16), w.to(device)).cpu()    # stride-0 batch
    torch.testing.assert_close(got, lat @ w, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,d", [(5, 37), (1000, 64), (7, 200), (4099, 64), (70001, 38), (65536, 12), (66000, 6)])
def test_layernorm(rows, d, device):
    """(the last three take the thread-per-row kernels for short rows: d <= 64, d % 8 != 0, >= 65 536 rows)"""
    _, PF = _mods()
    g = torch.Generator().manual_seed(rows + d)
    x = (torch.randn(rows, d, generator=g) * 3 + 1).requires_grad_(True)
    w = torch.randn(d, generator=g).requires_grad_(True)
    b = torch.randn(d, generator=g).requires_grad_(True)
    ref = F.layer_norm(x, (d,), w, b)
    dy = torch.randn(rows, d, generator=g)
    ref.backward(dy)
    xd, wd, bd = (t.detach().to(device).requires_grad_(True) for t in (x, w, b))
    y = PF.layer_norm(xd, wd, bd)
    y.backward(dy.to(device))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("rows,n", [(6, 128), (3, 4096), (2, 5000), (5, 37)])
def test_softmax_scaled(rows, n, device):
    _, PF = _mods()
    g = torch.Generator().manual_seed(rows * n)
    x = (torch.randn(rows, n, generator=g) * 4).requires_grad_(True)
    scale = 0.125
    ref = (x * scale).softmax(dim=-1)
    dy = torch.randn(rows, n, generator=g)
    ref.backward(dy)
    xd = x.detach().to(device).requires_grad_(True)
    p = PF.softmax_scaled_(xd * 1.0, scale)          # in place on the temporary
    p.backward(dy.to(device))
    torch.testing.assert_close(p.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-7)


def test_geglu_and_mean(device):
    _, PF = _mods()
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(50, 2 * 96, generator=g) * 2).requires_grad_(True)
    a, gate = x.chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    dy = torch.randn(50, 96, generator=g)
    ref.backward(dy)
    xd = x.detach().to(device).requires_grad_(True)
    y = PF.geglu(xd)
    y.backward(dy.to(device))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-5, atol=5e-6)     # erff vs the CPU erf
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=5e-6)
    z = torch.randn(4, 9, 13, generator=g, requires_grad=True)
    zr = z.mean(dim=1)
    dz = torch.randn(4, 13, generator=g)
    zr.backward(dz)
    zd = z.detach().to(device).requires_grad_(True)
    m = PF.mean_axis1(zd)
    m.backward(dz.to(device))
    torch.testing.assert_close(m.detach().cpu(), zr.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(zd.grad.cpu(), z.grad, rtol=1e-6, atol=1e-7)


def test_linear_rows_and_matmul_autograd(device):
    _, PF = _mods()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 700, 37, generator=g, requires_grad=True)
    w = (torch.randn(128, 37, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(128, generator=g, requires_grad=True)
    ref = F.linear(x, w, b)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    xd, wd, bd = (t.detach().to(device).requires_grad_(True) for t in (x, w, b))
    y = PF.linear(xd, wd, bd)
    y.backward(dy.to(device))
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(bd.grad.cpu(), b.grad, rtol=1e-4, atol=1e-3)
    # attention-shaped products with per-head permuted views
    q = torch.randn(2, 10, 3 * 8, generator=g, requires_grad=True)
    kk = torch.randn(2, 50, 3 * 8, generator=g, requires_grad=True)
    qh = q.view(2, 10, 3, 8).permute(0, 2, 1, 3)
    kh = kk.view(2, 50, 3, 8).permute(0, 2, 1, 3)
    ref = qh @ kh.transpose(-1, -2)
    ds = torch.randn(ref.shape, generator=g)
    ref.backward(ds)
    qd, kd = q.detach().to(device).requires_grad_(True), kk.detach().to(device).requires_grad_(True)
    s = PF.matmul(qd.view(2, 10, 3, 8).permute(0, 2, 1, 3), kd.view(2, 50, 3, 8).permute(0, 2, 1, 3).transpose(-1, -2))
    s.backward(ds.to(device))
    torch.testing.assert_close(s.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(qd.grad.cpu(), q.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(kd.grad.cpu(), kk.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("b,t,inp,h,layers", [(3, 7, 20, 16, 2), (2, 1, 9, 16, 1), (5, 24, 33, 8, 2)])
def test_gru_vs_torch(b, t, inp, h, layers, device):
    """nn.GRU(batch_first) forward/backward (outputs and final hidden state both carry gradients, optional h0)."""
    _, PF = _mods()
    torch.manual_seed(b * t)
    ref = torch.nn.GRU(input_size=inp, hidden_size=h, num_layers=layers, batch_first=True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, t, inp, generator=g, requires_grad=True)
    h0 = torch.randn(layers, b, h, generator=g, requires_grad=True)
    out_r, hn_r = ref(x, h0)
    d_out, d_hn = torch.randn(out_r.shape, generator=g), torch.randn(hn_r.shape, generator=g)
    (out_r * d_out).sum().add((hn_r * d_hn).sum()).backward()
    mod = torch.nn.GRU(input_size=inp, hidden_size=h, num_layers=layers, batch_first=True)
    mod.load_state_dict(ref.state_dict())
    mod.to(device)
    xd, h0d = x.detach().to(device).requires_grad_(True), h0.detach().to(device).requires_grad_(True)
    out, hn = PF.gru(xd, mod, h0d)
    (out * d_out.to(device)).sum().add((hn * d_hn.to(device)).sum()).backward()
    torch.testing.assert_close(out.detach().cpu(), out_r.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(hn.detach().cpu(), hn_r.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(h0d.grad.cpu(), h0.grad, rtol=1e-3, atol=1e-5)
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=1e-3, atol=1e-4, msg=k)
    # without an initial state (zeros), as the encoder is called
    out2, _ = PF.gru(xd.detach(), mod)
    torch.testing.assert_close(out2.cpu(), ref(x.detach())[0], rtol=1e-4, atol=1e-5)


def test_gru_with_the_weights_in_registers_against_the_general_kernels(device, monkeypatch):
    """Hidden size 16 (every GRU of the reference's models) runs kernels that keep W_hh and dW_hh in registers
    (gru_seq_{fwd,bwd}_hs_f32<16>): the forward is the general kernel's bits, the backward too except the gradient that flows
    through the hidden state (summed over the units in index order instead of a shuffle tree)."""
    _, PF = _mods()
    torch.manual_seed(3)
    mod = torch.nn.GRU(input_size=24, hidden_size=16, num_layers=2, batch_first=True).to(device)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(8, 19, 24, generator=g).to(device)
    d_out = torch.randn(8, 19, 16, generator=g).to(device)
    res = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("PV_GRU_GENERAL", "1")
        mod.zero_grad()
        xd = x.clone().requires_grad_(True)
        out, hn = PF.gru(xd, mod)
        (out * d_out).sum().add(hn.sum()).backward()
        res.append((out.detach(), hn.detach(), xd.grad.clone(), [p.grad.clone() for p in mod.parameters()]))
    monkeypatch.delenv("PV_GRU_GENERAL")
    (o1, h1, gx1, gp1), (o0, h0, gx0, gp0) = res
    assert torch.equal(o1, o0) and torch.equal(h1, h0)
    torch.testing.assert_close(gx1, gx0, rtol=1e-5, atol=1e-6)
    for a, b_ in zip(gp1, gp0):
        torch.testing.assert_close(a, b_, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("b,h,nq,nk", [(3, 1, 128, 4096), (2, 8, 128, 128), (2, 2, 40, 70), (1, 1, 130, 33)])
def test_fused_attention_forward(b, h, nq, nk, device):
    """pv_attention_fwd_f32 (online softmax, scores never materialised) vs softmax(scale q k^T) v with torch on the CPU."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(b * nq + nk)
    q = torch.randn(b, nq, h * 64, generator=g)
    kv = torch.randn(b, nk, 2 * h * 64, generator=g)
    scale = 0.125
    qh = q.view(b, nq, h, 64).permute(0, 2, 1, 3)
    kh = kv[..., :h * 64].reshape(b, nk, h, 64).permute(0, 2, 1, 3)
    vh = kv[..., h * 64:].reshape(b, nk, h, 64).permute(0, 2, 1, 3)
    sim = (qh @ kh.transpose(-1, -2)) * scale
    ref = (sim.softmax(dim=-1) @ vh).permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    out, lse = K.attention_fwd(q.to(device), kv.to(device), h, scale)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(sim, dim=-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,h,nq,nk", [(2, 1, 128, 4096), (2, 8, 128, 128), (2, 2, 40, 70), (1, 1, 97, 33)])
def test_fused_attention_backward(b, h, nq, nk, device):
    """pv_attention_bwd_f32 (probabilities recomputed from the saved log-sum-exp) vs torch autograd on the CPU."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(b * nq + nk + 1)
    q = torch.randn(b, nq, h * 64, generator=g, requires_grad=True)
    kv = torch.randn(b, nk, 2 * h * 64, generator=g, requires_grad=True)
    scale = 0.125
    qh = q.view(b, nq, h, 64).permute(0, 2, 1, 3)
    kh = kv[..., :h * 64].reshape(b, nk, h, 64).permute(0, 2, 1, 3)
    vh = kv[..., h * 64:].reshape(b, nk, h, 64).permute(0, 2, 1, 3)
    ref = (((qh @ kh.transpose(-1, -2)) * scale).softmax(dim=-1) @ vh).permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    qd, kvd = q.detach().to(device), kv.detach().to(device)
    out, lse = K.attention_fwd(qd, kvd, h, scale)
    dq, dkv = K.attention_bwd(qd, kvd, out, dout.to(device), lse, h, scale)
    torch.testing.assert_close(dq.cpu(), q.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(dkv.cpu(), kv.grad, rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n,bias,relu", [(2048, 64, 64, True, False), (4099, 38, 128, False, False), (2500, 37, 70, True, True),
                                             (19456, 64, 1024, True, False), (3000, 16, 5, False, False), (2048, 6, 64, True, False)])
def test_gemm_rows_form_matches_f64(m, k, n, bias, relu):
    """pv_gemm_f32's persistent "rows" form (tall row-major A, K <= 64: A operands straight into registers, weights split
    once per workgroup) against an f64 product: every vector width (K % 4, K % 2, odd K), ragged M and N, bias and ReLU,
    and B given as a transposed view (nn.Linear's weight.t()) as well as row-major."""
    K, _ = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(m + k + n)
    a = torch.randn(m, k, generator=g, device=dev)
    w = torch.randn(n, k, generator=g, device=dev)
    bv = torch.randn(n, generator=g, device=dev) if bias else None
    for b_op in (w.t(), w.t().contiguous()):
        c = K.gemm(a, b_op, bias=bv, relu=relu)
        ref = a.double() @ w.double().t()
        if bias:
            ref = ref + bv.double()
        if relu:
            ref = ref.clamp_min(0)
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (m, k, n, err)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,n", [(300, 256, 96), (4099, 38, 128), (2048, 64, 64), (257, 1000, 130)])
def test_gemm_with_bf16_operands_is_the_product_of_the_rounded_operands(m, k, n):
    """bf16_operands=True (PV_GEMM_BF16_OPERANDS): both operands rounded once to bf16 (nearest even), f32 accumulation -- the
    f64 product of the ROUNDED operands to f32 accuracy, and visibly not the f32-accurate product; tiled form, rows form
    (tall A, K <= 64), the bf16-storing rows form and the split-K weight-gradient form."""
    K, _ = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(m + n)
    a = torch.randn(m, k, generator=g, device=dev)
    w = torch.randn(n, k, generator=g, device=dev)
    bias = torch.randn(n, generator=g, device=dev)
    r = lambda t: t.to(torch.bfloat16).double()
    ref16 = r(a) @ r(w).t() + bias.double()
    ref32 = a.double() @ w.double().t() + bias.double()
    c = K.gemm(a, w.t(), bias=bias, bf16_operands=True)
    assert float((c.double() - ref16).abs().max() / ref16.abs().max()) < 2e-6
    assert float((c.double() - ref32).abs().max() / ref32.abs().max()) > 1e-4
    assert float((K.gemm(a, w.t(), bias=bias).double() - ref32).abs().max() / ref32.abs().max()) < 2e-6
    if K.gemm_rows_bf16out_supported(a, w.t()):
        assert torch.equal(K.gemm_rows_bf16out(a, w.t(), bias=bias, bf16_operands=True), c.to(torch.bfloat16))
    dw = K.gemm_splitk(c.t(), a, bf16_operands=True)      # [n, m] @ [m, k]
    refw = r(c).t() @ r(a)
    assert float((dw.double() - refw).abs().max() / refw.abs().max()) < 1e-5
