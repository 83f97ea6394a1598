"""GPU parity: Conv3D / fully-connected / loss / Adam HIP kernels (through the C ABI) vs the torch-CPU oracle.

Tolerances:
  fp32 path : activations rtol 1e-4 / atol 1e-5, wgrad 1e-3 rel (f32 accumulation-order differences only)
  bf16 path : compared against the oracle evaluated on bf16-ROUNDED operands (what the MFMA consumes);
              the remaining difference is f32 accumulation order + one bf16 rounding of the output:
              rtol 1e-2 (2 bf16 ulps) / atol 2e-3.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv3d_oracle as co

pytestmark = pytest.mark.gpu


def _mods():
    from predict_pv_yield_amd import functional, hip_ops
    return hip_ops, functional


def _conv_case(seed, b, ci, co_, t, h, w):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, ci, t, h, w, generator=g)
    wt = torch.randn(co_, ci, 3, 3, 3, generator=g) * (1.0 / np.sqrt(ci * 27))
    bias = torch.randn(co_, generator=g) * 0.1
    return x, wt, bias


CASES = [
    # b, ci, co, t, h, w, padding
    (2, 11, 32, 7, 16, 16, (0, 0, 0)),
    (1, 32, 32, 5, 14, 20, (0, 0, 0)),
    (2, 11, 32, 6, 12, 12, (1, 0, 0)),   # model_sat_nwp padding
    (1, 10, 16, 4, 9, 9, (1, 1, 1)),     # Conv3dMaxPool padding, c_out < 32
    (1, 32, 32, 4, 70, 66, (0, 0, 0)),   # more than one column block / row block
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_f32_forward_backward(device, case):
    K, Fn = _mods()
    b, ci, co_, t, h, w, pad = case
    x, wt, bias = _conv_case(1, b, ci, co_, t, h, w)
    x.requires_grad_(True); wt.requires_grad_(True); bias.requires_grad_(True)
    y_ref = F.relu(F.conv3d(x, wt, bias, padding=pad))
    gy = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(2))
    y_ref.backward(gy)

    xd = x.detach().to(device).requires_grad_(True)
    wd = wt.detach().to(device).requires_grad_(True)
    bd = bias.detach().to(device).requires_grad_(True)
    y = Fn.conv3d_general_f32(xd, wd, bd, stride=(1, 1, 1), padding=pad, relu=True)
    y.backward(gy.to(device))
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wd.grad.cpu(), wt.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(bd.grad.cpu(), bias.grad, rtol=1e-3, atol=1e-4)


def test_pack_unpack_roundtrip(device):
    K, _ = _mods()
    for c in (11, 32, 5):
        x = torch.randn(2, c, 3, 5, 7)
        xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
        assert xp.shape[-1] == (16 if c <= 16 else 32)
        ref = x.to(torch.bfloat16).permute(0, 2, 3, 4, 1)
        assert torch.equal(xp[..., :c].cpu(), ref)
        assert torch.count_nonzero(xp[..., c:]).item() == 0
        back = K.unpack_ndhwc_bf16_to_ncdhw_f32(xp, c)
        assert torch.equal(back.cpu(), co.bf16_round(x))


@pytest.mark.parametrize("case", CASES)
def test_conv3d_bf16_forward(device, case):
    K, _ = _mods()
    b, ci, co_, t, h, w, pad = case
    x, wt, bias = _conv_case(3, b, ci, co_, t, h, w)
    y_ref = F.relu(F.conv3d(co.bf16_round(x), co.bf16_round(wt), bias, padding=pad))
    xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
    wp = K.conv3d_pack_weight_bf16(wt.to(device))
    y = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), ci, co_, pad, relu=True, y_ncdhw=False)
    got = y.float().cpu().permute(0, 4, 1, 2, 3)[:, :co_]
    torch.testing.assert_close(got, y_ref, rtol=1e-2, atol=2e-3)
    if co_ < 32:
        assert torch.count_nonzero(y[..., co_:]).item() == 0
    # NCDHW epilogue (the flatten order fc1 consumes): same numbers up to the f32 accumulation order of the kernel
    # variant that serves it (32 -> 32 NDHWC layers run the two-waves-per-SIMD 16x16x32 kernel)
    y2 = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), ci, co_, pad, relu=True, y_ncdhw=True)
    torch.testing.assert_close(y2.float().cpu(), got.contiguous(), rtol=1e-2, atol=2e-3)
    torch.testing.assert_close(y2.float().cpu(), y_ref, rtol=1e-2, atol=2e-3)


@pytest.mark.parametrize("case", CASES)
def test_conv3d_bf16_backward(device, case):
    """wgrad (transposed-LDS-read MFMA kernel) and dgrad (forward kernel on mirrored weights)."""
    K, _ = _mods()
    b, ci, co_, t, h, w, pad = case
    if co_ != 32:
        pytest.skip("bf16 backward is built for 32 output channels")
    x, wt, bias = _conv_case(4, b, ci, co_, t, h, w)
    xr = co.bf16_round(x).requires_grad_(True)
    wr = co.bf16_round(wt).requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    y_ref = F.relu(F.conv3d(xr, wr, br, padding=pad))
    gy = co.bf16_round(torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(5)))
    y_ref.backward(gy)

    xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
    wp = K.conv3d_pack_weight_bf16(wt.to(device))
    y = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), ci, co_, pad, relu=True)
    gyp = K.pack_ncdhw_f32_to_ndhwc_bf16(gy.to(device))
    # the gate must come from the same activations the oracle gated on: use the oracle's y > 0 pattern
    gate = K.pack_ncdhw_f32_to_ndhwc_bf16(y_ref.detach().to(device))
    dw, db = K.conv3d_bwd_weight_bf16(xp, gyp, gate, ci, co_, pad)
    scale = wr.grad.abs().max().item()
    assert (dw.cpu() - wr.grad).abs().max().item() <= 2e-3 * scale + 1e-4
    assert (db.cpu() - br.grad).abs().max().item() <= 2e-3 * br.grad.abs().max().item() + 1e-4
    # dgrad
    wpt = K.conv3d_pack_weight_bf16(wt.to(device), transpose_flip=True)
    pad_b = tuple(2 - p for p in pad)
    dx = K.conv3d_fwd_bf16(gyp, gate, wpt, None, co_, ci, pad_b, relu=False)
    got = dx.float().cpu().permute(0, 4, 1, 2, 3)[:, :ci]
    torch.testing.assert_close(got, xr.grad, rtol=1e-2, atol=2e-3 * max(1.0, xr.grad.abs().max().item()))
    if ci == 32:
        # out_gate: the same dgrad, zeroed where the gate tensor (here x itself) is not positive -- bit-identical
        # to masking the ungated result afterwards
        dxg = K.conv3d_fwd_bf16(gyp, gate, wpt, None, co_, ci, pad_b, relu=False, out_gate=xp)
        assert torch.equal(dxg, torch.where(xp > 0, dx, torch.zeros_like(dx)))
    del y


V3_SHAPES = [
    # b, t, h, w, padding: the input-stationary 32 -> 32 kernel (8 x 32 tiles, >= 2 output slices per time chunk)
    (1, 4, 10, 34, (0, 0, 0)),    # to = 2: head + tail steps only, one full-width tile
    (1, 5, 11, 35, (0, 0, 0)),    # to = 3, 33 output columns: second column block holds a single column
    (2, 9, 9, 20, (0, 0, 0)),     # to = 7, split into time chunks (few tiles); ragged rows / columns
    (1, 7, 19, 67, (1, 0, 0)),    # to = 7 with time padding, three column blocks
    (1, 3, 12, 12, (2, 2, 2)),    # dgrad-style full padding, to = 5 -> chunks of 3 + 2
    (3, 6, 8, 8, (2, 2, 2)),      # to = 8 in chunks
    (1, 3, 6, 6, (0, 0, 0)),      # to = 1: not covered by v3, served by the v2 kernel
    (2, 5, 12, 18, (0, 0, 0)),    # w_out = 16: the NCDHW epilogue of v3 (16-byte pieces of an output line)
    (1, 6, 9, 42, (0, 0, 0)),     # w_out = 40: NCDHW epilogue with a ragged second column block, ragged rows
]


@pytest.mark.parametrize("shape", V3_SHAPES)
@pytest.mark.parametrize("relu", [True, False])
def test_conv3d_bf16_input_stationary_kernel(device, shape, relu):
    """conv3d_fwd_bf16_v3_kernel (conv3d_bf16_v3.hip): every step kind of the time march (head, interior, tail for all
    three accumulator phases), time chunking, ragged tiles; and its OUT_GATE epilogue = masking the plain result."""
    K, _ = _mods()
    b, t, h, w, pad = shape
    x, wt, bias = _conv_case(11, b, 32, 32, t, h, w)
    y_ref = F.conv3d(co.bf16_round(x), co.bf16_round(wt), bias, padding=pad)
    if relu:
        y_ref = F.relu(y_ref)
    xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
    wp = K.conv3d_pack_weight_bf16(wt.to(device))
    y = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=False)
    torch.testing.assert_close(y.float().cpu().permute(0, 4, 1, 2, 3), y_ref, rtol=1e-2, atol=2e-3)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(12)).to(device).to(torch.bfloat16)
    g[0, 0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), 1e-30, -1e-30]).to(g)
    yg = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=False, out_gate=g)
    assert torch.equal(yg, torch.where(g > 0, y, torch.zeros_like(y)))
    # NCDHW epilogue (what fc1 consumes): same kernel and accumulation order when w_out % 8 == 0 -> bit-identical
    yn = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=True)
    if y.shape[3] % 8 == 0 and y.shape[1] >= 2:
        assert torch.equal(yn, y.permute(0, 4, 1, 2, 3))
    else:
        torch.testing.assert_close(yn.float(), y.permute(0, 4, 1, 2, 3).float(), rtol=1e-2, atol=2e-3)


@pytest.mark.parametrize("shape", [
    # b, ci, t, h, w, padding
    (2, 11, 7, 16, 16, (0, 0, 0)),
    (1, 11, 6, 12, 70, (1, 0, 0)),     # two column blocks, time padding (model_sat_nwp)
    (3, 12, 5, 21, 9, (1, 1, 0)),      # ragged rows and a ragged last voxel quad, t / h padding
    (1, 1, 3, 8, 8, (0, 0, 0)),        # a single input channel, one output slice
    (2, 16, 9, 10, 67, (2, 2, 0)),     # all 16 channels, widest t / h padding, time chunks, odd width
])
def test_conv3d_first_layer_from_f32_input(device, shape):
    """pv_conv3d_fwd_bf16_f32in = pack + conv in one pass: same bits for y, and the NDHWC bf16 image it leaves behind
    for the weight gradient equals the pack kernel's (every input voxel written exactly once)."""
    K, Fn = _mods()
    b, ci, t, h, w, pad = shape
    x, wt, bias = _conv_case(21, b, ci, 32, t, h, w)
    xd = x.to(device)
    wp = K.conv3d_pack_weight_bf16(wt.to(device))
    xp_ref = K.pack_ncdhw_f32_to_ndhwc_bf16(xd)
    y_ref = K.conv3d_fwd_bf16(xp_ref, None, wp, bias.to(device), ci, 32, pad, relu=True, y_ncdhw=False)
    xp_got = torch.full_like(xp_ref, float("nan"))
    y, xp = K.conv3d_fwd_bf16_f32in(xd, wp, bias.to(device), 32, pad, relu=True)
    assert torch.equal(y, y_ref)
    assert torch.equal(xp, xp_ref)
    y2, none = K.conv3d_fwd_bf16_f32in(xd, wp, bias.to(device), 32, pad, relu=True, want_packed=False)
    assert none is None and torch.equal(y2, y_ref)
    del xp_got
    # autograd wrapper: same weight / bias gradients as the two-kernel path
    w1 = wt.to(device).requires_grad_(True); b1 = bias.to(device).requires_grad_(True)
    w2 = wt.to(device).requires_grad_(True); b2 = bias.to(device).requires_grad_(True)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(device).to(torch.bfloat16)
    Fn.conv3d_first_layer_bf16(xd, w1, b1, pad, relu=True, dy_pregated=False).backward(g)
    Fn.conv3d_relu_bf16(Fn.PackInputBF16.apply(xd), w2, b2, ci, pad, relu=True, y_ncdhw=False).backward(g)
    assert torch.equal(w1.grad, w2.grad) and torch.equal(b1.grad, b2.grad)


def test_conv3d_bf16_random_shapes(device):
    """Seeded sweep over ragged shapes / paddings / batch sizes of the 32 -> 32 bf16 forward (all three epilogues): the
    time-chunking, head / tail phases and tile clipping of the marching kernels depend on every one of these numbers."""
    K, _ = _mods()
    rng = np.random.default_rng(2024)
    for it in range(24):
        b = int(rng.integers(1, 4))
        pad = tuple(int(v) for v in rng.integers(0, 3, size=3))
        t = int(rng.integers(max(1, 3 - 2 * pad[0]), 9))
        h = int(rng.integers(max(1, 3 - 2 * pad[1]), 21))
        w = int(rng.integers(max(1, 3 - 2 * pad[2]), 71))
        relu = bool(rng.integers(0, 2))
        x, wt, bias = _conv_case(100 + it, b, 32, 32, t, h, w)
        y_ref = F.conv3d(co.bf16_round(x), co.bf16_round(wt), bias, padding=pad)
        if relu:
            y_ref = F.relu(y_ref)
        xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
        wp = K.conv3d_pack_weight_bf16(wt.to(device))
        y = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=False)
        msg = f"shape #{it}: b={b} t={t} h={h} w={w} pad={pad} relu={relu}"
        torch.testing.assert_close(y.float().cpu().permute(0, 4, 1, 2, 3), y_ref, rtol=1e-2, atol=2e-3, msg=msg)
        yn = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=True)
        torch.testing.assert_close(yn.float().cpu(), y_ref, rtol=1e-2, atol=2e-3, msg=msg + " (NCDHW)")
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(it)).to(device).to(torch.bfloat16)
        yg = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=False, out_gate=g)
        assert torch.equal(yg, torch.where(g > 0, y, torch.zeros_like(y))), msg + " (out_gate)"
        # 1-bit relu masks: the mask written next to y holds exactly (y > 0) per channel bit, and a dgrad-style launch
        # gated through the mask of g equals the one that reads the bf16 tensor g
        y2, mask = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=False, want_relu_mask=True)
        assert torch.equal(y2, y), msg + " (mask variant changed y)"
        assert torch.equal(_mask_bits(mask, y.shape), y > 0), msg + " (relu mask)"
        gmask = _mask_of(g, K)
        ygm = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), 32, 32, pad, relu=relu, y_ncdhw=False, out_gate=g,
                                out_gate_mask=gmask)
        assert torch.equal(ygm, yg), msg + " (out_gate_mask)"


def test_conv3d_bf16_weight_gradient_and_first_layer_random_shapes(device):
    """Seeded sweep over ragged shapes / paddings / batch sizes / channel counts of (a) the weight gradient -- the
    loader-wave kernel, with dY arriving gated or a gate handed over (a gated copy of dY is staged in the workspace), 16 and 32
    padded input channels, one to three column tiles, time chunks -- and (b) the first layer from the f32 NCDHW input
    (loader-wave kernel where w % 4 == 0 and no width padding, the one-role kernel otherwise): against torch on
    bf16-rounded operands, and the two first-layer paths against pack + conv bit for bit."""
    K, _ = _mods()
    rng = np.random.default_rng(77)
    for it in range(20):
        b = int(rng.integers(1, 4))
        ci = int(rng.choice([3, 11, 16, 32, 32]))
        pad = tuple(int(v) for v in rng.integers(0, 2, size=3))
        t = int(rng.integers(3, 8))
        h = int(rng.integers(4, 24))
        w = int(rng.integers(4, 72))
        msg = f"case #{it}: b={b} ci={ci} t={t} h={h} w={w} pad={pad}"
        x, wt, bias = _conv_case(300 + it, b, ci, 32, t, h, w)
        xr = co.bf16_round(x)
        wr = co.bf16_round(wt).requires_grad_(True)
        br = bias.clone().requires_grad_(True)
        y_ref = F.relu(F.conv3d(xr, wr, br, padding=pad))
        gy = co.bf16_round(torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(it)))
        y_ref.backward(gy)
        xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
        gate = K.pack_ncdhw_f32_to_ndhwc_bf16(y_ref.detach().to(device))
        gy_gated = torch.where(y_ref.detach() > 0, gy, torch.zeros_like(gy))
        scale_w, scale_b = wr.grad.abs().max().item(), br.grad.abs().max().item()
        for gated_outside in (True, False):
            if gated_outside:       # dY already carries the ReLU derivative: the loader-wave kernel
                dw, db = K.conv3d_bwd_weight_bf16(xp, K.pack_ncdhw_f32_to_ndhwc_bf16(gy_gated.to(device)), None, ci, 32, pad)
            else:                   # a gate is handed over: the entry point stages a gated copy of dY, same kernel
                dw, db = K.conv3d_bwd_weight_bf16(xp, K.pack_ncdhw_f32_to_ndhwc_bf16(gy.to(device)), gate, ci, 32, pad)
            assert (dw.cpu() - wr.grad).abs().max().item() <= 2e-3 * scale_w + 1e-4, msg + f" dw (gated_outside={gated_outside})"
            assert (db.cpu() - br.grad).abs().max().item() <= 2e-3 * scale_b + 1e-4, msg + f" db (gated_outside={gated_outside})"
        if ci <= 16 and pad[2] == 0:      # (the entry point takes no width padding)
            wp = K.conv3d_pack_weight_bf16(wt.to(device))
            y1, xp1 = K.conv3d_fwd_bf16_f32in(x.to(device), wp, bias.to(device), 32, pad, relu=True)
            y2 = K.conv3d_fwd_bf16(xp, None, wp, bias.to(device), ci, 32, pad, relu=True)
            assert torch.equal(y1, y2) and torch.equal(xp1, xp), msg + " first layer vs pack + conv"
            torch.testing.assert_close(y1.float().cpu().permute(0, 4, 1, 2, 3), y_ref.detach(), rtol=1e-2, atol=2e-3, msg=msg)


@pytest.mark.parametrize("ci", [1, 3, 11, 12])
def test_weight_gradient_packed_three_pieces_per_tap_equals_the_paired_form(device, monkeypatch, ci):
    """At most 12 input channels (the first layer's 11): the loader-wave weight-gradient kernel packs the N axis as (tap,
    4-channel piece) with three pieces per tap -- 11 column tiles instead of 14.  Every dW element is the same sequence of
    matrix-instruction terms as in the paired form (PV_WGRAD_NO_PACK12=1): bit for bit, incl. the bias gradient, ragged sizes,
    padding, several column tiles and time chunks."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(40 + ci)
    for (b, t, h, w, pad) in [(2, 5, 11, 70, (0, 0, 0)), (3, 4, 9, 33, (1, 1, 1)), (32, 6, 16, 16, (0, 1, 0)), (1, 3, 5, 5, (0, 0, 0))]:
        x = torch.randn(b, ci, t, h, w, generator=g)
        to, ho, wo = t + 2 * pad[0] - 2, h + 2 * pad[1] - 2, w + 2 * pad[2] - 2
        gy = torch.randn(b, 32, to, ho, wo, generator=g)
        xp = K.pack_ncdhw_f32_to_ndhwc_bf16(x.to(device))
        gyp = K.pack_ncdhw_f32_to_ndhwc_bf16(gy.to(device))
        monkeypatch.delenv("PV_WGRAD_NO_PACK12", raising=False)
        dw_a, db_a = K.conv3d_bwd_weight_bf16(xp, gyp, None, ci, 32, pad)
        monkeypatch.setenv("PV_WGRAD_NO_PACK12", "1")
        dw_b, db_b = K.conv3d_bwd_weight_bf16(xp, gyp, None, ci, 32, pad)
        monkeypatch.delenv("PV_WGRAD_NO_PACK12", raising=False)
        assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b), (ci, b, t, h, w, pad)
        ref = torch.nn.grad.conv3d_weight(co.bf16_round(x), (32, ci, 3, 3, 3), co.bf16_round(gy), padding=pad)
        assert (dw_a.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize("magnitude", [3.0, 2e-7, 5e4])
def test_two_term_half_float_split_and_the_weight_gradient_built_on_it(device, magnitude):
    """pv_pack_split2_...: x s = h + l to 2^-22 of the largest element (s the power of two the call reports), whatever the
    tensor's magnitude (gradients of 1e-7 would vanish in a half float without the scale); and the weight gradient formed from
    three f16 matrix-core launches on the planes, un-scaled, equals torch's f64 gradient to a few 1e-6."""
    K, _ = _mods()
    shape = (3, 11, 6, 10, 12)
    b, c, t, h, w = shape
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(shape, generator=g) * magnitude).to(device)
    hh, ll, st = K.pack_split2_ncdhw_f32_to_ndhwc_f16(x)
    s, inv = float(st[1]), float(st[2])
    assert s * inv == 1.0 and np.log2(s) == round(np.log2(s))
    assert 2.0 ** 13 <= float(x.abs().max()) * s < 2.0 ** 14
    rec = ((hh.double() + ll.double()) * inv).permute(0, 4, 1, 2, 3)[:, :c]
    assert float((rec - x.double()).abs().max()) <= 2.0 ** -22 * float(x.abs().max())
    assert bool((hh[..., c:] == 0).all()) and bool((ll[..., c:] == 0).all())
    wt = torch.randn(32, c, 3, 3, 3, generator=g) * 0.1
    xr = x.cpu().double()
    wr = wt.double().requires_grad_(True)
    br = torch.zeros(32, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(xr, wr, br)
    gy = torch.randn(y.shape, generator=g) * (1e-3 / magnitude)
    y.backward(gy.double())
    dw, db = K.conv3d_bwd_weight_f32_on_f16x2(x, gy.to(device).contiguous())
    assert (dw.cpu().double() - wr.grad).abs().max().item() <= 2e-5 * wr.grad.abs().max().item()
    assert (db.cpu().double() - br.grad).abs().max().item() <= 2e-5 * br.grad.abs().max().item()
    # the gate pass that produces a gated gradient can leave its largest magnitude for the split: same planes, same result
    yy = torch.randn(gy.shape, generator=g).to(device)
    gated, st = K.relu_gate_f32(gy.to(device).contiguous(), yy, want_max=True)
    assert torch.equal(gated, K.relu_gate_f32(gy.to(device).contiguous(), yy))
    assert st[0:1].view(torch.int32).item() == gated.abs().max().view(torch.int32).item()
    a = K.pack_split2_ncdhw_f32_to_ndhwc_f16(gated)
    b_ = K.pack_split2_ncdhw_f32_to_ndhwc_f16(gated, maxabs_state=st)
    assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]) and torch.equal(a[2][1:], b_[2][1:])


@pytest.mark.parametrize("shape,pad,magnitude", [
    ((8, 32, 8, 44, 44), (0, 0, 0), 1.0),        # the model's layers: valid convolution
    ((6, 32, 6, 40, 52), (1, 1, 1), 3e-6),       # 'same' padding, gradient-sized activations (the scale matters)
    ((8, 32, 5, 58, 58), (0, 0, 0), 40.0),       # 56-pixel planes: ragged tiles (8-row blocks, 32-column blocks)
    ((128, 32, 4, 20, 20), (0, 0, 0), 1.0),      # many samples of few voxels (648 = 20.25 tiles of the sum pass each)
    ((8, 11, 8, 44, 44), (0, 0, 0), 2.0),        # the first layer: 11 channels in an operand image padded to 32, no dx
])
def test_f32_conv_forward_and_dgrad_as_three_half_float_products(device, shape, pad, magnitude):
    """csrc/conv3d_f16x2.hip: nn.Conv3d(32, 32, 3) in float32 from two-term half-float splits of x and w -- three launches of
    the input-stationary kernel in its half-float form (f32 accumulators out) + one ordered-sum pass -- against float64 on the
    CPU: forward (+ bias, ReLU), data gradient (+ the producer's ReLU gate), weight and bias gradient from the same operand
    images.  Bound: 4e-6 of the largest element (22-bit operands, f32 accumulation over 864 terms; the f32 kernels sit at
    1-2e-6 on the same data, the bf16 path at 1e-2).  The sum pass's own split of y reconstructs it to 2^-21 of its largest element."""
    K, _ = _mods()
    from predict_pv_yield_amd import functional as Fn
    b, c, t, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + pad[0])
    x = (torch.randn(shape, generator=g).abs() * magnitude)      # a ReLU output: its zeros gate dx
    x[torch.rand(shape, generator=g) < 0.3] = 0.0
    first = c < 32
    if first:
        x = x - 0.5 * magnitude      # (the model's input is normalised data, not a ReLU output)
    wt = torch.randn(32, c, 3, 3, 3, generator=g) * 0.05
    bias = torch.randn(32, generator=g) * 0.1 * magnitude
    xr = x.double().requires_grad_(True)
    wr, br = wt.double().requires_grad_(True), bias.double().requires_grad_(True)
    y_ref = F.relu(F.conv3d(xr, wr, br, padding=pad))
    gy = torch.randn(y_ref.shape, generator=g) * 1e-4
    # (no gradient where the pre-activation is within rounding of zero: there the ReLU's derivative is a coin toss between any two
    # arithmetics, and one flipped voxel moves dx by |gy w| = 1e-5 of its largest element)
    gy = gy * (y_ref.detach() > 1e-4 * float(y_ref.max())).float()
    (y_ref * gy.double()).sum().backward()
    dx_ref = xr.grad * (x > 0)

    xd = x.to(device).requires_grad_(not first)
    wd, bd = wt.to(device).requires_grad_(True), bias.to(device).requires_grad_(True)
    assert Fn._conv_on_f16x2(xd, wd, (1, 1, 1), pad)
    y = Fn.conv3d_general_f32(xd, wd, bd, stride=(1, 1, 1), padding=pad, relu=True, x_is_relu_output=not first, dy_pregated=False)
    assert y.grad_fn is not None and type(y.grad_fn).__name__.startswith("Conv3dF32OnF16x2")
    # the sum pass leaves y's own two-term split for the next layer (scale from a bound of |y| known beforehand), and y's maximum
    yh, yl, ys, _ = y._pv_planes
    assert ys[0:1].view(torch.int32).item() == y.detach().abs().max().view(torch.int32).item()
    s_y = float(ys[1])
    assert s_y * float(ys[2]) == 1.0 and np.log2(s_y) == round(np.log2(s_y)) and float(y.detach().abs().max()) * s_y < 2.0 ** 14
    rec = ((yh.double() + yl.double()) * float(ys[2])).permute(0, 4, 1, 2, 3)
    assert float((rec - y.detach().double()).abs().max()) <= 2.0 ** -21 * float(y.detach().abs().max())
    assert float(y.detach().abs().max()) * s_y >= 2.0 ** 3, "the bound is more than 2^11 above the largest element: the split loses bits"
    y.backward(gy.to(device))

    def close(a, ref, what):
        err = float((a.detach().cpu().double() - ref).abs().max())
        assert err <= 4e-6 * float(ref.abs().max()), (what, err, float(ref.abs().max()))
    close(y, y_ref.detach(), "y")
    if not first:
        close(xd.grad, dx_ref, "dx")
    assert wd.grad.shape == wt.shape
    close(wd.grad, wr.grad, "dw")
    close(bd.grad, br.grad, "db")
    # the two small products kept as f32 instead of half floats (the switch's other arm): same bound, not the same bits
    K.F16X2_SMALL_PRODUCTS_F16 = False
    try:
        x3 = x.to(device).requires_grad_(not first)
        w3, b3 = wt.to(device).requires_grad_(True), bias.to(device).requires_grad_(True)
        y3 = Fn.conv3d_general_f32(x3, w3, b3, stride=(1, 1, 1), padding=pad, relu=True, x_is_relu_output=not first, dy_pregated=False)
        y3.backward(gy.to(device))
    finally:
        K.F16X2_SMALL_PRODUCTS_F16 = True
    close(y3, y_ref.detach(), "y (f32 small products)")
    if not first:
        close(x3.grad, dx_ref, "dx (f32 small products)")
    assert not torch.equal(y3.detach(), y.detach()), "the two arms are different kernels: identical bits mean the switch is dead"
    # the f32 matrix-instruction kernels on the same data: both forms within the bound of each other
    Fn.F32_CONV_ON_F16X2 = False
    try:
        x2 = x.to(device).requires_grad_(not first)
        w2, b2 = wt.to(device).requires_grad_(True), bias.to(device).requires_grad_(True)
        y2 = Fn.conv3d_general_f32(x2, w2, b2, stride=(1, 1, 1), padding=pad, relu=True, x_is_relu_output=not first, dy_pregated=False)
        assert type(y2.grad_fn).__name__.startswith("Conv3dGeneralF32")
        y2.backward(gy.to(device))
    finally:
        Fn.F32_CONV_ON_F16X2 = True
    close(y2, y_ref.detach(), "y (f32 kernels)")
    if not first:
        close(x2.grad, dx_ref, "dx (f32 kernels)")
    assert float((y - y2).abs().max()) <= 4e-6 * float(y_ref.abs().max())


def test_f32_conv_on_half_floats_edge_inputs(device):
    """Edges of the half-float form's scaling: an all-zero input (maximum 0: scale 1, output = ReLU(bias) exactly, zero
    gradients for w), an input whose largest element is 10^20 times the typical one (the scale follows the outlier; the
    result still sits within 4e-6 of float64 at ITS largest element, and the ordinary voxels keep an absolute error of 2^-32
    of it), and magnitudes at the ends of the float32 range (1e-30, 1e30)."""
    K, _ = _mods()
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator().manual_seed(99)
    shape = (8, 32, 8, 44, 44)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    b = torch.randn(32, generator=g) * 0.1
    wd, bd = w.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    xz = torch.zeros(shape, device=device, requires_grad=True)
    y = Fn.conv3d_general_f32(xz, wd, bd, stride=(1, 1, 1), padding=(0, 0, 0), relu=True, x_is_relu_output=True)
    assert type(y.grad_fn).__name__.startswith("Conv3dF32OnF16x2")
    assert torch.equal(y, torch.relu(bd.detach()).view(1, 32, 1, 1, 1).expand_as(y))
    y.sum().backward()
    assert float(wd.grad.abs().max()) == 0.0 and float(xz.grad.abs().max()) == 0.0 and bool(torch.isfinite(bd.grad).all())
    for magnitude, outlier in ((1.0, 1e20), (1e-30, None), (1e30, None)):
        x = torch.randn(shape, generator=g).abs() * magnitude
        if outlier:
            x[3, 7, 4, 20, 20] = outlier
        ref = F.relu(F.conv3d(x.double(), w.double() * (1.0 if magnitude == 1.0 else 1.0), b.double() * magnitude))
        with torch.no_grad():
            y = Fn.conv3d_general_f32(x.to(device), wd.detach(), (b * magnitude).to(device), stride=(1, 1, 1), padding=(0, 0, 0), relu=True)
        assert bool(torch.isfinite(y).all())
        assert float((y.cpu().double() - ref).abs().max()) <= 4e-6 * float(ref.abs().max()), (magnitude, outlier)


def test_f32_conv_on_half_floats_is_exactly_homogeneous_at_the_headline_layer_size(device):
    """A size-independent property at the full size of the model's second layer (B = 32, 32 x 16 x 62 x 62 -> 14 x 60 x 60): with
    no bias, conv(2^a x, 2^b w) = 2^(a+b) conv(x, w) BIT FOR BIT, and so are dx, dw, db under the matching scaling of the output
    gradient -- every scale of the half-float form (operand splits from the tensors' maxima, the output's split from a bound,
    the 2^-12 of the small products) is a power of two derived from the data, so scaling the data by powers of two may not
    change a single mantissa.  (An arithmetic that clipped, flushed or mis-scaled any term would break this at some exponent.)"""
    K, _ = _mods()
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator(device=device).manual_seed(21)
    x = torch.randn(32, 32, 16, 62, 62, generator=g, device=device).abs_()
    x[torch.rand(x.shape, generator=g, device=device) < 0.3] = 0.0
    w = torch.randn(32, 32, 3, 3, 3, generator=g, device=device) * 0.05
    gy = torch.randn(32, 32, 14, 60, 60, generator=g, device=device) * 1e-4

    def run(a, b, c):      # x 2^a, w 2^b, gy 2^c
        xd = (x * 2.0 ** a).requires_grad_(True)
        wd = (w * 2.0 ** b).requires_grad_(True)
        assert Fn._conv_on_f16x2(xd, wd, (1, 1, 1), (0, 0, 0))
        y = Fn.conv3d_general_f32(xd, wd, None, stride=(1, 1, 1), padding=(0, 0, 0), relu=True, x_is_relu_output=True, dy_pregated=False)
        y.backward(gy * 2.0 ** c)
        return y.detach(), xd.grad, wd.grad

    y0, dx0, dw0 = run(0, 0, 0)
    assert float(y0.abs().max()) > 0 and float(dx0.abs().max()) > 0
    for a, b, c in ((7, -3, 0), (-20, 5, 9), (0, -12, -6)):
        y1, dx1, dw1 = run(a, b, c)
        assert torch.equal(y1, y0 * 2.0 ** (a + b)), (a, b, c)
        assert torch.equal(dx1, dx0 * 2.0 ** (b + c)), (a, b, c)
        assert torch.equal(dw1, dw0 * 2.0 ** (a + c)), (a, b, c)


def test_f32_conv_layers_chained_through_their_operand_images(device):
    """Three Conv3d(32, 32, 3) + ReLU in float32, the first two called with chain_out=True: what travels between them forward
    (and backward) is the pair of half-float operand images the producing sum pass wrote -- no float32 activation, no split
    pass -- and every result has the bits of the unchained call's (whose layers find the same images attached to the float32
    tensors); both sit at the single layer's bound against float64 wherever no inner ReLU flips."""
    K, _ = _mods()
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    shape = (8, 32, 10, 52, 52)      # 160 000, 110 592 and 67 712 output voxels: every layer is worth the three launches
    x = torch.randn(shape, generator=g).abs()
    x[torch.rand(shape, generator=g) < 0.3] = 0.0
    ws = [torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05 for _ in range(3)]
    bs = [torch.randn(32, generator=g) * 0.1 for _ in range(3)]
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    h = xr
    pre = []
    for w, b in zip(wr, br):
        h = F.relu(F.conv3d(h, w, b))
        pre.append(h)
    gy = torch.randn(h.shape, generator=g) * 1e-3
    gy = gy * (h.detach() > 1e-4 * float(h.max())).float()
    (h * gy.double()).sum().backward()

    def run(chain):
        xd = x.to(device).requires_grad_(True)
        wd = [w.to(device).requires_grad_(True) for w in ws]
        bd = [b.to(device).requires_grad_(True) for b in bs]
        out = xd
        kinds = []
        for i in range(3):
            out = Fn.conv3d_general_f32(out, wd[i], bd[i], stride=(1, 1, 1), padding=(0, 0, 0), relu=True, x_is_relu_output=True,
                                        dy_pregated=i < 2, chain_out=chain and i < 2)
            kinds.append(Fn.is_operand_images(out))
        out = Fn.relu_gate_f32(out)      # (the last layer's ReLU derivative, as the model's tower applies it)
        out.backward(gy.to(device))
        return out.detach(), xd.grad, [w.grad for w in wd], [b.grad for b in bd], kinds

    y_c, dx_c, dw_c, db_c, kinds_c = run(True)
    y_u, dx_u, dw_u, db_u, kinds_u = run(False)
    assert kinds_c == [True, True, False] and kinds_u == [False, False, False]
    # operand images that ride on a float32 tensor are dropped when the tensor was changed in place since
    with torch.no_grad():
        xd = x.to(device)
        y1 = Fn.conv3d_general_f32(xd, ws[0].to(device), bs[0].to(device), stride=(1, 1, 1), padding=(0, 0, 0), relu=True)
        assert y1._pv_planes[3] == y1._version
        ref2 = Fn.conv3d_general_f32(y1.clone().mul_(2.0), ws[1].to(device), bs[1].to(device), stride=(1, 1, 1), padding=(0, 0, 0), relu=True)
        y1.mul_(2.0)
        got2 = Fn.conv3d_general_f32(y1, ws[1].to(device), bs[1].to(device), stride=(1, 1, 1), padding=(0, 0, 0), relu=True)
        assert torch.equal(got2, ref2)

    def close(a, ref, what, tol=4e-6):
        err = float((a.detach().cpu().double() - ref).abs().max())
        assert err <= tol * float(ref.abs().max()), (what, err, float(ref.abs().max()))
    def close_but_for_flips(a, ref, what):
        # the ReLUs of the two inner layers are not masked: a pre-activation within rounding of zero (a voxel or two per layer)
        # switches a unit in one arithmetic and not the other, and every gradient element that unit feeds moves by up to a few
        # per cent of the largest -- a sparse set, everything else sits at the single-layer bound
        err = (a.detach().cpu().double() - ref).abs()
        scale = float(ref.abs().max())
        off = (err > 4e-6 * scale)
        assert float(off.double().mean()) <= 1e-3 and float(err.max()) <= 0.1 * scale, (what, float(off.double().mean()), float(err.max()), scale)
    for y, dx, dw, db, tag in ((y_c, dx_c, dw_c, db_c, "chained"), (y_u, dx_u, dw_u, db_u, "unchained")):
        close(y, h.detach(), "y " + tag)
        close_but_for_flips(dx, xr.grad * (x > 0), "dx " + tag)
        for i in range(3):
            close(dw[i], wr[i].grad, f"dw{i} " + tag, 5e-3)      # (sums over every voxel: a flipped unit is one term, up to 1e-3 of the largest sum)
            close(db[i], br[i].grad, f"db{i} " + tag, 5e-3)
    # the two arms against each other: the unchained layers read the same operand images (they ride on the float32 tensor) and
    # gate with the same h image, so every bit agrees -- what chaining drops is only the float32 copy nobody reads
    assert torch.equal(y_c, y_u) and torch.equal(dx_c, dx_u)
    for i in range(3):
        assert torch.equal(dw_c[i], dw_u[i]) and torch.equal(db_c[i], db_u[i])


@pytest.mark.parametrize("m,n,k", [(32, 128, 1 << 18), (5, 64, 65536 + 128), (32, 100, 1 << 17), (1, 128, 1 << 16)])
def test_f32_linear_as_streams_over_the_weight(device, monkeypatch, m, n, k):
    """csrc/linear_f32_skinny.hip: fc1-sized F.linear in float32 (<= 32 rows, <= 128 outputs) -- forward and input gradient with
    exact f32 products on the f32 matrix instruction, one pass over the weight each -- against float64, and against the
    split-product GEMM they replace (hip_ops.LINEAR_F32_SKINNY = False).  Bound: 2e-6 of the largest element at these k."""
    K, _ = _mods()
    from predict_pv_yield_amd import functional as Fn
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.01
    b = torch.randn(n, generator=g)
    gy = torch.randn(m, n, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double(), b.double()
    y_ref = F.relu(F.linear(xr, wr, br))
    gy = gy * (y_ref.detach() > 1e-4 * float(y_ref.max())).float()      # (no gradient through pre-activations within rounding of zero)
    (y_ref * gy.double()).sum().backward()

    def run():
        xd = x.to(device).requires_grad_(True)
        wd, bd = w.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
        y = Fn.linear_f32(xd, wd, bd, relu=True)
        y.backward(gy.to(device))
        return y.detach(), xd.grad, wd.grad

    assert K.linear_f32_skinny_covers(m, n, k)
    y, dx, dw = run()
    K.LINEAR_F32_SKINNY = False
    try:
        y2, dx2, dw2 = run()
    finally:
        K.LINEAR_F32_SKINNY = True
    for a, a2, ref, what in ((y, y2, y_ref.detach(), "y"), (dx, dx2, xr.grad, "dx")):
        scale = float(ref.abs().max())
        assert float((a.cpu().double() - ref).abs().max()) <= 2e-6 * scale, what
        assert float((a2.cpu().double() - ref).abs().max()) <= 2e-6 * scale, what + " (GEMM)"
        assert not torch.equal(a, a2), "the two arms are different kernels: identical bits mean the switch is dead"
    assert torch.equal(dw, dw2)      # (the weight gradient is the same call in both arms)
    # the forward's first form (fragments straight from memory, PV_LINEAR_F32_SKINNY_DIRECT=1) against the LDS-staged one: other
    # split-k slabs, so not the same bits
    monkeypatch.setenv("PV_LINEAR_F32_SKINNY_DIRECT", "1")
    y3, _, _ = run()
    assert float((y3.cpu().double() - y_ref.detach()).abs().max()) <= 2e-6 * float(y_ref.abs().max())


def _mask_bits(mask, y_shape):
    """int32 [B,T,hp,wp] relu mask -> bool [B,T,H,W,32]."""
    b, t, h, w, c = y_shape
    m = mask[:, :, :h, :w].to(torch.int64) & 0xFFFFFFFF
    return ((m.unsqueeze(-1) >> torch.arange(32, device=mask.device)) & 1).bool()


def _mask_of(act, K):
    """Reference construction of the padded relu mask of an NDHWC bf16 activation."""
    b, t, h, w, c = act.shape
    bits = ((act > 0).to(torch.int64) << torch.arange(32, device=act.device)).sum(-1)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    mask = torch.zeros(K.relu_mask_shape(b, t, h, w), dtype=torch.int32, device=act.device)
    mask[:, :, :h, :w] = bits
    return mask


def test_first_layer_relu_mask_and_tower_gradients_with_masks(device):
    """The f32-input first-layer kernel (v1 tile geometry) writes the same mask; and a 3-layer tower trained with the masks
    wired through (functional.conv3d_relu_bf16 x_relu_mask / want_relu_mask) has bit-identical gradients to the one that
    gates with the bf16 activations."""
    K, Fn = _mods()
    for (b, ci, t, h, w, pad) in [(2, 11, 7, 16, 16, (0, 0, 0)), (1, 11, 6, 12, 70, (1, 0, 0)), (2, 16, 9, 10, 67, (2, 2, 0))]:
        x, wt, bias = _conv_case(31, b, ci, 32, t, h, w)
        wp = K.conv3d_pack_weight_bf16(wt.to(device))
        y, xp = K.conv3d_fwd_bf16_f32in(x.to(device), wp, bias.to(device), 32, pad, relu=True)
        y2, xp2, mask = K.conv3d_fwd_bf16_f32in(x.to(device), wp, bias.to(device), 32, pad, relu=True, want_relu_mask=True)
        assert torch.equal(y, y2) and torch.equal(xp, xp2)
        assert torch.equal(_mask_bits(mask, y.shape), y > 0)
    # tower: layer 0 (f32 input) -> layer 1 -> layer 2, with and without masks
    torch.manual_seed(5)
    x = torch.randn(2, 11, 9, 20, 40).to(device)
    ws = [(torch.randn(32, 11 if i == 0 else 32, 3, 3, 3) * 0.08).to(device) for i in range(3)]
    bs = [(torch.randn(32) * 0.1).to(device) for _ in range(3)]
    grads = []
    for use_masks in (False, True):
        w = [v.clone().requires_grad_(True) for v in ws]
        bb = [v.clone().requires_grad_(True) for v in bs]
        out = Fn.conv3d_first_layer_bf16(x, w[0], bb[0], (0, 0, 0), relu=True, dy_pregated=True, want_relu_mask=use_masks)
        out, mask = out if use_masks else (out, None)
        out = Fn.conv3d_relu_bf16(out, w[1], bb[1], 32, (0, 0, 0), relu=True, x_is_relu_output=True, dy_pregated=True,
                                  x_relu_mask=mask, want_relu_mask=use_masks)
        out, mask = out if use_masks else (out, None)
        out = Fn.conv3d_relu_bf16(out, w[2], bb[2], 32, (0, 0, 0), relu=True, x_is_relu_output=True, dy_pregated=False,
                                  x_relu_mask=mask)
        g = torch.randn(out.shape, generator=torch.Generator().manual_seed(8)).to(device).to(torch.bfloat16)
        out.backward(g)
        grads.append([v.grad.clone() for v in w + bb])
    for a, c in zip(*grads):
        assert torch.equal(a, c)


def test_repack_gate(device):
    K, _ = _mods()
    dy = torch.randn(2, 32, 3, 5, 6).to(torch.bfloat16)
    yv = torch.relu(torch.randn(2, 32, 3, 5, 6)).to(torch.bfloat16)
    out = K.repack_gate_ncdhw_to_ndhwc_bf16(dy.to(device), yv.to(device)).cpu()
    ref = (dy * (yv > 0)).permute(0, 2, 3, 4, 1)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("m,n,k", [(2, 16, 34816), (5, 128, 1000), (32, 64, 128), (3, 6, 64)])
def test_linear_f32(device, m, n, k):
    K, Fn = _mods()
    g = torch.Generator().manual_seed(m * n)
    x = torch.randn(m, k, generator=g, requires_grad=True)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).requires_grad_(True)
    bias = torch.randn(n, generator=g, requires_grad=True)
    y_ref = F.relu(F.linear(x, w, bias))
    gy = torch.randn(m, n, generator=g)
    y_ref.backward(gy)
    xd, wd, bd = (t.detach().to(device).requires_grad_(True) for t in (x, w, bias))
    y = Fn.linear_f32(xd, wd, bd, relu=True)
    y.backward(gy.to(device))
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wd.grad.cpu(), w.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bd.grad.cpu(), bias.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("m,n,k", [(32, 128, 4096 + 64), (5, 64, 1024), (40, 128, 2048), (3, 6, 64)])
def test_linear_bwd_bf16_gates_dx_by_the_producers_relu(device, m, n, k):
    """pv_linear_bwd_bf16(gate_dx_by_x): dx multiplied by (x > 0) by a vectorised pass behind either dx kernel (the LDS-staged one
    for m <= 32, n in 32..128; the register-tiled one otherwise): bit for bit the ungated dx with the non-positive positions
    (incl. -0) zeroed."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g)
    x[0, :8] = -0.0
    xb = x.to(torch.bfloat16).to(device)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).to(torch.bfloat16).to(device)
    dy = torch.randn(m, n, generator=g).to(device)
    y = torch.rand(m, n, generator=g).sub_(0.3).to(device)
    plain, _, db0 = K.linear_bwd_bf16(xb, w, dy, y, need_dx=True, need_dw=False)
    gated, _, db1 = K.linear_bwd_bf16(xb, w, dy, y, need_dx=True, need_dw=False, gate_dx_by_x=True)
    want = torch.where(xb.float() > 0, plain, torch.zeros_like(plain))
    assert torch.equal(gated.view(torch.int16), want.view(torch.int16)) and torch.equal(db0, db1)
    assert (gated != 0).any() and (gated == 0).sum() > (plain == 0).sum()


@pytest.mark.parametrize("m,n,k", [(2, 16, 34816), (33, 128, 4096 + 64), (64, 128, 8200), (7, 5, 72),
                                   # LDS-staged v2 kernels (m <= 32, n in 32..128): ragged k tiles, partial rows
                                   (32, 128, 256 * 37), (5, 128, 256 * 3 + 72), (32, 64, 1000), (17, 96, 131072 + 8)])
def test_linear_bf16(device, m, n, k):
    K, Fn = _mods()
    g = torch.Generator().manual_seed(m + n)
    x = co.bf16_round(torch.randn(m, k, generator=g)).requires_grad_(True)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k))
    wr = co.bf16_round(w).requires_grad_(True)
    bias = torch.randn(n, generator=g, requires_grad=True)
    y_ref = F.relu(F.linear(x, wr, bias))
    gy = torch.randn(m, n, generator=g)
    y_ref.backward(gy)
    xd = x.detach().to(torch.bfloat16).to(device).requires_grad_(True)
    wd = w.to(device).requires_grad_(True)
    bd = bias.detach().to(device).requires_grad_(True)
    y = Fn.linear_bf16(xd, wd, bd, relu=True)
    y.backward(gy.to(device))
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bd.grad.cpu(), bias.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xd.grad.float().cpu(), x.grad, rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("m,n,k", [(64, 128, 256 * 9 + 72), (96, 128, 128 * 21), (200, 64, 128 * 7 + 8), (513, 128, 1024), (33, 96, 4096)])
def test_linear_fwd_bf16_in_row_blocks_of_64(device, m, n, k):
    """pv_linear_fwd_bf16 with more than 32 rows (a per-GPU batch of 64, the K-sharded fc1's 256 / 512 rows of the global batch):
    the LDS-staged kernel carries two row blocks beside the weight tile (64 rows per stream over the weights) and one reduce
    serves every launch of the call; ragged last blocks (96 = 64 + 32, 200 = 3 x 64 + 8, 513) and ragged k tiles.  Against the
    float64 product of the same bf16 operands, and bit for bit against the same rows taken 32 at a time."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(device)
    w = (torch.randn(n, k, generator=g) / np.sqrt(k)).to(torch.bfloat16).to(device)
    bias = torch.randn(n, generator=g).to(device)
    y = K.linear_fwd_bf16(x, w, bias, relu=True)
    want = torch.relu(x.double() @ w.double().t() + bias.double())
    torch.testing.assert_close(y.double(), want, rtol=1e-5, atol=1e-5)
    y32 = torch.cat([K.linear_fwd_bf16(x[r:r + 32], w, bias, relu=True) for r in range(0, m, 32)])
    assert torch.equal(y, y32)      # (same split of k over workgroups, same order inside a workgroup and in the reduce)


def test_swap01_segments(device):
    """pv_swap01_segments: [n0][n1][seg] -> [n1][n0][seg], the staging copy of the K-sharded fc1's all-to-alls, against torch."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(3)
    for n0, n1, seg, dt in ((4, 8, 1000, torch.bfloat16), (8, 3, 8, torch.bfloat16), (5, 2, 36, torch.float32), (1, 7, 16, torch.int16)):
        x = (torch.randn(n0, n1, seg, generator=g) * 100).to(dt).to(device)
        assert torch.equal(K.swap01_segments(x), x.transpose(0, 1).contiguous())
    with pytest.raises(TypeError):
        K.swap01_segments(torch.zeros(2, 2, 3, dtype=torch.bfloat16, device=device))      # 6-byte segments


def test_forecast_losses(device):
    K, Fn = _mods()
    g = torch.Generator().manual_seed(9)
    y_hat = torch.randn(8, 6, generator=g, requires_grad=True)
    yield_t = torch.rand(8, 19, 128, generator=g)
    y = co.select_target(yield_t, 6)
    mse, nmae, mse_exp, mae_exp = co.forecast_losses(y_hat, y)
    nmae.backward()
    yd = y_hat.detach().to(device).requires_grad_(True)
    out4 = Fn.forecast_losses(yd, co.select_target(yield_t.to(device), 6))  # strided view, no copy
    out4[1].backward()
    assert not out4[0].requires_grad and not out4[2].requires_grad and not out4[3].requires_grad
    ref = torch.stack([mse, nmae, mse_exp, mae_exp]).detach()
    torch.testing.assert_close(torch.stack(out4).detach().cpu(), ref, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(yd.grad.cpu(), y_hat.grad, rtol=1e-6, atol=0)
    # per-horizon metrics from the same launch (base_model.py:121-141)
    four, mse_h, mae_h = Fn.forecast_losses_with_horizons(yd.detach(), co.select_target(yield_t.to(device), 6))
    torch.testing.assert_close(torch.stack(four).cpu(), ref, rtol=1e-5, atol=1e-7)
    ref_mse_h, ref_mae_h = co.horizon_metrics(y_hat.detach(), y)
    torch.testing.assert_close(mse_h.cpu(), ref_mse_h, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(mae_h.cpu(), ref_mae_h, rtol=1e-6, atol=1e-7)


def test_adam_matches_torch(device):
    K, _ = _mods()
    g = torch.Generator().manual_seed(10)
    n = 1000 + 3
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=0.0005)
    p = p0.clone().to(device)
    m = torch.zeros(n, device=device)
    v = torch.zeros(n, device=device)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=device)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        p_ref.grad = grad.clone()
        opt.step()
        K.adam_step(p, grad.to(device), m, v, step, lr=0.0005, bf16_shadow=shadow)
        torch.testing.assert_close(p.cpu(), p_ref.detach(), rtol=2e-6, atol=1e-9)
    st = opt.state[p_ref]
    torch.testing.assert_close(m.cpu(), st["exp_avg"], rtol=2e-6, atol=1e-12)
    torch.testing.assert_close(v.cpu(), st["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    assert torch.equal(shadow.cpu(), p.cpu().to(torch.bfloat16))


def test_fused_wgrad_adam_is_bit_identical_to_two_pass(device):
    """pv_linear_wgrad_adam_bf16 == pv_linear_bwd_bf16(dw) followed by pv_adam_step_f32, bit for bit."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(77)
    m, n, k = 32, 128, 8192
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(device)
    dy = torch.randn(m, n, generator=g).to(device)
    y = torch.relu(torch.randn(m, n, generator=g)).to(device)
    p0 = (torch.randn(n, k, generator=g) * 0.01).to(device)
    wb = p0.to(torch.bfloat16)
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    sa = torch.empty(n, k, dtype=torch.bfloat16, device=device)
    sb = torch.empty(n, k, dtype=torch.bfloat16, device=device)
    for step in (1, 2, 3):
        _, dw, _ = K.linear_bwd_bf16(x, wb, dy, y, need_dx=False)
        K.adam_step(pa, dw, ma, va, step, lr=5e-4, bf16_shadow=sa)
        K.linear_wgrad_adam_bf16(x, dy, y, pb, mb, vb, sb, step, lr=5e-4)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(sa, sb), step


@pytest.mark.parametrize("m,n,k", [(32, 128, 8192), (5, 24, 1024 + 8)])
def test_f32_fused_wgrad_adam_follows_torch_adam_on_the_f64_gradient(device, m, n, k):
    """pv_linear_wgrad_adam_f32 (the f32 model's fc1: gradient from the f32 activations + Adam in one pass): three steps against
    torch.optim.Adam fed with the gradient computed in f64 -- parameters and moments to a few ulps."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(78)
    p0 = torch.randn(n, k, generator=g) * 0.01
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=5e-4)
    pb = p0.clone().to(device)
    mb, vb = torch.zeros_like(pb), torch.zeros_like(pb)
    for step in (1, 2, 3):
        x = torch.randn(m, k, generator=g)
        dy = torch.randn(m, n, generator=g)
        y = torch.relu(torch.randn(m, n, generator=g))
        gated = dy * (y > 0)
        ref.grad = gated.double().t() @ x.double()
        opt.step()
        if step == 2:      # dy already gated by the caller (functional.LinearF32 passes relu_gate_f32's result)
            K.linear_wgrad_adam_f32(x.to(device), gated.to(device), None, pb, mb, vb, step, lr=5e-4)
        else:
            K.linear_wgrad_adam_f32(x.to(device), dy.to(device), y.to(device), pb, mb, vb, step, lr=5e-4)
        st = opt.state[ref]
        # (f32 accumulation of 32 products with mixed signs: absolute error ~ 1e-6 of the largest term)
        assert torch.allclose(mb.cpu().double(), st["exp_avg"], rtol=1e-5, atol=2e-6), step
        assert torch.allclose(vb.cpu().double(), st["exp_avg_sq"], rtol=2e-5, atol=1e-8), step
        # (Adam's update is lr * m / (sqrt(v) + eps): where the gradient is ~0 its sign decides a whole lr)
        assert float((pb.cpu().double() - ref.detach()).abs().max()) <= 2e-6, step


@pytest.mark.parametrize("m,n,k", [(32, 128, 8192), (7, 128, 1024 + 8), (32, 16, 4096), (4, 128, 128 * 33)])
def test_fused_wgrad_dx_adam_single_pass(device, m, n, k):
    """pv_linear_wgrad_dx_adam_bf16: parameters / moments / operand copy bit-identical to pv_linear_wgrad_adam_bf16, and dx
    equal to the dx kernel's (same bf16 hi + lo gradient operand, same bf16 weights; only the f32 summation order of the
    128-deep contraction differs: isolated 1-ulp bf16 flips)."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(79)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(device)
    dy = torch.randn(m, n, generator=g).to(device)
    y = torch.relu(torch.randn(m, n, generator=g)).to(device)
    p0 = (torch.randn(n, k, generator=g) * 0.01).to(device)
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    sa, sb = p0.to(torch.bfloat16), p0.to(torch.bfloat16)
    for step in (1, 2, 3):
        dx_ref, _, _ = K.linear_bwd_bf16(x, sa, dy, y, need_dx=True, need_dw=False)       # from the PRE-update operand copy
        K.linear_wgrad_adam_bf16(x, dy, y, pa, ma, va, sa, step, lr=5e-4)
        dx = K.linear_wgrad_dx_adam_bf16(x, dy, y, pb, mb, vb, sb, step, lr=5e-4)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(sa, sb), step
        d = (dx.float() - dx_ref.float()).abs()
        scale = dx_ref.float().abs().max().item()
        assert d.max().item() <= 2 ** -7 * scale and (d > 0).float().mean().item() < 0.02, (d.max().item(), scale)
    # gate_dx_by_x: the same dx with (x > 0) applied (x as a ReLU output: some zeros, no negatives), bit for bit
    xr = torch.relu(x.float()).to(torch.bfloat16)
    pc, mc, vc, sc = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), p0.to(torch.bfloat16)
    pd, md, vd, sd = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), p0.to(torch.bfloat16)
    dx_plain = K.linear_wgrad_dx_adam_bf16(xr, dy, y, pc, mc, vc, sc, 1, lr=5e-4)
    dx_gated = K.linear_wgrad_dx_adam_bf16(xr, dy, y, pd, md, vd, sd, 1, lr=5e-4, gate_dx_by_x=True)
    assert torch.equal(pc, pd) and torch.equal(sc, sd)
    assert torch.equal(dx_gated, torch.where(xr > 0, dx_plain, torch.zeros_like(dx_plain)))
    assert (xr == 0).any() and (dx_gated == 0).sum() > (dx_plain == 0).sum()
    # moments_tiled: exp_avg / exp_avg_sq held tile by tile ([K/128][N][128]); three steps give the row-major run's numbers
    if k % K.MOMENT_TILE == 0:
        pe, se = p0.clone(), p0.to(torch.bfloat16)
        me, ve = K.moments_to_tiled(torch.zeros_like(p0)), K.moments_to_tiled(torch.zeros_like(p0))
        pf, mf, vf, sf = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), p0.to(torch.bfloat16)
        for step in (1, 2, 3):
            dx_t = K.linear_wgrad_dx_adam_bf16(x, dy, y, pe, me, ve, se, step, lr=5e-4, moments_tiled=True)
            dx_r = K.linear_wgrad_dx_adam_bf16(x, dy, y, pf, mf, vf, sf, step, lr=5e-4)
            assert torch.equal(dx_t, dx_r) and torch.equal(pe, pf) and torch.equal(se, sf), step
            assert torch.equal(K.moments_to_rows(me), mf) and torch.equal(K.moments_to_rows(ve), vf), step
        probe = torch.arange(n * k, dtype=torch.float32, device=device).view(n, k)
        assert torch.equal(K.moments_to_rows(K.moments_to_tiled(probe)), probe)
        assert K.moments_to_tiled(probe).view(k // K.MOMENT_TILE, n, K.MOMENT_TILE)[3, 5, 7] == probe[5, 3 * K.MOMENT_TILE + 7]
    else:
        with pytest.raises(RuntimeError):
            K.linear_wgrad_dx_adam_bf16(x, dy, y, pc, mc, vc, sc, 2, lr=5e-4, moments_tiled=True)


@pytest.mark.parametrize("m,n,k", [(16, 24, 4096), (32, 128, 4096 + 72), (64, 128, 2048), (7, 40, 1000), (70, 128, 1024)])
def test_bf16_gradient_wire_format(device, monkeypatch, m, n, k):
    """pv_linear_wgrad_bf16out = bf16(RNE) of the f32 gradient -- bit for bit in the register-tiled form (PV_WGRAD_BF16OUT_VALU=1,
    and rows beyond 64); the matrix-core form (g as a bf16 hi + lo pair, another summation order in f32) within one bf16 ulp of it
    with more than 99 % of the elements equal; pv_adam_step_bf16grad = Adam on the widened gradient."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(78 + m)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(device)
    dy = torch.randn(m, n, generator=g).to(device)
    y = torch.relu(torch.randn(m, n, generator=g)).to(device)
    wb = torch.zeros(n, k, dtype=torch.bfloat16, device=device)
    _, dw, _ = K.linear_bwd_bf16(x, wb, dy, y, need_dx=False)
    monkeypatch.setenv("PV_WGRAD_BF16OUT_VALU", "1")
    assert torch.equal(K.linear_wgrad_bf16out(x, dy, y, n), dw.to(torch.bfloat16))
    monkeypatch.delenv("PV_WGRAD_BF16OUT_VALU")
    dwb = K.linear_wgrad_bf16out(x, dy, y, n)
    ref = dw.to(torch.bfloat16)
    same = (dwb == ref).float().mean().item()
    gg = torch.where(y > 0, dy, torch.zeros_like(dy)).double()
    exact, terms = torch.einsum("mn,mk->nk", gg, x.double()), torch.einsum("mn,mk->nk", gg.abs(), x.double().abs())
    # one bf16 ulp of the result, plus the hi + lo pair's 2^-16 of the TERMS (a sum that cancels keeps the terms' absolute error)
    ulp = (dwb.double() - ref.double()).abs() <= ref.double().abs() * 2.0 ** -7 + terms * 2.0 ** -15
    assert same > 0.99 and bool(ulp.all()), (same, (~ulp).sum().item())
    assert ((dwb.double() - exact).abs() <= exact.abs() * 2.0 ** -8 + terms * 2.0 ** -15).all()      # half a bf16 ulp of the exact sum
    p0 = (torch.randn(n, k, generator=g) * 0.01).to(device)
    pa, ma, va = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb, mb, vb = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in (1, 2):
        K.adam_step(pa, dwb.float(), ma, va, step, lr=5e-4, grad_scale=0.125)
        K.adam_step_bf16grad(pb, dwb, mb, vb, step, lr=5e-4, grad_scale=0.125)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


def test_adam_multi_tensor_equals_single(device):
    """pv_adam_step_multi_f32 = pv_adam_step_f32 applied tensor by tensor, bit for bit (ragged sizes, optional shadow)."""
    K, _ = _mods()
    g = torch.Generator().manual_seed(12)
    sizes = [1, 3, 1024, 1027, 4096 + 5, 32 * 11 * 27, 6]
    def fresh():
        gg = torch.Generator().manual_seed(13)
        return [[torch.randn(n, generator=gg).to(device) for _ in range(2)] + [torch.rand(n, generator=gg).to(device) * 1e-3]
                for n in sizes]
    grads = [torch.randn(n, generator=g).to(device) for n in sizes]
    a, b = fresh(), fresh()
    sh_a = [torch.zeros(n, dtype=torch.bfloat16, device=device) if i % 2 == 0 else None for i, n in enumerate(sizes)]
    sh_b = [torch.zeros(n, dtype=torch.bfloat16, device=device) if i % 2 == 0 else None for i, n in enumerate(sizes)]
    for step in (1, 2, 7):
        for (p, m, v), gr, sh in zip(a, grads, sh_a):
            K.adam_step(p, gr, m, v, step, lr=3e-3, bf16_shadow=sh, grad_scale=0.5)
        K.adam_step_multi([(p, gr, m, v, sh) for (p, m, v), gr, sh in zip(b, grads, sh_b)], step, lr=3e-3, grad_scale=0.5)
    for (pa, ma, va), (pb, mb, vb), sa, sb in zip(a, b, sh_a, sh_b):
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
        assert sa is None or torch.equal(sa, sb)


def test_pack_weights_multi_equals_single_and_cache_invalidation(device):
    K, Fn = _mods()
    g = torch.Generator().manual_seed(14)
    ws = [torch.randn(32, 11, 3, 3, 3, generator=g).to(device), torch.randn(32, 32, 3, 3, 3, generator=g).to(device),
          torch.randn(20, 32, 3, 3, 3, generator=g).to(device)]
    jobs = []
    for w in ws:
        for flip in (False, True):
            jobs.append((w, torch.zeros_like(K.conv3d_pack_weight_bf16(w, flip)), flip))
    K.conv3d_pack_weights_multi(jobs)
    for w, wp, flip in jobs:
        assert torch.equal(wp, K.conv3d_pack_weight_bf16(w, flip))
    # cache on the parameter: same tensor while the weight is untouched, re-packed after an in-place torch update
    w = torch.nn.Parameter(ws[1].clone())
    wp1 = Fn.packed_conv_weight(w, False)
    assert Fn.packed_conv_weight(w, False) is wp1
    with torch.no_grad():
        w.mul_(2.0)
    wp2 = Fn.packed_conv_weight(w, False)
    assert wp2 is not wp1 and torch.equal(wp2, K.conv3d_pack_weight_bf16(w.detach(), False))
    # refresh after an optimiser-style update through a raw kernel (version unchanged): same buffers, new contents
    K.adam_step(w.detach(), torch.ones_like(w), torch.zeros_like(w), torch.zeros_like(w), 1, lr=0.1)
    Fn.refresh_packed_conv_weights([w])
    assert Fn.packed_conv_weight(w, False) is wp2 and torch.equal(wp2, K.conv3d_pack_weight_bf16(w.detach(), False))
