"""GPU parity: HIP optical-flow advection kernels (through the C ABI) vs the CPU oracle.

Bars: u8 conversion, weighted mean, normalisation and remap are BIT-EXACT (integer
indices, exact fixed-point weights, correctly-rounded f32/f64 arithmetic in a fixed order);
Farnebäck flow within 1e-3 px max-abs of the oracle on identical u8 inputs.
"""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from predict_pv_yield_amd.data.synthetic import advected_counts, blob_texture_sequence

pytestmark = pytest.mark.gpu


def _ops():
    from predict_pv_yield_amd import hip_ops
    return hip_ops


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else a.dtype)


def same_f32(a, b):
    """bit-exact on every non-NaN value, NaNs in the same places (NaN payloads are not part of the contract)"""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(bits(a)[~na], bits(b)[~nb])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", [0, 5, 4096 + 3])
def test_u8_from_10bit_bit_exact(device, mode, n):
    K = _ops()
    rng = np.random.default_rng(n + mode)
    xi = rng.integers(0, 1024, n).astype(np.int16)
    if n > 8:
        xi[:8] = [0, 1, 2, 3, 1021, 1022, 1023, 6]  # ties of x/4 and the top of the range
    xf = xi.astype(np.float32)
    for x in (xi, xf):
        ref, ref_flag = fo.convert_10bpp_to_uint8(x, mode)
        got, flag = K.u8_from_10bit(torch.from_numpy(x).to(device), mode, return_flag=True)
        assert np.array_equal(got.cpu().numpy(), ref)
        assert bool(flag.item()) == ref_flag


def test_u8_range_flag(device):
    K = _ops()
    x = np.array([0, 4, 1023, 1030, -3, 8, 12, 16, 20], np.int16)
    ref, ref_flag = fo.convert_10bpp_to_uint8(x, 0)
    got, flag = K.u8_from_10bit(torch.from_numpy(x).to(device), 0, return_flag=True)
    assert ref_flag and bool(flag.item())
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(3, 6, 8, 8, 2), (2, 11, 64, 64, 2), (1, 5, 7, 9, 2)])
def test_weighted_mean_bit_exact(device, shape):
    K = _ops()
    rng = np.random.default_rng(7)
    flows = rng.normal(0, 3, shape).astype(np.float32)
    got = K.flow_weighted_mean(torch.from_numpy(flows).to(device)).cpu().numpy()
    for g in range(shape[0]):
        ref = fo.weighted_average(flows[g])
        # the numpy one-liner of the reference (optical_flow_1.ipynb:293-294)
        ref_np = np.average(flows[g], axis=0, weights=range(1, shape[1] + 1)).astype(np.float32)
        assert np.array_equal(bits(ref), bits(ref_np))
        assert np.array_equal(bits(got[g]), bits(ref))


def test_normalise_bit_exact(device):
    K = _ops()
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 1024, (2, 5, 3, 16, 16)).astype(np.int16)  # [B,T,C,H,W]
    mean = np.array([93.23458, 131.6, 843.2], np.float32)
    std = np.array([115.34247, 38.1, 53.8], np.float32)
    for x in (raw, raw.astype(np.float32)):
        ref = fo.normalise(x, mean, std, inner=16 * 16)
        got = K.normalise(torch.from_numpy(x).to(device), torch.from_numpy(mean).to(device),
                          torch.from_numpy(std).to(device), inner=16 * 16).cpu().numpy()
        assert np.array_equal(bits(got), bits(ref))


def _random_flow(rng, h, w, scale):
    yy, xx = np.mgrid[0:h, 0:w]
    fl = np.stack([np.sin(yy / 7.0) * scale + rng.normal(0, 0.3, (h, w)),
                   np.cos(xx / 5.0) * scale + rng.normal(0, 0.3, (h, w))], -1)
    return fl.astype(np.float32)


@pytest.mark.parametrize("h,w", [(64, 64), (37, 50), (128, 128)])
@pytest.mark.parametrize("border", [fo.BORDER_CONSTANT, fo.BORDER_REPLICATE])
def test_remap_f32_bit_exact(device, h, w, border):
    K = _ops()
    rng = np.random.default_rng(h * w + border)
    n = 3
    imgs = rng.normal(0, 1, (n, h, w)).astype(np.float32)
    flows = np.stack([_random_flow(rng, h, w, 2.5) for _ in range(n)])
    flows[0] = np.round(flows[0] * 32) / 32           # exact 1/32 fractions
    flows[1, :4] = np.round(flows[1, :4])             # integer shifts
    flows[2, 0, 0] = [np.nan, 1e12]                   # NaN / overflow coordinates
    flows[2, 1, 1] = [0.015625, -0.046875]            # cvRound ties (x.5/32)
    n_steps = 4
    got = K.remap_bilinear(torch.from_numpy(imgs).to(device), torch.from_numpy(flows).to(device), n_steps=n_steps,
                           step0=1.0, border_mode=border, border_value=float("nan")).cpu().numpy()
    for i in range(n):
        for s in range(n_steps):
            ref = fo.remap_image(imgs[i], flows[i], float(1 + s), border, np.nan)
            assert same_f32(got[i, s], ref), (i, s)


@pytest.mark.parametrize("border", [fo.BORDER_CONSTANT, fo.BORDER_REPLICATE])
def test_remap_u8_bit_exact(device, border):
    K = _ops()
    rng = np.random.default_rng(11)
    h, w, n = 48, 64, 2
    imgs = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    flows = np.stack([_random_flow(rng, h, w, 3.0) for _ in range(n)])
    got = K.remap_bilinear(torch.from_numpy(imgs).to(device), torch.from_numpy(flows).to(device), n_steps=3, step0=0.0,
                           border_mode=border, border_value=7).cpu().numpy()
    for i in range(n):
        for s in range(3):
            ref = fo.remap_image(imgs[i], flows[i], float(s), border, 7)
            assert np.array_equal(got[i, s], ref), (i, s)


@pytest.mark.parametrize("dtype", ["f32", "u8"])
@pytest.mark.parametrize("border", [fo.BORDER_CONSTANT, fo.BORDER_REPLICATE])
def test_remap_lds_path_bit_exact(device, dtype, border):
    """>= 128 small images take the LDS-staged kernel: identical to the oracle and to the global-gather kernel."""
    K = _ops()
    rng = np.random.default_rng(77 + border)
    n, h, w = 131, 64, 64
    if dtype == "f32":
        imgs = rng.normal(0, 1, (n, h, w)).astype(np.float32)
        bv = float("nan")
    else:
        imgs = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
        bv = 9
    flows = np.stack([_random_flow(rng, h, w, 4.0) for _ in range(n)])
    flows[5, 0, 0] = [np.nan, 1e12]
    flows[6] *= 20.0                                   # mostly outside the image
    t_img, t_fl = torch.from_numpy(imgs).to(device), torch.from_numpy(flows).to(device)
    got = K.remap_bilinear(t_img, t_fl, n_steps=3, step0=1.0, border_mode=border, border_value=bv).cpu().numpy()
    small = K.remap_bilinear(t_img[:40], t_fl[:40], n_steps=3, step0=1.0, border_mode=border, border_value=bv).cpu().numpy()
    eq = same_f32 if dtype == "f32" else np.array_equal
    assert eq(got[:40], small)
    for i in (0, 5, 6, 130):
        for s in range(3):
            ref = fo.remap_image(imgs[i], flows[i], float(1 + s), border, bv if dtype == "u8" else np.nan)
            assert eq(got[i, s], ref), (i, s)


def _pair(rng, h, w, v):
    seq = blob_texture_sequence(rng, 2, h, w, v)
    u8, _ = fo.convert_10bpp_to_uint8(np.clip(np.rint(seq), 0, 1023).astype(np.int16), 0)
    return u8


@pytest.mark.parametrize("h,w", [(64, 64), (32, 48), (96, 80), (160, 200), (65, 67), (20, 200), (131, 70)])     # (the last three: frame kernels on odd sizes)
def test_farneback_vs_oracle(device, h, w):
    K = _ops()
    rng = np.random.default_rng(h + w)
    vels = [(1.0, 0.0), (1.5, 0.75), (-2.25, 1.3)]
    pairs = [_pair(rng, h, w, v) for v in vels]
    prev = torch.from_numpy(np.stack([p[0] for p in pairs])).to(device)
    nxt = torch.from_numpy(np.stack([p[1] for p in pairs])).to(device)
    got = K.farneback_pairs(prev, nxt).cpu().numpy()
    for i, p in enumerate(pairs):
        ref = fo.calc_optical_flow_farneback(p[0], p[1])
        err = np.abs(got[i] - ref).max()
        assert err <= 1e-3, (i, err)
    # known answer: interior flow recovers the synthetic translation (K2)
    inner = got[1][12:-12, 12:-12] if min(h, w) > 40 else got[1]
    if min(h, w) >= 64:
        assert abs(np.median(inner[..., 0]) - 1.5) < 0.1 and abs(np.median(inner[..., 1]) - 0.75) < 0.1


def test_farneback_full_extent_vs_oracle(device):
    """The reference's own image size (704 x 548 HRV frames, 13_...ipynb; two coarse pyramid levels): one pair."""
    K = _ops()
    rng = np.random.default_rng(704)
    u8 = _pair(rng, 704, 548, (2.2, -1.4))
    assert fo.farneback_num_levels(704, 548) == 2
    got = K.farneback_pairs(torch.from_numpy(u8[0:1]).to(device), torch.from_numpy(u8[1:2]).to(device))[0].cpu().numpy()
    ref = fo.calc_optical_flow_farneback(u8[0], u8[1])
    assert got.shape == ref.shape == (704, 548, 2)
    assert np.abs(got - ref).max() < 1e-3
    inner = ref[60:-60, 60:-60]
    # and both recover the known translation in the interior (algorithmic accuracy of Farnebäck on this texture)
    assert np.median(np.abs(inner[..., 0] - 2.2)) < 0.1 and np.median(np.abs(inner[..., 1] + 1.4)) < 0.1
    assert np.percentile(np.abs(inner[..., 0] - 2.2), 95) < 0.25 and np.percentile(np.abs(inner[..., 1] + 1.4), 95) < 0.25


@pytest.mark.parametrize("h,w,n", [(548, 704, 3), (160, 200, 2), (96, 80, 5), (70, 131, 2), (65, 67, 1), (20, 200, 2), (200, 20, 2),
                                   (64, 65, 1), (65, 64, 3), (33, 300, 1)])
def test_frame_window_blur_in_register_windows_is_bit_identical_to_one_load_per_tap(device, monkeypatch, h, w, n):
    """Levels larger than a 64 x 64 tile (the notebooks' 704 x 548 frames): the 41-tap window blur as sliding register
    windows over a transposed intermediate (fb_blur_v_run_kernel / fb_blur_h_solve_run_kernel) performs the operations of the
    one-load-per-tap kernels in the same order -- the flow is the same bits, for any height / width (odd ones, runs that end
    beyond the image, clamped borders)."""
    K = _ops()
    rng = np.random.default_rng(h * 1000 + w)
    u8 = torch.from_numpy(rng.integers(0, 256, (n + 1, h, w), dtype=np.uint8)).to(device)
    fast = K.farneback_stack(u8)
    monkeypatch.setenv("PV_FARNEBACK_TAP_LOADS", "1")
    slow = K.farneback_stack(u8)
    monkeypatch.delenv("PV_FARNEBACK_TAP_LOADS")
    assert torch.isfinite(fast).all()
    assert torch.equal(fast, slow)


@pytest.mark.parametrize("h,w,n,kw", [
    (548, 704, 2, {}),                                   # the notebooks' frames: levels at scale 1, 1/2 (2 x 2 mean), 1/4 (bilinear)
    (160, 200, 2, {}),
    (131, 70, 3, {}),                                    # odd sizes: the coarse level is no exact half
    (97, 150, 2, dict(pyr_scale=0.7, levels=3)),         # every coarse level resampled bilinearly, 5- and 7-tap smoothing
    (200, 120, 1, dict(poly_n=7, poly_sigma=1.5, levels=1)),
    (20, 200, 2, {}), (200, 20, 1, {}), (64, 65, 2, {}), (65, 64, 1, {}), (33, 300, 1, dict(levels=3)),
])
def test_frame_prep_polyexp_in_one_kernel_is_bit_identical_to_the_three_kernels(device, monkeypatch, h, w, n, kw):
    """Frames larger than a tile: smoothing, resize and both PolyExp passes of a level as ONE launch with the stages in LDS
    (fb_prep_polyexp_frame_kernel) against the three global-memory kernels: same expressions, same bits in the flow."""
    K = _ops()
    rng = np.random.default_rng(h * 1000 + w)
    u8 = torch.from_numpy(rng.integers(0, 256, (n + 1, h, w), dtype=np.uint8)).to(device)
    fused = K.farneback_stack(u8, **kw)
    pairs = K.farneback_pairs(u8[:-1].contiguous(), u8[1:].contiguous(), **kw)     # unchained image indexing
    monkeypatch.setenv("PV_FARNEBACK_THREE_KERNEL_POLYEXP", "1")
    three = K.farneback_stack(u8, **kw)
    monkeypatch.delenv("PV_FARNEBACK_THREE_KERNEL_POLYEXP")
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, three)
    assert torch.equal(fused, pairs)


def test_farneback_stack_and_params(device):
    K = _ops()
    raw, _ = advected_counts(batch=1, t=5, channels=2, h=64, w=64, seed=5)
    stacks = np.ascontiguousarray(raw[0].transpose(1, 0, 2, 3))  # [C, T, H, W]
    u8, _ = fo.convert_10bpp_to_uint8(stacks, 0)
    got = K.farneback_stack(torch.from_numpy(u8).to(device), levels=3, winsize=15, iterations=2, poly_n=7,
                            poly_sigma=1.5).cpu().numpy()
    assert got.shape == (2, 4, 64, 64, 2)
    for c in range(2):
        for t in range(4):
            ref = fo.calc_optical_flow_farneback(u8[c, t], u8[c, t + 1], levels=3, winsize=15, iterations=2, poly_n=7,
                                                 poly_sigma=1.5)
            assert np.abs(got[c, t] - ref).max() <= 1e-3


@pytest.mark.parametrize("h,w,t", [(64, 64, 12), (40, 56, 3), (96, 80, 4), (64, 64, 2)])
def test_farneback_stack_computes_each_frame_once_bit_identically(device, monkeypatch, h, w, t):
    """Consecutive frames of a stack: the per-image stages (smoothing, resize, PolyExp) run once per FRAME instead of once
    per pair side.  The flows must equal, bit for bit, those of the unchained path (PV_FARNEBACK_NO_FRAME_CHAIN=1) and of
    the same pairs handed over as separate prev / next tensors (tile kernels at 64 x 64, generic kernels above)."""
    K = _ops()
    raw, _ = advected_counts(batch=2, t=t, channels=2, h=h, w=w, seed=h + t)
    stacks = np.ascontiguousarray(raw.transpose(0, 2, 1, 3, 4)).reshape(4, t, h, w)  # [B*C, T, H, W]
    u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(stacks, 0)[0]).to(device)
    chained = K.farneback_stack(u8)
    monkeypatch.setenv("PV_FARNEBACK_NO_FRAME_CHAIN", "1")
    unchained = K.farneback_stack(u8)
    monkeypatch.delenv("PV_FARNEBACK_NO_FRAME_CHAIN")
    assert chained.shape == (4, t - 1, h, w, 2)
    assert torch.equal(chained, unchained)
    prev = u8[:, :-1].reshape(-1, h, w).contiguous()
    nxt = u8[:, 1:].reshape(-1, h, w).contiguous()
    assert torch.equal(chained.reshape(-1, h, w, 2), K.farneback_pairs(prev, nxt))
    ref = fo.calc_optical_flow_farneback(u8[1, 0].cpu().numpy(), u8[1, 1].cpu().numpy())
    assert np.abs(chained[1, 0].cpu().numpy() - ref).max() <= 1e-3


@pytest.mark.parametrize("h,w,t,batch", [(64, 64, 12, 3), (64, 64, 4, 100), (64, 64, 12, 100), (40, 56, 3, 3), (48, 64, 4, 3),
                                         (64, 36, 2, 3), (50, 62, 3, 3)])
def test_fused_iteration_against_the_two_launch_form(device, monkeypatch, h, w, t, batch):
    """Levels up to 64 x 64 run every iteration's UpdateMatrices + window blur + solve in ONE launch per level
    (fb_level_u_kernel: eight uniform waves, M never written; levels up to 32 x 32 as four pairs per 64 x 64 tile with
    block-diagonal window matrices).  UpdateMatrices is the same arithmetic as in the two-launch form
    (PV_FARNEBACK_TWO_LAUNCH_ITERATION=1); the blur runs on two-term f16 operands (2^-22) and the solve in compensated f32
    instead of three-term bf16 and f64, so the flows agree to a few 1e-6 px, not bit for bit (bound: 2e-5 px, 50 x under
    the 1e-3 px contract).  What MUST be identical bits: the same pairs handed over as frame stacks (chained R images,
    ranges of pairs that cross stack boundaries, a last unit of fewer than four pairs) and as separate prev / next tensors
    -- a pair's flow may not depend on which other pairs share its launch, its workgroup or its 64 x 64 mosaic (the f16
    operand scale is per pair).  A width that is no multiple of 4 is not taken by the fused form: both calls then run the
    same kernels and must agree exactly."""
    K = _ops()
    # (both forms on the vector-ALU PolyExp, which is the only one the two-launch form has: this test compares the ITERATION
    # kernels; the matrix-core PolyExp has its own test below)
    monkeypatch.setenv("PV_FARNEBACK_POLYEXP_VALU", "1")
    # batch = 100: 200 stacks x 3 pairs = 600 pairs over 256 workgroups -- ranges of 2..3 pairs that cross stack boundaries;
    # with t = 12 (2 200 pairs) the coarse level's units of four pairs come in ranges of two or more as well
    raw, _ = advected_counts(batch=batch, t=t, channels=2, h=h, w=w, seed=3 * h + t)
    stacks = np.ascontiguousarray(raw.transpose(0, 2, 1, 3, 4)).reshape(2 * batch, t, h, w)
    u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(stacks, 0)[0]).to(device)
    prev = u8[:, :-1].reshape(-1, h, w).contiguous()
    nxt = u8[:, 1:].reshape(-1, h, w).contiguous()
    fused_stack, fused_pairs = K.farneback_stack(u8), K.farneback_pairs(prev, nxt)
    fused_it1 = K.farneback_stack(u8, iterations=1)
    monkeypatch.setenv("PV_FARNEBACK_TWO_LAUNCH_ITERATION", "1")
    two_stack, two_pairs = K.farneback_stack(u8), K.farneback_pairs(prev, nxt)
    two_it1 = K.farneback_stack(u8, iterations=1)
    monkeypatch.delenv("PV_FARNEBACK_TWO_LAUNCH_ITERATION")
    assert torch.isfinite(fused_stack).all()
    assert torch.equal(fused_stack.reshape(-1, h, w, 2), fused_pairs) and torch.equal(two_stack.reshape(-1, h, w, 2), two_pairs)
    if w % 4:
        assert torch.equal(fused_stack, two_stack) and torch.equal(fused_it1, two_it1)
    else:
        assert not torch.equal(fused_stack, two_stack), "the two forms are different kernels: identical bits mean the switch is dead"
        for a, b in ((fused_stack, two_stack), (fused_it1, two_it1)):
            assert float((a - b).abs().max()) <= 2e-5, float((a - b).abs().max())
    # one pair evaluated alone (a launch of its own) has the bits it has inside the batch
    alone = K.farneback_pairs(prev[5:6].contiguous(), nxt[5:6].contiguous())
    assert torch.equal(alone[0], fused_pairs[5])
    ref = fo.calc_optical_flow_farneback(u8[2, 0].cpu().numpy(), u8[2, 1].cpu().numpy())
    assert np.abs(fused_stack[2, 0].cpu().numpy() - ref).max() <= 1e-3


@pytest.mark.parametrize("h,w,t,stacks,kw", [
    (64, 64, 12, 40, {}),                                                    # the PV-site tiles, default parameters
    (64, 64, 3, 300, {}),                                                    # 16-17 images per workgroup at level 0
    (40, 56, 4, 7, dict(levels=3, iterations=2)),                            # margins; the 20 x 28 and 10 x 14 levels as mosaics
    (24, 28, 3, 11, dict(levels=2, winsize=9)),                              # a source that is itself a mosaic tile (mode 0)
    (64, 36, 2, 5, dict(levels=2, winsize=21, poly_n=7, poly_sigma=1.5)),    # 15-tap PolyExp
])
def test_polyexp_on_the_matrix_cores_against_the_vector_alu_form(device, monkeypatch, h, w, t, stacks, kw):
    """fb_prep_polyexp_mfma_kernel (the two PolyExp passes as nine products with two-term half-float operands) against
    fb_prep_polyexp_tile_kernel (tap loops, PV_FARNEBACK_POLYEXP_VALU=1): different summation order and operand precision
    (2^-22), so not bit for bit -- the flows built on them agree to 1e-4 px, both sit within the 1e-3 px contract of the
    oracle, and the same pairs as stacks and as separate tensors stay bit-identical."""
    K = _ops()
    raw, _ = advected_counts(batch=stacks, t=t, channels=1, h=h, w=w, seed=7 * h + w)
    u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(np.ascontiguousarray(raw[:, :, 0]), 0)[0]).to(device)
    mfma = K.farneback_stack(u8, **kw)
    pairs = K.farneback_pairs(u8[:, :-1].reshape(-1, h, w).contiguous(), u8[:, 1:].reshape(-1, h, w).contiguous(), **kw)
    monkeypatch.setenv("PV_FARNEBACK_POLYEXP_VALU", "1")
    valu = K.farneback_stack(u8, **kw)
    assert bool(torch.isfinite(mfma).all()) and torch.equal(mfma.reshape(-1, h, w, 2), pairs)
    assert not torch.equal(mfma, valu), "the two forms are different kernels: identical bits mean the switch is dead"
    assert float((mfma - valu).abs().max()) <= 1e-4, float((mfma - valu).abs().max())
    ref = fo.calc_optical_flow_farneback(u8[1, 0].cpu().numpy(), u8[1, 1].cpu().numpy(), **kw)
    assert np.abs(mfma[1, 0].cpu().numpy() - ref).max() <= 1e-3 and np.abs(valu[1, 0].cpu().numpy() - ref).max() <= 1e-3


def test_level_kernel_repeated_launches_give_identical_bits(device):
    """1 200 pairs (4-5 units per workgroup, the next unit's prefetches under a unit's last iteration in play), ten launches: identical
    bits every time.  (Round 4: a scheduling hint in the multiplying waves' operand reads made repeated launches differ by
    ~1e-6 px -- a read of a hand-over buffer ahead of its barrier; tools/stress_flow_fused.py is the longer form of this.)"""
    K = _ops()
    raw, _ = advected_counts(batch=300, t=3, channels=2, h=64, w=64, seed=11)
    stacks = np.ascontiguousarray(raw.transpose(0, 2, 1, 3, 4)).reshape(600, 3, 64, 64)
    u8 = torch.from_numpy(fo.convert_10bpp_to_uint8(stacks, 0)[0]).to(device)
    first = K.farneback_stack(u8)
    for _ in range(10):
        assert torch.equal(K.farneback_stack(u8), first)


@pytest.mark.parametrize("dtype", [torch.int16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 5, 3, 8, 8), (1, 12, 11, 64, 64), (3, 2, 1, 4, 6)])
def test_prepare_stacks_equals_permute_u8_normalise(device, dtype, shape):
    """pv_prepare_stacks_* (one pass over the time-major raw counts) = permute + pv_u8_from_10bit + pv_normalise, bit for
    bit; the slices past T of the model input stay untouched."""
    from predict_pv_yield_amd import hip_ops as K
    b, t, c, h, w = shape
    g = torch.Generator().manual_seed(5)
    raw = torch.randint(0, 1024, shape, generator=g).to(dtype).to(device)
    mean = (torch.rand(c, generator=g) * 300 + 100).to(device)
    std = (torch.rand(c, generator=g) * 100 + 50).to(device)
    u8, out = K.prepare_stacks(raw, mean, std, t + 3)
    stacks = raw.permute(0, 2, 1, 3, 4).contiguous()
    assert torch.equal(u8, K.u8_from_10bit(stacks, 0))
    assert torch.equal(out[:, :, :t], K.normalise(stacks, mean, std, inner=t * h * w))


def test_errors_are_loud(device):
    K = _ops()
    with pytest.raises(RuntimeError):
        K.u8_from_10bit(torch.zeros(8, dtype=torch.int16), 0)  # CPU tensor: no CPU path
    with pytest.raises(RuntimeError):
        K.farneback_pairs(torch.zeros((1, 64, 64), dtype=torch.uint8, device=device),
                          torch.zeros((1, 64, 64), dtype=torch.uint8, device=device), flags=0)


@pytest.mark.gpu
def test_advecting_loader_matches_inline_pipeline(device):
    """optical_flow.AdvectingLoader (the advection in front of the model, on the training stream) hands out exactly what the inline call
    computes, for every batch and in order, and Model(future_frames="optical_flow") consumes the tagged tensor as is."""
    from predict_pv_yield_amd import optical_flow as of
    g = torch.Generator(device=device).manual_seed(7)
    raws = [(torch.rand(2, 12, 11, 64, 64, generator=g, device=device) * 1023).to(torch.int16) for _ in range(3)]
    batches = [{"satellite": {"data": r}, "tag": i} for i, r in enumerate(raws)]
    want = [of.advect_future_frames(r, 6) for r in raws]
    got = list(of.AdvectingLoader(batches, n_future=6))
    assert [b["tag"] for b in got] == [0, 1, 2]
    torch.cuda.synchronize()
    for w, b in zip(want, got):
        x = b["satellite"]["data"]
        assert getattr(x, "_pv_advected", False) and x.shape == w.shape
        assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(w, nan=-7.0))
    assert batches[0]["satellite"]["data"].dtype == torch.int16          # the caller's batches are not modified
    assert list(of.AdvectingLoader([], n_future=6)) == []

    from predict_pv_yield_amd.models.conv3d.model import Model
    torch.manual_seed(0)
    model = Model(include_pv_yield=False, include_nwp=False, forecast_minutes=30, history_minutes=55, number_of_conv3d_layers=4,
                  conv3d_channels=32, image_size_pixels=64, number_sat_channels=11, fc1_output_features=128,
                  fc2_output_features=128, fc3_output_features=64, output_variable="pv_yield", precision="bf16",
                  future_frames="optical_flow").to(device)
    with torch.no_grad():
        y_inline = model({"satellite": {"data": raws[1]}, "pv": {"pv_yield": torch.rand(2, 18, 128, device=device)}})
        y_loader = model({"satellite": {"data": got[1]["satellite"]["data"]}, "pv": {"pv_yield": torch.rand(2, 18, 128, device=device)}})
    assert torch.equal(y_inline, y_loader)


def test_hip_forecasts_score_like_the_oracles_under_skimage_ssim(device):
    """The reference's own quality number for its optical-flow forecasts (notebooks/optical_flow_1.ipynb cells 31, 35, 38:
    metrics.structural_similarity(ground_truth, remap_image(image_t0, flow * i))): the HIP pipeline -- Farneback of the
    warm-up pairs, weighted average, uint8 remap with BORDER_REPLICATE -- on the frames of tests/golden/ssim_skimage.npz,
    scored with oracle/ssim.py (pinned to scikit-image 0.18.3's scores by tests/test_oracle_flow.py), against the scores
    scikit-image gave the oracle's forecasts.  Bounds: each score within 1e-3 of the golden; forecasts differ from the oracle's
    in at most 1e-3 of the pixels (a flow that differs by 1e-6 px can move a fixed-point remap result by one count)."""
    import os
    from oracle.ssim import structural_similarity
    from predict_pv_yield_amd import optical_flow as of
    K = _ops()
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_skimage.npz"))
    frames, w0 = d["frames"], int(d["warm_up"])
    n_steps = d["forecasts"].shape[0] - 1
    fr = torch.from_numpy(frames[:w0]).to(device)
    flows = K.farneback_pairs(fr[:-1].contiguous(), fr[1:].contiguous())
    flow = K.flow_weighted_mean(flows[None])[0]
    assert np.abs(flow.cpu().numpy() - d["flow"]).max() <= 1e-3
    got = K.remap_bilinear(fr[w0 - 1:w0].contiguous(), flow[None].contiguous(), n_steps=n_steps, step0=1.0,
                           border_mode=of.BORDER_REPLICATE, border_value=0)[0].cpu().numpy()
    for i in range(1, n_steps + 1):
        assert (got[i - 1] != d["forecasts"][i]).mean() <= 1e-3, i
        score = structural_similarity(frames[w0 - 1 + i], got[i - 1])
        assert abs(score - d["ssim_flow"][i]) <= 1e-3, (i, score, d["ssim_flow"][i])
        assert score > d["ssim_persistence"][i] + 0.05


def test_ssim_kernel_against_the_oracle_and_the_skimage_goldens(device):
    """pv_ssim_mean_{u8,f32} (skimage.metrics.structural_similarity's defaults; optical_flow_1.ipynb cells 31, 35, 38) against
    oracle/ssim.py -- which tests/test_oracle_flow.py pins to scikit-image 0.18.3's own scores -- on the golden forecasts
    (scores within 1e-9 of the ones scikit-image gave), on random uint8 and float32 stacks incl. a non-square image, and the
    loud refusal of an image smaller than the window."""
    import os
    from oracle.ssim import structural_similarity as ssim_ref
    from predict_pv_yield_amd import optical_flow as of
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_skimage.npz"))
    frames, w0 = d["frames"], int(d["warm_up"])
    n = d["forecasts"].shape[0]
    truth = torch.from_numpy(frames[w0 - 1:w0 - 1 + n]).to(device)
    got = of.structural_similarity(truth, torch.from_numpy(d["forecasts"]).to(device)).cpu().numpy()
    np.testing.assert_allclose(got, d["ssim_flow"], rtol=0, atol=1e-9)
    pers = of.structural_similarity(truth, truth[0:1].expand(n, -1, -1).contiguous()).cpu().numpy()
    np.testing.assert_allclose(pers, d["ssim_persistence"], rtol=0, atol=1e-9)
    rng = np.random.default_rng(5)
    for shape, dtype in (((3, 64, 64), np.uint8), ((2, 33, 71), np.uint8), ((2, 40, 40), np.float32)):
        a = rng.integers(0, 256, shape).astype(dtype) if dtype == np.uint8 else rng.normal(0, 0.3, shape).astype(dtype)
        b = np.clip(a.astype(np.float64) + rng.normal(0, 20 if dtype == np.uint8 else 0.05, shape), 0 if dtype == np.uint8 else -9, 255 if dtype == np.uint8 else 9).astype(dtype)
        got = of.structural_similarity(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)).cpu().numpy()
        want = [ssim_ref(a[i], b[i]) for i in range(shape[0])]
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
        assert abs(of.structural_similarity(a[0], b[0]) - want[0]) <= 1e-9            # the cv2-style single-image call
    with pytest.raises(RuntimeError, match="at least 7 x 7"):
        of.structural_similarity(np.zeros((6, 20), np.uint8), np.zeros((6, 20), np.uint8))


def test_parameter_search_of_optical_flow_1(device):
    """notebooks/optical_flow_1.ipynb cells 38-42: one Farneback argument at a time varied around the defaults, a setting scored
    by the mean structural similarity of its forecasts remap_image(images[1], flow * i) against images[i + 1].  On a synthetic
    advected sequence: every score within 2e-3 of the oracle's (pv_oracle.c flow + remap, oracle/ssim.py), the same winner per
    parameter where the oracle's margin is clear, NaN for a value the library does not take (poly_n = 9), and the
    reference's arguments beat a 5-pixel window (what the notebook's search found)."""
    from oracle import flow_oracle as fo
    from oracle.ssim import structural_similarity as ssim_ref
    from predict_pv_yield_amd import optical_flow as of
    from predict_pv_yield_amd.data.synthetic import advected_counts
    steps = 8
    raw, _ = advected_counts(batch=1, t=steps + 1, channels=1, h=64, w=64, seed=77, vmax=2.0)
    images = fo.convert_10bpp_to_uint8(raw[0, :, 0].astype(np.float32))[0]                   # [T, 64, 64] uint8
    ranges = dict(winsize=[5, 40], iterations=[1, 3], poly_sigma=[0.7, 1.2], poly_n=[5, 9])
    scores, durations, best = of.search_farneback_params(torch.from_numpy(images).to(device), param_ranges=ranges, num_timesteps=steps)
    assert set(scores) == {(k, v) for k, vs in ranges.items() for v in vs} and set(durations) == set(scores)
    assert scores[("poly_n", 9)] != scores[("poly_n", 9)] and best["poly_n"] == 5           # NaN: not taken, never wins

    def oracle_score(**kw):
        params = dict(of.REFERENCE_FARNEBACK_KWARGS)
        params.update(kw)
        params.pop("flags")
        flow = fo.calc_optical_flow_farneback(images[0], images[1], **params)
        return float(np.mean([ssim_ref(images[i + 1], fo.remap_image(images[1], flow, k=float(i), border_mode=fo.BORDER_REPLICATE))
                              for i in range(1, steps)]))
    ref = {}
    for name, vals in ranges.items():
        for v in vals:
            if (name, v) == ("poly_n", 9):
                continue
            ref[(name, v)] = oracle_score(**{name: v})
            assert abs(scores[(name, v)] - ref[(name, v)]) <= 2e-3, (name, v, scores[(name, v)], ref[(name, v)])
    for name, vals in ranges.items():
        rs = sorted(((ref[(name, v)], v) for v in vals if (name, v) in ref), reverse=True)
        if len(rs) > 1 and rs[0][0] - rs[1][0] > 5e-3:
            assert best[name] == rs[0][1], (name, best[name], rs)
    assert scores[("winsize", 40)] > scores[("winsize", 5)]
    single = of.compute_opt_flow_and_score(images, num_timesteps=steps)                      # NumPy in, reference defaults
    assert abs(float(np.mean(single)) - scores[("winsize", 40)]) <= 1e-12
