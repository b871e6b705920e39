"""GPU parity tests: every C-ABI kernel against the CPU oracle / plain PyTorch fp32 on the same
seeded inputs.  Run on the MI355X box with ``pytest -m gpu``."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, rel_err
from oracle import unet_ref as U
from oracle.haar_ref import haar_dwt2d, haar_idwt2d
from oracle.make_golden import TINY, FULL, case_inputs, randn

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_mode():
    """Sampling runs under torch.no_grad() in the reference (diffusion_general.py:154, :203, :210): these tests exercise
    that (fused inference) path of WavBEST.forward; the differentiable path is covered by test_gpu_training.py /
    test_gpu_backward.py."""
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from tmdiff_amd import ops as _ops
    return _ops


def cu(t):
    return t.cuda().contiguous()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 6, 8, 8), (1, 16, 16, 12), (3, 5, 64, 64), (2, 3, 6, 10), (1, 1, 2, 2)])
def test_haar_dwt_idwt(ops, shape, golden):
    x = randn(11, *shape)
    want = haar_dwt2d(x)
    got = ops.haar_dwt2d(cu(x))
    for w, g in zip(want, got):
        assert_close(g.cpu(), w, 1e-6, 1e-6, "dwt band")
    ll_only = ops.haar_dwt2d(cu(x), want_high=False, ll_scale=0.5)
    assert ll_only[1] is None
    assert torch.equal(ll_only[0].cpu(), got[0].cpu() * 0.5)          # /2 is exact
    bands = [randn(20 + i, *want[0].shape) for i in range(4)]
    y = ops.haar_idwt2d([cu(bands[0])], *[cu(b) for b in bands[1:]])[0]
    assert_close(y.cpu(), haar_idwt2d(*bands), 1e-6, 1e-6, "idwt")
    rec = ops.haar_idwt2d([got[0]], got[1], got[2], got[3])[0]
    assert_close(rec.cpu(), x, 2e-6, 2e-6, "perfect reconstruction")
    if shape == (2, 6, 8, 8):
        g = golden("haar")
        assert_close(got[0].cpu(), g["a_ll"], 1e-6, 1e-6, "dwt vs reference fixture")
        assert_close(y.cpu(), g["a_idwt"], 1e-6, 1e-6, "idwt vs reference fixture")


def test_haar_idwt_two_lowbands_stacked(ops):
    b, c, n, h, w = 2, 3, 4, 8, 8
    hb, xb = randn(1, b, c, n, h, w), randn(2, b, c, n, h, w)
    bands = randn(3, b, 3 * c, n, h, w)
    outs = ops.haar_idwt2d([cu(hb), cu(xb)], None, None, None, in_scale=2.0, stacked_bands=cu(bands))
    fold = lambda v: v.reshape(b, -1, h, w)
    lh, hl, hh = fold(bands[:, :c]), fold(bands[:, c:2 * c]), fold(bands[:, 2 * c:])
    for got, low in zip(outs, (hb, xb)):
        want = haar_idwt2d(2.0 * fold(low), lh, hl, hh).reshape(b, c, n, 2 * h, 2 * w)
        assert_close(got.cpu(), want, 1e-6, 1e-6, "stacked idwt")


def test_haar_rejects_odd_sizes(ops):
    from tmdiff_amd._lib import TmdiffError
    with pytest.raises(TmdiffError):
        ops.haar_dwt2d(torch.zeros(1, 1, 3, 4, device="cuda"))


# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, Cin, Cout, N, H, W, k, groups
    (2, 8, 8, 4, 8, 8, 3, 1),
    (1, 5, 7, 4, 6, 6, 3, 1),       # odd channels, partial tile
    (2, 32, 64, 8, 16, 16, 3, 1),
    (1, 64, 64, 8, 16, 16, 3, 1),
    (3, 4, 12, 8, 2, 2, 3, 1),      # tiny spatial
    (2, 12, 8, 3, 10, 12, 3, 1),    # N=3, ragged
    (2, 16, 32, 8, 16, 16, 1, 1),
    (1, 5, 3, 4, 6, 6, 1, 1),
    (2, 24, 12, 8, 8, 8, 3, 3),     # grouped
    (1, 96, 32, 8, 8, 8, 3, 1),
    (1, 40, 96, 4, 8, 8, 3, 1),     # cout 96: one full + one partial 64-tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_plain(ops, case):
    b, ci, co, n, h, w, k, g = case
    x = randn(1, b, ci, n, h, w)
    wt = randn(2, co, ci // g, k, k, k) / (ci // g * k ** 3) ** 0.5
    bias = randn(3, co)
    want = F.conv3d(x, wt, bias, 1, k // 2, 1, g)
    wp = ops.pack_conv_weight(cu(wt), groups=g)
    got = ops.conv3d([cu(x)], wp, co, k, groups=g, bias=cu(bias))
    assert_close(got.cpu(), want, 1e-5, 1e-5, f"conv3d {case}")


def test_conv3d_fused_prologue_epilogue_segments(ops):
    b, n, h, w = 3, 8, 16, 16
    segs = [randn(10 + i, b, c, n, h, w) for i, c in enumerate((8, 16, 8))]
    ci, co = 32, 24
    wt = randn(2, co, ci, 3, 3, 3) / (ci * 27) ** 0.5
    bias, shift, scale = randn(3, co), randn(4, b, ci), 1 + 0.3 * randn(5, b, ci)
    res = randn(6, b, co, n, h, w)
    mask = (torch.rand(b, ci, n, h, w, generator=torch.Generator().manual_seed(7)) > 0.2).float() / 0.8
    xin = torch.cat(segs, 1)
    xp = U.silu(xin + shift[:, :, None, None, None]) * scale[:, :, None, None, None] * mask
    want = (F.conv3d(xp, wt, None, 1, 1) + 2.0 * bias[None, :, None, None, None] + res) * 0.5
    wp = ops.pack_conv_weight(cu(wt))
    got = ops.conv3d([cu(s) for s in segs], wp, co, 3, bias=cu(bias), bias_scale=2.0, in_shift=cu(shift),
                     in_scale=cu(scale), in_act=True, in_mask=cu(mask), residual=cu(res), out_scale=0.5)
    assert_close(got.cpu(), want, 1e-5, 1e-5, "fused conv")
    # padding must stay exactly zero under the prologue: an all-zero input with a shift still sees zero halo
    z = torch.zeros(1, 4, 4, 8, 8)
    sh = torch.full((1, 4), 0.7)
    w2 = randn(8, 4, 4, 3, 3, 3)
    want = F.conv3d(U.silu(z + 0.7), w2, None, 1, 1)
    got = ops.conv3d([cu(z)], ops.pack_conv_weight(cu(w2)), 4, 3, in_shift=cu(sh), in_act=True)
    assert_close(got.cpu(), want, 1e-5, 1e-5, "zero padding after the prologue")
    # bank-style strided / broadcast modulation rows
    bank = randn(9, b, 100)
    got = ops.conv3d([cu(xin)], wp, co, 3, in_scale=cu(bank).data_ptr() + 4 * 10, scale_stride=100)
    want = F.conv3d(xin * bank[:, 10:42, None, None, None], wt, None, 1, 1)
    assert_close(got.cpu(), want, 1e-5, 1e-5, "strided scale rows")
    got = ops.conv3d([cu(xin)], wp, co, 3, in_scale=cu(bank[:1, 10:42]), scale_stride=-1)
    want = F.conv3d(xin * bank[:1, 10:42, None, None, None], wt, None, 1, 1)
    assert_close(got.cpu(), want, 1e-5, 1e-5, "broadcast scale row")


def test_conv3d_modulated_matches_reference_fixture(ops, golden):
    g = golden("modconv")
    for b in (1, 3):
        for k in (1, 3):
            x = randn(40, b, 5, 4, 6, 6)
            w = randn(41, 7, 5, k, k, k) / (5 * k ** 3) ** 0.5
            s = (1 + 0.3 * randn(42, b, 5, 1, 1))
            got = ops.conv3d([cu(x)], ops.pack_conv_weight(cu(w)), 7, k, in_scale=cu(s.reshape(b, 5)))
            assert_close(got.cpu(), g[f"b{b}k{k}_y"], 1e-5, 1e-5, "modulated conv vs reference")


def test_conv3d_rejects_bad_args(ops):
    from tmdiff_amd._lib import TmdiffError
    x = torch.zeros(1, 4, 4, 8, 8, device="cuda")
    wp = torch.zeros(4 * 4 * 27, device="cuda")
    with pytest.raises(TmdiffError):
        ops.conv3d([x], wp, 4, 5)                      # ksize 5
    with pytest.raises(TmdiffError):
        ops.conv3d([x], wp, 4, 3, groups=2)
    with pytest.raises(ValueError):
        ops.conv3d([x.cpu()], wp, 4, 3)
    y = ops.conv3d([x[:0]], wp, 4, 3)                  # empty batch is a no-op
    assert y.shape[0] == 0


# ---------------------------------------------------------------------------------------------
def test_stem_head_linear_gamma(ops, golden):
    b, n, h, w, c0 = 2, 8, 16, 16, 12
    d = case_inputs(5, b, n, h, w)
    wt, bias = randn(1, c0), randn(2, c0)
    cond = (d["PAN"].repeat(1, n, 1, 1) - d["MS"]).unsqueeze(1)
    want = U.silu(cond * wt[None, :, None, None, None] + bias[None, :, None, None, None])
    got = ops.stem(cu(wt), cu(bias), c0, pan=cu(d["PAN"]), ms=cu(d["MS"]))
    assert_close(got.cpu(), want, 1e-6, 1e-6, "stem(pan-ms)")
    got = ops.stem(cu(wt), cu(bias), c0, xin=cu(d["x_t"]))
    want = U.silu(d["x_t"].unsqueeze(1) * wt[None, :, None, None, None] + bias[None, :, None, None, None])
    assert_close(got.cpu(), want, 1e-6, 1e-6, "stem(x)")
    x5 = randn(3, b, c0, n, h, w)
    s = 1 + 0.2 * randn(4, b, c0)
    want = (U.silu(x5) * (wt[None] * s)[:, :, None, None, None]).sum(1)
    assert_close(ops.head(cu(x5), cu(wt), cu(s)).cpu(), want, 2e-6, 2e-6, "head")
    for i, o in ((32, 128), (768, 512), (128, 3000), (100, 7)):
        x, wl, bl = randn(5, 5, i), randn(6, o, i) / i ** 0.5, randn(7, o)
        assert_close(ops.linear(cu(x), cu(wl), cu(bl)).cpu(), F.linear(x, wl, bl), 2e-6, 2e-6, "linear")
        assert_close(ops.linear(cu(x), cu(wl), cu(bl), act=True).cpu(), U.silu(F.linear(x, wl, bl)), 2e-6, 2e-6)
    g = golden("gamma_embedding")
    freqs = torch.exp(-np.log(10000) * torch.arange(16, dtype=torch.float32) / 16)
    for key_t, key_e in (("t_int", "e_int"), ("t_frac", "e_frac")):
        t = torch.tensor(g[key_t]).float()
        got = ops.gamma_embedding(cu(t), cu(freqs), 32).cpu()
        assert np.abs(got.numpy() - g[key_e]).max() <= 1e-6


# ---------------------------------------------------------------------------------------------
def test_sampler_kernels(ops, golden):
    x, e, nz, ms = (randn(i, 3, 8, 16, 16) for i in range(4))
    a, bq, c1, c2, sg = 1.25, 0.75, 0.3, 0.69, 0.11
    x0 = (a * x - bq * e).clamp(-1, 1)
    want = c1 * x0 + c2 * x + nz * sg
    img = torch.empty_like(x).cuda()
    got = ops.ddpm_step(cu(x), cu(e), cu(nz), a, bq, c1, c2, sg, ms=cu(ms), img_out=img)
    assert_close(got.cpu(), want, 1e-6, 1e-6, "ddpm step")
    assert_close(img.cpu(), want + ms, 1e-6, 1e-6, "ddpm step image")
    got = ops.ddpm_step(cu(x), cu(e), None, a, bq, c1, c2, 0.0)
    assert_close(got.cpu(), c1 * x0 + c2 * x, 1e-6, 1e-6, "ddpm last step")
    want = 0.5 * x - 1.5 * e + 0.25 * nz
    assert_close(ops.axpby([cu(x), cu(e), cu(nz)], [0.5, -1.5, 0.25]).cpu(), want, 1e-6, 1e-6, "axpby3")
    assert_close(ops.axpby([cu(x), cu(e), cu(nz), cu(ms)], [0.5, -1.5, 0.25, 2.0]).cpu(), want + 2 * ms, 1e-6, 1e-6)
    al, sd = 0.8, 0.6
    eps = (x - al * e) / sd
    assert_close(ops.x0_from_model(cu(x), cu(e), al, sd).cpu(), (x - sd * eps) / al, 1e-5, 1e-5, "x0")
    assert_close(ops.add(cu(x), cu(ms)).cpu(), x + ms, 0, 0, "add")
    assert_close(ops.add(cu(x), cu(ms), -1.0).cpu(), x - ms, 0, 0, "sub")
    av = torch.tensor([0.9, 0.5, 0.1])
    want = av.view(-1, 1, 1, 1) * x + (1 - av.view(-1, 1, 1, 1) ** 2).sqrt() * e
    assert_close(ops.q_sample(cu(x), cu(e), cu(av)).cpu(), want, 1e-6, 1e-6, "q_sample")
    # dynamic thresholding against torch.quantile and the reference fixture
    g = golden("dpm_solver")
    x0 = randn(151, 3, 8, 16, 16)
    x0[0, 0, 0, :5] = torch.tensor([9.0, -7.0, 5.0, 30.0, -2.5])
    x0[1] *= 0.2
    t = cu(x0.clone())
    s = ops.abs_quantile_clamp_(t, 0.995, 1.0)
    want_s = torch.maximum(torch.quantile(x0.abs().reshape(3, -1), 0.995, dim=1), torch.ones(3))
    assert_close(s.cpu(), want_s, 1e-7, 1e-7, "quantile threshold")
    assert_close(t.cpu(), g["thresh_out"], 1e-6, 1e-6, "dynamic thresholding vs reference")
    for n, q in ((1000, 0.5), (4097, 0.995), (37, 1.0), (64, 0.0)):
        v = randn(n, 2, n)
        t = cu(v.clone())
        s = ops.abs_quantile_clamp_(t, q, 0.0)
        assert_close(s.cpu(), torch.quantile(v.abs(), q, dim=1), 1e-6, 1e-6, f"quantile n={n} q={q}")


# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def nets():
    from tmdiff_amd.Hyper_unet_general import WavBEST
    ref = U.fill_weights_(U.WavBESTRef(channels=TINY)).eval()
    hip = WavBEST(channels=TINY)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.cuda().eval()


def test_unet_tiny_vs_oracle_and_fixture(nets, golden):
    ref, hip = nets
    g = golden("unet_tiny")
    for c in (4, 8):
        d = case_inputs(100 + c, 2, c, 16)
        for prompt in ("QB", "WV3", "GF2", "WV2", "WV4"):
            t = torch.tensor([[3], [977]])
            y = hip(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), prompt).cpu()
            assert_close(y, g[f"c{c}_{prompt}_int"], 1e-4, 1e-5, f"unet vs reference c{c} {prompt}")
            with torch.no_grad():
                assert_close(y, ref(d["x_t"], t, d["PAN"], d["MS"], prompt), 1e-4, 1e-5, "unet vs oracle")
        y = hip(cu(d["x_t"]), torch.tensor([0.25, 731.4]).cuda(), cu(d["PAN"]), cu(d["MS"]), "WV3").cpu()
        assert_close(y, g[f"c{c}_WV3_frac"], 1e-4, 1e-5, "fractional t")
    d = case_inputs(120, 1, 8, 32, 16)
    y = hip(cu(d["x_t"]), torch.tensor([[500]]).cuda(), cu(d["PAN"]), cu(d["MS"]), "WV3").cpu()
    assert_close(y, g["nonsquare"], 1e-4, 1e-5, "non-square tile")
    with pytest.raises(AttributeError):
        hip(cu(d["x_t"]), torch.tensor([[500]]).cuda(), cu(d["PAN"]), cu(d["MS"]), "LANDSAT")


def test_unet_condition_cache_and_per_sample_prompts(nets):
    ref, hip = nets
    d = case_inputs(7, 2, 8, 16)
    pan, ms = cu(d["PAN"]), cu(d["MS"])
    t = torch.tensor([[10], [400]]).cuda()
    base = hip(cu(d["x_t"]), t, pan, ms, "GF2")
    hip.begin_condition_cache(pan, ms, "GF2")
    try:
        again = hip(cu(d["x_t"]), t, pan, ms, "GF2")
        other = hip(cu(d["x_t"]) * 0.5, t, pan, ms, "GF2")
    finally:
        hip.end_condition_cache()
    assert torch.equal(base, again)
    assert not torch.equal(base, other)
    mixed = hip(cu(d["x_t"]), t, pan, ms, ["GF2", "WV3"]).cpu()
    with torch.no_grad():
        w0 = ref(d["x_t"][:1], t[:1].cpu(), d["PAN"][:1], d["MS"][:1], "GF2")
        w1 = ref(d["x_t"][1:], t[1:].cpu(), d["PAN"][1:], d["MS"][1:], "WV3")
    assert_close(mixed, torch.cat([w0, w1]), 1e-4, 1e-5, "per-sample prompts")


def test_unet_full_width_vs_reference_fixture(golden):
    from tmdiff_amd.Hyper_unet_general import WavBEST
    ref = U.fill_weights_(U.WavBESTRef(channels=FULL))
    hip = WavBEST(channels=FULL)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda().eval()
    d = case_inputs(3407, 1, 8, 64)
    y = hip(cu(d["x_t"]), torch.tensor([[250]]).cuda(), cu(d["PAN"]), cu(d["MS"]), "WV3").cpu()
    m, l2 = rel_err(y, golden("unet_full")["y"])
    print(f"full-width forward vs reference: max-rel {m:.3e} rel-L2 {l2:.3e}")
    assert m <= 1e-4 and l2 <= 1e-5     # SURVEY 8(d) single-forward tolerance
    # batch consistency at the benchmark batch size: 4 copies of the tile give 4 identical outputs
    x4 = [cu(d[k].repeat(4, 1, 1, 1)) for k in ("x_t", "PAN", "MS")]
    y4 = hip(x4[0], torch.full((4, 1), 250).cuda(), x4[1], x4[2], "WV3").cpu()
    for i in range(4):
        assert torch.equal(y4[i], y4[0])
    # (B = 1 and B = 4 grids are split over the input channels by different factors -- tmdiff_conv3d_fwd_splitk_workspace_bytes --
    #  so the fp32 summation order differs between them; same-batch rows stay bit-identical, above)
    assert_close(y4[:1], y, 5e-6, 5e-6, "batch-4 vs batch-1")


def test_per_operator_abi_names(ops):
    """The one-symbol-per-operator fronts (include/tmdiff_hip.h) agree with the general entry points."""
    import ctypes as C
    from tmdiff_amd import _lib
    L, S = _lib.lib, None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    torch.manual_seed(5)
    # conv k3 / k1 fronts: same result as conv3d_fwd; ksize mismatch is TMDIFF_E_INVALID
    x, w3, w1 = cu(torch.randn(2, 8, 4, 8, 8)), cu(torch.randn(16, 8, 3, 3, 3)), cu(torch.randn(16, 8, 1, 1, 1))
    for w, k, fn, bad in ((w3, 3, L.tmdiff_conv3d_k3_fwd, L.tmdiff_conv3d_k1_fwd), (w1, 1, L.tmdiff_conv3d_k1_fwd, L.tmdiff_conv3d_k3_dgrad)):
        wp = ops.pack_conv_weight(w)
        y0, y1 = ops.conv3d([x], wp, 16, k), torch.empty(2, 16, 4, 8, 8, device="cuda")
        d = ops.make_conv_desc([x], wp, 16, k, y1)
        assert fn(C.byref(d), S) == 0
        assert torch.equal(y0, y1)
        assert bad(C.byref(d), S) == -1 and b"ksize" in L.tmdiff_last_error_string()
    # wgrad front
    g = cu(torch.randn(2, 16, 4, 8, 8))
    y = torch.empty(2, 16, 4, 8, 8, device="cuda")
    d = ops.make_conv_desc([x], ops.pack_conv_weight(w3), 16, 3, y)
    ws = torch.empty(max(1, L.tmdiff_conv3d_wgrad_workspace_bytes(C.byref(d))), dtype=torch.uint8, device="cuda")
    dw = torch.empty_like(w3)
    assert L.tmdiff_conv3d_k3_wgrad(C.byref(d), p(g), p(dw), p(ws), S) == 0
    with torch.enable_grad():
        wd = w3.cpu().double().requires_grad_()
        F.conv3d(x.cpu().double(), wd, padding=1).backward(g.cpu().double())
    assert_close(dw.cpu(), wd.grad.float(), 2e-5, 2e-5, "k3_wgrad")
    # haar fronts and their adjoints: <DWT(x), g> == <x, DWT_bwd(g)>, likewise for IDWT
    xs = cu(torch.randn(6, 8, 12))
    bands = [torch.empty(6, 4, 6, device="cuda") for _ in range(4)]
    assert L.tmdiff_haar_dwt2d_fwd(p(xs), *[p(b) for b in bands], 6, 8, 12, 0.5, 1.0, S) == 0
    ref_b = haar_dwt2d(xs.cpu())
    for b, r, s in zip(bands, ref_b, (0.5, 1, 1, 1)):
        assert_close(b.cpu(), r * s, 1e-6, 1e-6, "dwt2d_fwd")
    gb = [cu(torch.randn(6, 4, 6)) for _ in range(4)]
    dx = torch.empty_like(xs)
    assert L.tmdiff_haar_dwt2d_bwd(*[p(b) for b in gb], p(dx), 6, 8, 12, 0.5, 1.0, S) == 0
    lhs = sum((b.double() * g_.double()).sum() for b, g_ in zip(bands, gb))
    assert abs(lhs - (xs.double() * dx.double()).sum()) < 1e-4 * abs(lhs) + 1e-5
    rec = torch.empty_like(xs)
    assert L.tmdiff_haar_idwt2d_fwd(*[p(b) for b in bands], p(rec), 6, 4, 6, 2.0, S) == 0
    assert_close(rec, xs, 1e-6, 1e-6, "idwt2d_fwd inverts dwt2d_fwd")
    gi = [torch.empty(6, 4, 6, device="cuda") for _ in range(4)]
    assert L.tmdiff_haar_idwt2d_bwd(p(dx), *[p(b) for b in gi], 6, 4, 6, 2.0, S) == 0
    ref_g = haar_dwt2d(dx.cpu())
    for b, r, s in zip(gi, ref_g, (2.0, 1, 1, 1)):
        assert_close(b.cpu(), r * s, 1e-6, 1e-6, "idwt2d_bwd")
    assert L.tmdiff_haar_dwt2d_bwd(p(gb[0]), p(gb[1]), None, None, p(dx), 6, 8, 12, 0.5, 1.0, S) == -1
    # dpm axpby 2/3/4
    v = [cu(torch.randn(1000)) for _ in range(4)]
    c = [0.3, -1.7, 2.5, 0.01]
    out = torch.empty(1000, device="cuda")
    assert L.tmdiff_dpm_axpby2(p(v[0]), c[0], p(v[1]), c[1], p(out), 1000, S) == 0
    assert_close(out, c[0] * v[0] + c[1] * v[1], 1e-6, 1e-6, "axpby2")
    assert L.tmdiff_dpm_axpby3(p(v[0]), c[0], p(v[1]), c[1], p(v[2]), c[2], p(out), 1000, S) == 0
    assert_close(out, c[0] * v[0] + c[1] * v[1] + c[2] * v[2], 1e-6, 1e-6, "axpby3")
    assert L.tmdiff_dpm_axpby4(p(v[0]), c[0], p(v[1]), c[1], p(v[2]), c[2], p(v[3]), c[3], p(out), 1000, S) == 0
    assert_close(out, c[0] * v[0] + c[1] * v[1] + c[2] * v[2] + c[3] * v[3], 1e-6, 1e-6, "axpby4")


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("case", [
    # B, Cin(segments), Cout, groups, N, H, W, shift, scale, act, bias, residual
    dict(B=2, segs=[16], cout=64, g=1, N=4, H=8, W=8, shift=True, scale=True, act=True, bias=True, res=True),
    dict(B=1, segs=[8, 16, 8], cout=32, g=1, N=8, H=16, W=32, shift=True, scale=False, act=True, bias=True, res=False),
    dict(B=2, segs=[24], cout=32, g=1, N=4, H=6, W=10, shift=False, scale=True, act=False, bias=False, res=True),
    dict(B=1, segs=[16, 16, 16], cout=96, g=3, N=8, H=8, W=8, shift=False, scale=False, act=False, bias=True, res=False),
    dict(B=1, segs=[64], cout=128, g=1, N=8, H=20, W=12, shift=True, scale=True, act=True, bias=True, res=False),
    dict(B=3, segs=[32], cout=32, g=1, N=2, H=5, W=7, shift=False, scale=True, act=True, bias=False, res=False),
    # 1x1x1 (bandwidth kernel): the three channel-tile configurations, ragged position counts, segments
    dict(k=1, B=2, segs=[32], cout=128, g=1, N=4, H=9, W=7, shift=False, scale=False, act=False, bias=True, res=True),
    dict(k=1, B=1, segs=[16, 16, 16], cout=64, g=1, N=8, H=16, W=16, shift=True, scale=True, act=True, bias=True, res=False),
    dict(k=1, B=2, segs=[8, 24], cout=32, g=1, N=4, H=8, W=8, shift=False, scale=True, act=False, bias=False, res=False),
    dict(k=1, B=1, segs=[48], cout=96, g=3, N=4, H=8, W=8, shift=False, scale=False, act=False, bias=True, res=False),
])
def test_conv3d_bf16_operands_fp32_accumulate(ops, case):
    """bf16-compute conv (config 3 mode): equals an fp64 convolution of the bf16-rounded prologue output with the
    bf16-rounded weights up to fp32 accumulation error; and stays within the bf16 tolerance of the exact result."""
    torch.manual_seed(17)
    B, cin, cout, g = case["B"], sum(case["segs"]), case["cout"], case["g"]
    shp = (case["N"], case["H"], case["W"])
    segs = [torch.randn(B, c, *shp) for c in case["segs"]]
    k = case.get("k", 3)
    w = torch.randn(cout, cin // g, k, k, k) / (cin // g * k ** 3) ** 0.5
    bias = torch.randn(cout) if case["bias"] else None
    shift = torch.randn(B, cin) if case["shift"] else None
    scale = torch.rand(B, cin) + 0.5 if case["scale"] else None
    res = torch.randn(B, cout, *shp) if case["res"] else None
    x = torch.cat(segs, 1).double()
    if shift is not None:
        x = (x.float() + shift[:, :, None, None, None]).double()      # the kernel adds in fp32
    if case["act"]:
        x = x * torch.sigmoid(x)
    if scale is not None:
        x = x * scale[:, :, None, None, None].double()
    exact = F.conv3d(x, w.double(), None, padding=k // 2, groups=g)
    rounded = F.conv3d(_bf16_round(x.float()), _bf16_round(w), None, padding=k // 2, groups=g)
    for t in (exact, rounded):
        if bias is not None:
            t += 2.0 * bias.double()[None, :, None, None, None]
        if res is not None:
            t += res.double()
        t *= 0.5
    assert ops.bf16_conv_supported(cout, cin, k, g, case["segs"])
    wp = ops.pack_conv_weight_bf16(cu(w), groups=g)
    kw = dict(groups=g, math="bf16", bias=cu(bias) if bias is not None else None, bias_scale=2.0,
              in_shift=cu(shift) if shift is not None else None, in_scale=cu(scale) if scale is not None else None,
              in_act=case["act"], residual=cu(res) if res is not None else None, out_scale=0.5)
    y = ops.conv3d([cu(s) for s in segs], wp, cout, k, pack_input=False, **kw)
    y2 = ops.conv3d([cu(s) for s in segs], wp, cout, k, pack_input=True, **kw)
    assert torch.equal(y, y2), "fused and packed-input variants must agree bit for bit"
    # the prologue runs in fp32 on the GPU (fast exp), so a value may round to the neighbouring bf16: allow a few 1e-4
    assert_close(y.cpu(), rounded.float(), 2e-3, 3e-4, "bf16 conv vs bf16-rounded fp64 conv")
    assert_close(y.cpu(), exact.float(), 3e-2, 1e-2, "bf16 conv vs exact conv (bf16 tolerance)")


def test_conv3d_bf16_rejects_unsupported(ops):
    from tmdiff_amd._lib import TmdiffError
    x = cu(torch.randn(1, 8, 4, 8, 8))
    assert not ops.bf16_conv_supported(16, 8, 3) and not ops.bf16_conv_supported(32, 12, 3)
    assert not ops.bf16_conv_supported(32, 8, 1) and not ops.bf16_conv_supported(32, 16, 3, 1, [4, 12])
    assert ops.bf16_conv_supported(32, 16, 1) and not ops.bf16_conv_supported(32, 24, 1)
    with pytest.raises(ValueError):
        ops.pack_conv_weight_bf16(cu(torch.randn(32, 12, 3, 3, 3)))
    wp = ops.pack_conv_weight_bf16(cu(torch.randn(32, 8, 3, 3, 3)))
    with pytest.raises(TypeError):
        ops.conv3d([x], wp, 32, 3)                                     # bf16 packing handed to the fp32 kernel
    with pytest.raises(TmdiffError):
        ops.conv3d([x], wp, 32, 3, math="bf16", in_mask=torch.ones_like(x))
    with pytest.raises(TmdiffError):
        ops.conv3d([x], wp, 16, 3, math="bf16")                        # Cout not a multiple of 32


@pytest.mark.parametrize("case", [
    dict(B=2, segs=[8, 16, 8], cout=64, k=3, g=1, N=8, H=16, W=24, prologue=True),
    dict(B=1, segs=[32], cout=32, k=3, g=1, N=4, H=20, W=12, prologue=False),
    dict(B=3, segs=[16], cout=96, k=3, g=1, N=4, H=8, W=8, prologue=True),
    dict(B=2, segs=[16, 16, 16], cout=96, k=3, g=3, N=8, H=8, W=8, prologue=False),
    dict(B=1, segs=[64], cout=256, k=3, g=1, N=8, H=8, W=8, prologue=True),       # small-grid tile config
])
def test_conv3d_staged_equals_fused(ops, case):
    """The staged fp32 convolution (prologue pass + global_load_lds kernel) and the fused kernel run the same
    arithmetic in the same order: bit-identical outputs; and both match the fp64 reference."""
    torch.manual_seed(23)
    B, cin, cout, k, g = case["B"], sum(case["segs"]), case["cout"], case["k"], case["g"]
    shp = (case["N"], case["H"], case["W"])
    segs = [cu(torch.randn(B, c, *shp)) for c in case["segs"]]
    w = torch.randn(cout, cin // g, k, k, k) / (cin // g * k ** 3) ** 0.5
    wp = ops.pack_conv_weight(cu(w), groups=g)
    kw = dict(groups=g, bias=cu(torch.randn(cout)), residual=cu(torch.randn(B, cout, *shp)), out_scale=0.7)
    if case["prologue"]:
        kw.update(in_shift=cu(torch.randn(B, cin)), in_scale=cu(torch.rand(B, cin) + 0.5), in_act=True)
    d = ops.make_conv_desc(segs, wp, cout, k, torch.empty(B, cout, *shp, device="cuda"), **kw)
    from tmdiff_amd._lib import lib
    import ctypes as C
    assert lib.tmdiff_conv3d_fwd_staged_supported(C.byref(d)) == 1
    need = lib.tmdiff_conv3d_fwd_staged_workspace_bytes(C.byref(d))
    assert (need > 0) == (case["prologue"] or len(segs) > 1)
    y_f = ops.conv3d(segs, wp, cout, k, staged=False, **kw)
    y_s = ops.conv3d(segs, wp, cout, k, staged=True, **kw)
    assert torch.equal(y_f, y_s)
    x = torch.cat([s.cpu() for s in segs], 1).double()
    if case["prologue"]:
        x = x + kw["in_shift"].cpu().double()[:, :, None, None, None]
        x = x * torch.sigmoid(x) * kw["in_scale"].cpu().double()[:, :, None, None, None]
    ref = (F.conv3d(x, w.double(), kw["bias"].cpu().double(), padding=k // 2, groups=g) + kw["residual"].cpu().double()) * 0.7
    assert_close(y_s, ref.float(), 2e-5, 2e-6, "staged conv vs fp64")


def test_conv3d_staged_rejects_unsupported(ops):
    from tmdiff_amd._lib import lib, TmdiffError
    import ctypes as C
    x = cu(torch.randn(1, 6, 4, 8, 8))
    w = cu(torch.randn(16, 6, 3, 3, 3))
    wp = ops.pack_conv_weight(w)
    d = ops.make_conv_desc([x], wp, 16, 3, torch.empty(1, 16, 4, 8, 8, device="cuda"))
    assert lib.tmdiff_conv3d_fwd_staged_supported(C.byref(d)) == 0
    assert lib.tmdiff_conv3d_fwd_staged(C.byref(d), None, None) == -2
    y = ops.conv3d([x], wp, 16, 3)                 # falls back to the fused kernel by itself
    assert_close(y, F.conv3d(x.cpu(), w.cpu(), padding=1), 1e-5, 1e-5, "fallback")


def test_empty_batch_all_forward_variants(ops):
    """B == 0 is a no-op for the fused, staged, bf16 and 1x1x1 forward entry points (nothing read or written)."""
    w3, w1 = cu(torch.randn(32, 16, 3, 3, 3)), cu(torch.randn(32, 16, 1, 1, 1))
    x = torch.empty(0, 16, 4, 8, 8, device="cuda")
    for w, k in ((w3, 3), (w1, 1)):
        for kw in (dict(staged=False), dict(staged=True), dict(math="bf16", pack_input=True), dict(math="bf16", pack_input=False)):
            wp = ops.pack_conv_weight_bf16(w) if kw.get("math") == "bf16" else ops.pack_conv_weight(w)
            y = ops.conv3d([x], wp, 32, k, in_act=True, **kw)
            assert y.shape == (0, 32, 4, 8, 8)


@pytest.mark.parametrize("cfg", [dict(cin=32, cmid=64, N=8, H=16, W=16), dict(cin=16, cmid=32, N=4, H=12, W=20),
                                 dict(cin=64, cmid=128, N=8, H=8, W=8)])
def test_conv3d_second_output_is_the_consumers_prologue(ops, cfg):
    """y2 = act(y + shift2) * scale2 written by the producer's epilogue == the consumer applying that prologue itself:
    conv21(prologue(conv20(x))) gives the same bits either way, for the fused and the staged kernels."""
    torch.manual_seed(31)
    B, cin, cmid = 2, cfg["cin"], cfg["cmid"]
    shp = (cfg["N"], cfg["H"], cfg["W"])
    x = cu(torch.randn(B, cin, *shp))
    w20 = ops.pack_conv_weight(cu(torch.randn(cmid, cin, 3, 3, 3) / (cin * 27) ** 0.5))
    w21 = ops.pack_conv_weight(cu(torch.randn(cmid, cmid, 3, 3, 3) / (cmid * 27) ** 0.5))
    bias, sh1 = cu(torch.randn(cmid)), cu(torch.randn(B, cin))
    sh2, sc2 = cu(torch.randn(B, cmid) * 0.3), cu(torch.rand(B, cmid) + 0.5)
    for staged in (False, True):
        t1 = ops.conv3d([x], w20, cmid, 3, bias=bias, in_shift=sh1, in_act=True, staged=staged)
        want = ops.conv3d([t1], w21, cmid, 3, in_shift=sh2, in_scale=sc2, in_act=True, staged=staged)
        y, y2 = ops.conv3d([x], w20, cmid, 3, bias=bias, in_shift=sh1, in_act=True, staged=staged,
                           emit=dict(act=True, shift=sh2, scale=sc2))
        assert torch.equal(y, t1)
        only = ops.conv3d([x], w20, cmid, 3, bias=bias, in_shift=sh1, in_act=True, staged=staged,
                          emit=dict(act=True, shift=sh2, scale=sc2), keep_y=False)
        assert torch.equal(only, y2)
        for staged2 in (False, True):
            got = ops.conv3d([y2], w21, cmid, 3, staged=staged2)
            assert torch.equal(got, want), (staged, staged2)
    # no activation / no shift variants against plain torch arithmetic
    y, y2 = ops.conv3d([x], w20, cmid, 3, emit=dict(scale=sc2))
    assert_close(y2, y.cpu() * sc2.cpu()[:, :, None, None, None], 1e-6, 1e-6, "scale-only second output")


@pytest.mark.parametrize("cfg", [dict(cin=32, cmid=64, N=8, H=16, W=16), dict(cin=16, cmid=32, N=4, H=12, W=20),
                                 dict(cin=64, cmid=128, N=8, H=8, W=8), dict(cin=16, cmid=32, N=4, H=8, W=8)])
def test_conv3d_bf16_packed_second_output(ops, cfg):
    """bf16 mode: the producer's epilogue writes the consumer's prologue output as packed bf16 units (register quads
    regrouped with v_permlane32_swap); feeding that to the consumer == the consumer packing the fp32 tensor itself."""
    torch.manual_seed(37)
    B, cin, cmid = 2, cfg["cin"], cfg["cmid"]
    shp = (cfg["N"], cfg["H"], cfg["W"])
    x = cu(torch.randn(B, cin, *shp))
    w20 = ops.pack_conv_weight_bf16(cu(torch.randn(cmid, cin, 3, 3, 3) / (cin * 27) ** 0.5))
    w21 = ops.pack_conv_weight_bf16(cu(torch.randn(cmid, cmid, 3, 3, 3) / (cmid * 27) ** 0.5))
    bias, res = cu(torch.randn(cmid)), cu(torch.randn(B, cmid, *shp))
    sh2, sc2 = cu(torch.randn(B, cmid) * 0.3), cu(torch.rand(B, cmid) + 0.5)
    t1 = ops.conv3d([x], w20, cmid, 3, math="bf16", bias=bias, in_act=True)
    want = ops.conv3d([t1], w21, cmid, 3, math="bf16", in_shift=sh2, in_scale=sc2, in_act=True, residual=res)
    y, packed = ops.conv3d([x], w20, cmid, 3, math="bf16", bias=bias, in_act=True, emit=dict(act=True, shift=sh2, scale=sc2))
    assert torch.equal(y, t1) and packed.dtype == torch.int16 and packed.shape == (B, cmid // 8, shp[0] * shp[1] * shp[2], 8)
    # the packed tensor holds bf16(silu(t1 + shift) * scale), channel-octet major
    v = t1 + sh2[:, :, None, None, None]
    v = (v * torch.sigmoid(v)) * sc2[:, :, None, None, None]
    ref = v.reshape(B, cmid // 8, 8, -1).permute(0, 1, 3, 2).to(torch.bfloat16)
    got_bf = packed.view(torch.bfloat16)
    assert (got_bf.float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max()   # fast-exp vs torch: <= 1 bf16 ulp
    got = ops.conv3d([packed], w21, cmid, 3, math="bf16", residual=res, x_bf16_shape=shp)
    assert torch.equal(got, want)
    only = ops.conv3d([x], w20, cmid, 3, math="bf16", bias=bias, in_act=True, keep_y=False,
                      emit=dict(act=True, shift=sh2, scale=sc2))
    assert torch.equal(only, packed)


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, N, H, W           (output plane N x H/2 x W/2)
    (2, 8, 64, 8, 32, 32),            # 256-position tiles, dwordx4 epilogue, whole tiles
    (1, 6, 128, 3, 24, 12),           # odd chunk count, ragged tiles in n / h / w, W/2 = 6: scalar epilogue
    (3, 4, 64, 8, 16, 16),            # small grid: 128-position tiles (two bands)
    (1, 16, 64, 5, 20, 40),           # ragged n and h with the dwordx4 epilogue (W/2 = 20)
    (1, 64, 64, 8, 16, 16),           # 4 workgroups: split over the input channels (16 ranges) + reduction kernel
])
def test_conv3d_ll_is_conv_then_halved_ll_band(ops, case):
    """tmdiff_conv3d_ll_fwd: `3x3x3 convolution, then the LL band * 1/2` as ONE strided convolution on composed weights,
    against the oracle's Haar transform of the CPU convolution and against the two HIP kernels it replaces; with bias,
    residual, out_scale and the second output (the consumer's prologue)."""
    B, cin, cout, N, H, W = case
    torch.manual_seed(11 + cin)
    x, w, bias = torch.randn(B, cin, N, H, W), torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5, torch.randn(cout)
    res = torch.randn(B, cout, N, H // 2, W // 2)
    sh2, sc2 = torch.randn(B, cout) * 0.3, torch.rand(B, cout) + 0.5
    full = F.conv3d(x.double(), w.double(), bias.double(), padding=1)
    ll = 0.25 * (full[..., 0::2, 0::2] + full[..., 0::2, 1::2] + full[..., 1::2, 0::2] + full[..., 1::2, 1::2])
    ref_ll = haar_dwt2d(full.float().reshape(B, cout * N, H, W))[0].reshape(B, cout, N, H // 2, W // 2) * 0.5
    assert_close(ll.float(), ref_ll, 1e-5, 1e-6, "the test's LL formula vs the oracle transform")
    wp = ops.pack_conv_weight_ll(cu(w), 0.5)
    y = ops.conv3d_ll(cu(x), wp, cout, 0.5, bias=cu(bias))
    assert_close(y, ll.float(), 2e-5, 2e-6, "conv3d_ll")
    # the pair of kernels it replaces
    pair = ops.haar_dwt2d(ops.conv3d([cu(x)], ops.pack_conv_weight(cu(w)), cout, 3, bias=cu(bias)), want_high=False, ll_scale=0.5)[0]
    assert_close(y, pair.cpu(), 2e-5, 2e-6, "conv3d_ll vs conv3d + haar_dwt2d")
    # epilogue: residual, scale, second output = SiLU(out + shift) * scale
    want = (ll.float() + res) * 0.7071
    v = want + sh2[:, :, None, None, None]
    want2 = v * torch.sigmoid(v) * sc2[:, :, None, None, None]
    y, y2 = ops.conv3d_ll(cu(x), wp, cout, 0.5, bias=cu(bias), residual=cu(res), out_scale=0.7071,
                          emit=dict(act=True, shift=cu(sh2), scale=cu(sc2)))
    assert_close(y, want, 2e-5, 2e-6, "conv3d_ll + residual")
    assert_close(y2, want2, 2e-5, 2e-6, "conv3d_ll second output")
    only = ops.conv3d_ll(cu(x), wp, cout, 0.5, bias=cu(bias), residual=cu(res), out_scale=0.7071, keep_y=False,
                         emit=dict(act=True, shift=cu(sh2), scale=cu(sc2)))
    assert torch.equal(only, y2)
    with pytest.raises(ValueError):
        ops.conv3d_ll(cu(x[..., :-1]), wp, cout, 0.5)      # odd width


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, H, W   (8 bands; H, W of the full-resolution input)
    (2, 8, 32, 16, 32),       # whole tiles of the half-resolution output (8 x 16)
    (1, 6, 64, 24, 40),       # ragged tiles (12 x 20 output), two channel tiles
    (3, 16, 32, 16, 16),      # 8 x 8 output planes: pair mode, odd batch
    (1, 64, 32, 16, 32),      # one tile: split over the input channels (whole pairs of row-parity chunks per range)
    (4, 32, 64, 32, 32),
    (2, 32, 32, 32, 32, 4),   # 4-band tensors (GF-2 / QuickBird): one band tile, 16 x 16 positions
    (1, 6, 32, 24, 40, 4),    # ... ragged
])
def test_conv3d_ll_with_winograd_along_the_bands(ops, case):
    """tmdiff_conv3d_wfll_fwd: Conv_0 + halved LL band as one convolution with F(4,3) along the bands, on the space-to-depth
    second output of its producer (tmdiff_conv3d_wf_fwd with y2_s2d): the producer's s2d output is its plain second output
    rearranged (bit for bit), the composed convolution agrees with the CPU convolution + LL band (fp64), with conv3d_ll and
    with the convolution + DWT pair; bias, residual, scale and second output."""
    B, cin, cout, H, W = case[:5]
    N = case[5] if len(case) > 5 else 8
    torch.manual_seed(5 + cin + W)
    # ---- the producer: a 3x3x3 convolution whose second output feeds the down block
    xin = torch.randn(B, 2, N, H, W)                   # (one chunk of input channels: a small grid cannot split it)
    wprod = torch.randn(cin, 2, 3, 3, 3) / (2 * 27) ** 0.5
    sh2, sc2 = torch.randn(B, cin) * 0.3, torch.rand(B, cin) + 0.5
    wpp = ops.pack_conv_weight_wino(cu(wprod), groups=1, mode=2, planes=6) if cin % 32 == 0 else None
    if wpp is not None:
        em = dict(act=True, shift=cu(sh2), scale=cu(sc2))
        y_a, plain = ops.conv3d_wf([cu(xin)], wpp, cin, emit=em)
        y_b, s2d = ops.conv3d_wf([cu(xin)], wpp, cin, emit=dict(em, s2d=True))
        assert torch.equal(y_a, y_b)
        want = plain.view(B, cin, N, H // 2, 2, W // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(B, 4 * cin, N, H // 2, W // 2)
        assert torch.equal(s2d, want), "space-to-depth second output != the plain one rearranged"
        x = plain.cpu()
    else:
        x = torch.randn(B, cin, N, H, W)
    xs = cu(x).view(B, cin, N, H // 2, 2, W // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(B, 4 * cin, N, H // 2, W // 2).contiguous()
    # ---- the composed convolution
    w, bias = torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5, torch.randn(cout)
    res = torch.randn(B, cout, N, H // 2, W // 2)
    t2, c2 = torch.randn(B, cout) * 0.3, torch.rand(B, cout) + 0.5
    full = F.conv3d(x.double(), w.double(), bias.double(), padding=1)
    ll = (0.25 * (full[..., 0::2, 0::2] + full[..., 0::2, 1::2] + full[..., 1::2, 0::2] + full[..., 1::2, 1::2])).float()
    wq = ops.pack_conv_weight_wfll(cu(w), 0.5)
    before = None
    y = ops.conv3d_wf_ll(xs, wq, cout, 0.5, bias=cu(bias))
    assert_close(y, ll, 2e-5, 2e-6, "conv3d_wf_ll")
    if cout % 64 == 0 and cin % 2 == 0:
        old = ops.conv3d_ll(cu(x), ops.pack_conv_weight_ll(cu(w), 0.5), cout, 0.5, bias=cu(bias))
        assert_close(y, old.cpu(), 1e-5, 2e-6, "conv3d_wf_ll vs conv3d_ll")
    want = (ll + res) * 0.7071
    v = want + t2[:, :, None, None, None]
    want2 = v * torch.sigmoid(v) * c2[:, :, None, None, None]
    y, y2 = ops.conv3d_wf_ll(xs, wq, cout, 0.5, bias=cu(bias), residual=cu(res), out_scale=0.7071,
                             emit=dict(act=True, shift=cu(t2), scale=cu(c2)))
    assert_close(y, want, 2e-5, 2e-6, "conv3d_wf_ll + residual")
    assert_close(y2, want2, 2e-5, 2e-6, "conv3d_wf_ll second output")
    only = ops.conv3d_wf_ll(xs, wq, cout, 0.5, bias=cu(bias), residual=cu(res), out_scale=0.7071, keep_y=False,
                            emit=dict(act=True, shift=cu(t2), scale=cu(c2)))
    assert torch.equal(only, y2)


@pytest.mark.parametrize("shape", [(3, 8, 8), (2, 12, 20), (4, 16, 8)])
def test_space_to_depth_second_output_in_every_tile_mode(ops, shape):
    """desc.y2_s2d of tmdiff_conv3d_wf_fwd: pair mode (8-column planes, odd batch), ragged tiles -- always the plain second
    output rearranged, bit for bit, and the first output untouched."""
    B, H, W = shape
    torch.manual_seed(B + H + W)
    x, w = torch.randn(B, 2, 8, H, W), torch.randn(32, 2, 3, 3, 3) / 54 ** 0.5
    res = torch.randn(B, 32, 8, H, W)
    sh2, sc2 = torch.randn(B, 32) * 0.3, torch.rand(B, 32) + 0.5
    wp = ops.pack_conv_weight_wino(cu(w), groups=1, mode=2, planes=6)
    em = dict(act=True, shift=cu(sh2), scale=cu(sc2))
    y_a, plain = ops.conv3d_wf([cu(x)], wp, 32, emit=em, residual=cu(res), out_scale=0.5)
    y_b, s2d = ops.conv3d_wf([cu(x)], wp, 32, emit=dict(em, s2d=True), residual=cu(res), out_scale=0.5)
    assert torch.equal(y_a, y_b)
    want = plain.view(B, 32, 8, H // 2, 2, W // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(B, 128, 8, H // 2, W // 2)
    assert torch.equal(s2d, want)
    only = ops.conv3d_wf([cu(x)], wp, 32, emit=dict(em, s2d=True), residual=cu(res), out_scale=0.5, keep_y=False)
    assert torch.equal(only, s2d)


@pytest.mark.parametrize("case", [
    # B, segs, Cout, N, H, W, groups
    (2, (8,), 64, 8, 16, 16, 1),          # 64-channel tiles, whole tiles
    (1, (6,), 32, 4, 12, 20, 1),          # 32-channel tiles, ragged in t / h / w, odd chunk count
    (1, (8, 8, 8), 96, 6, 8, 12, 1),      # three segments (concat-free), Cout = 96, three pairs (ragged in t)
    (2, (4, 4, 4), 192, 4, 8, 8, 3),      # groups = 3 (convH_0): one segment per group, 64 output channels each
    # F(4,3) (N % 4 == 0, at least two tiles along the bands):
    (1, (8,), 32, 8, 16, 32, 1),          # the 32-channel tile (8 x 16 positions), whole tiles
    (1, (6,), 32, 8, 12, 20, 1),          # 32-channel tile, ragged in h and w, odd chunk count
    (1, (6,), 64, 8, 12, 20, 1),          # 64-channel tile, ragged in h and w
    (1, (8, 8, 8), 64, 8, 8, 12, 1),      # three segments (the up path's conv20)
    (2, (4, 4, 4), 192, 8, 8, 8, 3),      # groups = 3 at N = 8 (convH_0 of a WV3 batch)
    (1, (4,), 64, 12, 8, 8, 1),           # N = 12: three tiles, the second tile of the last workgroup is empty
    (1, (4,), 32, 16, 8, 16, 1),          # N = 16: two workgroups along the bands
])
@pytest.mark.fallback
def test_conv3d_winograd_along_bands(ops, case):
    """tmdiff_conv3d_wino_fwd: Winograd F(4,3) (N % 4 == 0) / F(2,3) along the band axis (input transform pass with the
    prologue + 54- / 36-tap kernel + output transform at the end of the tile) against the CPU convolution (fp64) and the direct HIP kernel; prologue, bias, residual,
    scale and second output included."""
    B, segc, cout, N, H, W, groups = case
    cin = sum(segc)
    torch.manual_seed(31 + cout)
    segs = [torch.randn(B, c, N, H, W) for c in segc]
    x = torch.cat(segs, 1)
    w, bias = torch.randn(cout, cin // groups, 3, 3, 3) / (cin // groups * 27) ** 0.5, torch.randn(cout)
    sh, sc = torch.randn(B, cin) * 0.3, torch.rand(B, cin) + 0.5
    res = torch.randn(B, cout, N, H, W)
    sh2, sc2 = torch.randn(B, cout) * 0.3, torch.rand(B, cout) + 0.5
    from tmdiff_amd import fallback, routing
    wp = ops.pack_conv_weight_wino(cu(w), groups=groups, planes=fallback.wino_planes(N))     # F(4,3) for N % 4 == 0, else F(2,3)
    want = F.conv3d(x.double(), w.double(), bias.double(), padding=1, groups=groups).float()
    y = fallback.conv3d_wino([cu(s_) for s_ in segs], wp, cout, bias=cu(bias), groups=groups)
    assert_close(y, want, 2e-5, 2e-6, "winograd, plain input")
    xs = x.double() + sh[:, :, None, None, None].double()
    xs = xs * torch.sigmoid(xs) * sc[:, :, None, None, None].double()
    want = ((F.conv3d(xs, w.double(), bias.double(), padding=1, groups=groups) + res.double()) * 0.7071).float()
    v = want + sh2[:, :, None, None, None]
    want2 = v * torch.sigmoid(v) * sc2[:, :, None, None, None]
    y, y2 = fallback.conv3d_wino([cu(s_) for s_ in segs], wp, cout, bias=cu(bias), in_shift=cu(sh), in_scale=cu(sc), in_act=True,
                            residual=cu(res), out_scale=0.7071, emit=dict(act=True, shift=cu(sh2), scale=cu(sc2)), groups=groups)
    assert_close(y, want, 2e-5, 2e-6, "winograd, prologue + residual")
    assert_close(y2, want2, 2e-5, 2e-6, "winograd, second output")
    direct = ops.conv3d([cu(s_) for s_ in segs], ops.pack_conv_weight(cu(w), groups=groups), cout, 3, groups=groups, bias=cu(bias),
                        in_shift=cu(sh), in_scale=cu(sc), in_act=True, residual=cu(res), out_scale=0.7071)
    assert_close(y, direct.cpu(), 1e-5, 1e-6, "winograd vs the direct kernel")
    # in-kernel dropout of the prologue output (finetune path): the same keep mask as the direct kernels, and x' kept
    xp = torch.empty(B, cin, N, H, W, device="cuda")
    yd = fallback.conv3d_wino([cu(s_) for s_ in segs], wp, cout, in_shift=cu(sh), in_act=True, drop=(1234, 0.2), groups=groups, xp_out=xp)
    staged = (cin // groups) % 4 == 0            # (the direct path keeps x' only on its staged kernel)
    xp_d = torch.empty_like(xp) if staged else None
    dd = ops.conv3d([cu(s_) for s_ in segs], ops.pack_conv_weight(cu(w), groups=groups), cout, 3, groups=groups, in_shift=cu(sh),
                    in_act=True, drop=(1234, 0.2), xp_out=xp_d)
    assert 0.1 < float((xp == 0).float().mean()) < 0.3 and (not staged or torch.equal(xp, xp_d))
    assert_close(yd, dd.cpu(), 1e-5, 1e-6, "winograd vs the direct kernel, dropout")
    # a grid too small for the kernel (no split-K) or an odd band count is routed to the direct kernels; a large grid of a
    # band count conv3d_wf does not take (not 4 / 8) is what reaches this module
    assert routing.conv3_family(B, cin, cout, N - 1, H, W, groups) in ("staged", "fused")
    assert routing.conv3_family(B, cin, cout, N, H, W, groups) in ("staged", "fused", "wf", "wf_pair")     # (small test grids)
    with ops.config.override(wino_min_blocks=1):
        fam = routing.conv3_family(B, cin, cout, N, H, W, groups)
        if N not in (4, 8):
            assert fam == ("wino4" if fallback.wino_planes(N) == 6 else "wino2"), fam
        else:         # (4 / 8 bands: conv3d_wf wherever its tiles are filled, this module only for the narrow planes it leaves)
            assert fam in ("wf", "wf_pair", "wino4", "wino2"), fam
        if N not in (4, 8):        # ... and ops.conv3d_auto dispatches here, with the same bits
            weights = ops.ConvWeights(lambda: ops.pack_conv_weight(cu(w), groups=groups), None,
                                      lambda planes: ops.pack_conv_weight_wino(cu(w), groups=groups, planes=planes))
            auto = ops.conv3d_auto([cu(s_) for s_ in segs], weights, cout, groups=groups, bias=cu(bias))
            assert torch.equal(auto, fallback.conv3d_wino([cu(s_) for s_ in segs], wp, cout, bias=cu(bias), groups=groups))


@pytest.mark.parametrize("case", [
    # B, segs, Cout, N, H, W, groups
    (2, (8,), 64, 8, 16, 16, 1),          # whole tiles (8 x 16 positions x 2 band tiles), two channel tiles
    (1, (6,), 32, 8, 12, 20, 1),          # ragged in h and w (W = 20: the second tile has one quad), odd chunk count
    (1, (2,), 32, 8, 8, 16, 1),           # a single chunk
    (1, (8, 8, 8), 96, 8, 8, 12, 1),      # three segments (prologue pass = the concatenation), three channel tiles
    (2, (4, 4, 4), 192, 8, 8, 16, 3),     # groups = 3 (convH_0)
    (2, (8,), 64, 4, 32, 32, 1),          # N = 4 (GF-2 / QuickBird): one band tile, 16 x 16 positions
    (1, (6,), 32, 4, 20, 24, 1),          # N = 4, ragged
    (1, (4, 4, 4), 96, 4, 16, 16, 3),     # N = 4, groups = 3
    (3, (64,), 64, 8, 32, 32, 1),         # 32 chunks, several workgroups per image
    (1, (64,), 64, 8, 16, 16, 1),         # 4 tiles: split over the input channels (16 ranges of 2 chunks) + reduction kernel
    (2, (48,), 32, 4, 16, 32, 1),         # N = 4, split-K (12 ranges)
    (1, (8, 8, 8), 96, 8, 8, 16, 3),      # groups = 3 with split-K (2 ranges of 2 chunks)
    (4, (8,), 64, 8, 8, 8, 1),            # 8-column planes: two images side by side in one tile (pair mode)
    (3, (6,), 32, 8, 12, 8, 1),           # pair mode, odd batch (the last tile holds one image), ragged in h
    (1, (4,), 32, 8, 8, 8, 1),            # pair mode, a single image
    (3, (8, 8, 8), 192, 8, 8, 8, 3),      # pair mode, groups = 3 on three segments (read in place), split-K
    (5, (64,), 64, 8, 16, 8, 1),          # pair mode, split-K, odd batch
    (8, (8,), 64, 8, 64, 64, 1),          # a grid of two full rounds (512 tiles x 2 channel tiles), several images
    (5, (6,), 128, 8, 40, 72, 1),         # ragged in h and w on a large grid, four channel tiles, odd chunk count
])
def test_conv3d_winograd_in_kernel_transform(ops, case):
    """tmdiff_conv3d_wf_fwd: Winograd F(4,3) along the bands with the input transform inside the kernel (no transformed
    copy of the input): plain input, prologue / segments (through the prologue pass), bias, residual, scale and second
    output, against the CPU convolution (fp64), the direct HIP kernel and the transform-pass Winograd kernel; the
    data-gradient weights against CPU autograd."""
    B, segc, cout, N, H, W, groups = case
    cin = sum(segc)
    torch.manual_seed(77 + cout + N)
    segs = [torch.randn(B, c, N, H, W) for c in segc]
    x = torch.cat(segs, 1)
    w, bias = torch.randn(cout, cin // groups, 3, 3, 3) / (cin // groups * 27) ** 0.5, torch.randn(cout)
    sh, sc = torch.randn(B, cin) * 0.3, torch.rand(B, cin) + 0.5
    res = torch.randn(B, cout, N, H, W)
    sh2, sc2 = torch.randn(B, cout) * 0.3, torch.rand(B, cout) + 0.5
    wp = ops.pack_conv_weight_wino(cu(w), groups=groups, mode=2, planes=6)
    want = F.conv3d(x.double(), w.double(), bias.double(), padding=1, groups=groups).float()
    y = ops.conv3d_wf([cu(x)], wp, cout, bias=cu(bias), groups=groups)                     # one plain tensor: no pass at all
    assert_close(y, want, 2e-5, 2e-6, "wf, plain input")
    y = ops.conv3d_wf([cu(s_) for s_ in segs], wp, cout, bias=cu(bias), groups=groups)     # segments: prologue pass
    assert_close(y, want, 2e-5, 2e-6, "wf, segments")
    xs = x.double() + sh[:, :, None, None, None].double()
    xs = xs * torch.sigmoid(xs) * sc[:, :, None, None, None].double()
    want = ((F.conv3d(xs, w.double(), bias.double(), padding=1, groups=groups) + res.double()) * 0.7071).float()
    v = want + sh2[:, :, None, None, None]
    want2 = v * torch.sigmoid(v) * sc2[:, :, None, None, None]
    kw = dict(bias=cu(bias), in_shift=cu(sh), in_scale=cu(sc), in_act=True, residual=cu(res), out_scale=0.7071, groups=groups)
    y, y2 = ops.conv3d_wf([cu(s_) for s_ in segs], wp, cout, emit=dict(act=True, shift=cu(sh2), scale=cu(sc2)), **kw)
    assert_close(y, want, 2e-5, 2e-6, "wf, prologue + residual")
    assert_close(y2, want2, 2e-5, 2e-6, "wf, second output")
    only = ops.conv3d_wf([cu(s_) for s_ in segs], wp, cout, emit=dict(act=True, shift=cu(sh2), scale=cu(sc2)), keep_y=False, **kw)
    assert torch.equal(only, y2)
    direct = ops.conv3d([cu(s_) for s_ in segs], ops.pack_conv_weight(cu(w), groups=groups), cout, 3, **kw)
    assert_close(y, direct.cpu(), 1e-5, 2e-6, "wf vs the direct kernel")
    from tmdiff_amd import fallback, routing
    old = fallback.conv3d_wino([cu(s_) for s_ in segs], ops.pack_conv_weight_wino(cu(w), groups=groups, planes=fallback.wino_planes(N)), cout, **kw)
    assert_close(y, old.cpu(), 1e-5, 2e-6, "wf vs the transform-pass Winograd kernel")
    # in-kernel dropout of the prologue output (finetune path) and the kept x'
    xp = torch.empty(B, cin, N, H, W, device="cuda")
    yd = ops.conv3d_wf([cu(s_) for s_ in segs], wp, cout, in_shift=cu(sh), in_act=True, drop=(1234, 0.2), groups=groups, xp_out=xp)
    dd = ops.conv3d([cu(s_) for s_ in segs], ops.pack_conv_weight(cu(w), groups=groups), cout, 3, groups=groups, in_shift=cu(sh),
                    in_act=True, drop=(1234, 0.2))
    assert 0.1 < float((xp == 0).float().mean()) < 0.3
    assert_close(yd, dd.cpu(), 1e-5, 2e-6, "wf vs the direct kernel, dropout")
    # data-gradient weights (mode bit 0) in natural column order
    g = torch.randn(B, cout, N, H, W)
    with torch.enable_grad():
        xg = x.double().requires_grad_(True)
        F.conv3d(xg, w.double(), None, padding=1, groups=groups).backward(g.double())
    if (cout // groups) % 2 == 0 and (cin // groups) % 32 == 0:
        dx = ops.conv3d_wf([cu(g)], ops.pack_conv_weight_wino(cu(w), groups=groups, mode=3, planes=6), cin, groups=groups)
        assert_close(dx, xg.grad.float(), 2e-5, 2e-6, "wf data gradient vs CPU autograd")
    # an odd band count / a mask tensor is never routed here (and the kernel itself refuses the shape)
    assert routing.conv3_family(B, cin, cout, N - 1, H, W, groups) in ("staged", "fused")
    assert routing.conv3_family(B, cin, cout, N, H, W, groups, masked=True) in ("staged", "fused")
    with pytest.raises(ValueError):
        ops.conv3d_wf([cu(x[:, :, :N - 1])], wp, cout, groups=groups)


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, N, H, W, groups   (of the FORWARD convolution whose data gradient is taken)
    (2, 32, 64, 8, 16, 16, 1),            # F(4,3), 32-channel tile of the gradient convolution (Cin = 32 outputs)
    (1, 64, 32, 8, 12, 20, 1),            # F(4,3), 64-channel tile, ragged
    (1, 64, 64, 4, 8, 16, 1),             # F(2,3)
    (1, 96, 192, 8, 8, 8, 3),             # groups = 3
])
@pytest.mark.fallback
def test_conv3d_winograd_data_gradient_weights(ops, case):
    """mode-1 packing of tmdiff_conv3d_wino_pack_weights: the Winograd weights of the DATA-GRADIENT convolution (transposed,
    taps mirrored) straight from the forward weight -- against CPU autograd's grad_input (fp64) and the direct kernel on
    the direct data-gradient packing."""
    B, cin, cout, N, H, W, groups = case
    torch.manual_seed(5 + cin + cout)
    x = torch.randn(B, cin, N, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin // groups, 3, 3, 3) / (cin // groups * 27) ** 0.5
    g = torch.randn(B, cout, N, H, W)
    with torch.enable_grad():             # (the module's fixture runs every test under no_grad)
        F.conv3d(x, w.double(), None, padding=1, groups=groups).backward(g.double())
    want = x.grad.float()
    from tmdiff_amd import fallback
    wp1 = ops.pack_conv_weight_wino(cu(w), groups=groups, mode=1, planes=fallback.wino_planes(N))
    dx = fallback.conv3d_wino([cu(g)], wp1, cin, groups=groups)
    assert_close(dx, want, 2e-5, 2e-6, "winograd data gradient vs CPU autograd")
    direct = ops.conv3d([cu(g)], ops.pack_conv_weight(cu(w), groups=groups, mode=1), cin, 3, groups=groups)
    assert_close(dx, direct.cpu(), 1e-5, 2e-6, "winograd data gradient vs the direct kernel")   # (F(4,3): 1e-6 vs fp64 by itself)


@pytest.mark.parametrize("case", [
    # B, Cx (res_conv input channels), Cin = Cout of conv21, N, H, W
    (2, 32, 64, 8, 16, 32),           # the level-0 shape class (32 -> 64): two channel tiles, whole tiles
    (3, 64, 128, 8, 12, 20),          # ragged in h and w, four channel tiles, two 32-channel groups of x
    (2, 32, 64, 4, 32, 32),           # 4 bands (GF-2 / QB): the 16 x 16 tile's position mapping
    (1, 128, 32, 8, 8, 16),           # more input channels than output channels, a single tile
])
def test_conv3d_wf_folds_the_residual_convolution(ops, case):
    """desc.rc_* (ABI v6): a ResBlock's 1x1x1 res_conv (reference Hyper_unet_general.py:231, :248) folded into conv21's epilogue
    -- W1^T x accumulated by the matrix pipe into the output blocks -- against the CPU convolutions (fp64) and against the
    two-launch form it replaces (1x1x1 kernel, then conv3d_wf with its result as residual); second output and scale included."""
    B, cx, c, N, H, W = case
    torch.manual_seed(500 + cx + c + N)
    xr, t1 = torch.randn(B, cx, N, H, W), torch.randn(B, c, N, H, W)
    w21, w1, b1 = torch.randn(c, c, 3, 3, 3) / (c * 27) ** 0.5, torch.randn(c, cx, 1, 1, 1) / cx ** 0.5, torch.randn(c)
    sh2, sc2 = torch.randn(B, c) * 0.3, torch.rand(B, c) + 0.5
    want = ((F.conv3d(t1.double(), w21.double(), None, padding=1) + F.conv3d(xr.double(), w1.double(), b1.double())) * 0.7071).float()
    v = want + sh2[:, :, None, None, None]
    want2 = v * torch.sigmoid(v) * sc2[:, :, None, None, None]
    wp, w1p, w1d = ops.pack_conv_weight_wino(cu(w21), mode=2, planes=6), ops.pack_conv_weight(cu(w1)), cu(w1)
    em = dict(act=True, shift=cu(sh2), scale=cu(sc2))
    y, y2 = ops.conv3d_wf([cu(t1)], wp, c, bias=cu(b1), res_conv=(cu(xr), w1d, cx), out_scale=0.7071, emit=em)
    assert_close(y, want, 2e-5, 2e-6, "folded res_conv")
    assert_close(y2, want2, 2e-5, 2e-6, "folded res_conv, second output")
    res = ops.conv3d([cu(xr)], w1p, c, 1, bias=cu(b1))
    y_two, y2_two = ops.conv3d_wf([cu(t1)], wp, c, residual=res, out_scale=0.7071, emit=em)
    assert_close(y, y_two.cpu(), 5e-6, 2e-6, "folded vs two launches")           # (same products, another summation order)
    assert_close(y2, y2_two.cpu(), 5e-6, 2e-6, "folded vs two launches, second output")
    only = ops.conv3d_wf([cu(t1)], wp, c, bias=cu(b1), res_conv=(cu(xr), w1d, cx), out_scale=0.7071, emit=em, keep_y=False)
    assert torch.equal(only, y2)
    # what the kernel does not take is refused, not computed wrongly: a residual tensor beside it, the direct kernels
    with pytest.raises(ValueError):
        ops.conv3d_wf([cu(t1)], wp, c, residual=res, res_conv=(cu(xr), w1d, cx))
    with pytest.raises(ValueError):
        ops.conv3d([cu(t1)], ops.pack_conv_weight(cu(w21)), c, 3, res_conv=(cu(xr), w1d, cx))


@pytest.mark.parametrize("case", [(2, 32, 64, 16, 32), (3, 8, 32, 12, 20), (1, 64, 64, 24, 16)])
def test_conv3d_wf_writes_the_ll_band_instead_of_y(ops, case, request):
    """desc.y_ll (ABI v6): the epilogue writes LL(y) / 2 -- (a + b + c + d) / 4 over every 2 x 2 pixel block -- instead of y, beside the
    second output (plain or space-to-depth): what a down block reads of its ResBlock's raw output (reference
    Hyper_unet_general.py:374, :390, :396).  Against the Haar kernel on the y of an ordinary launch, residual / folded
    res_conv / scale included; the second output is bit-identical to the ordinary launch's."""
    B, cin, c, H, W = case
    torch.manual_seed(900 + cin + c + H)
    x, res = torch.randn(B, cin, 8, H, W, device="cuda"), torch.randn(B, c, 8, H, W, device="cuda")
    w, bias = torch.randn(c, cin, 3, 3, 3, device="cuda") / (cin * 27) ** 0.5, torch.randn(c, device="cuda")
    sh2, sc2 = torch.randn(B, c, device="cuda") * 0.3, torch.rand(B, c, device="cuda") + 0.5
    wp = ops.pack_conv_weight_wino(w, mode=2, planes=6)
    em = dict(act=True, shift=sh2, scale=sc2)
    ctx = ops.config.override(wf_splitk=False)       # (these small grids would split their input channels: such launches write
    ctx.__enter__()                                   #  neither form -- in the network routing.wf_route is asked first)
    request.addfinalizer(lambda: ctx.__exit__(None, None, None))
    for s2d in (False, True):
        y, y2 = ops.conv3d_wf([x], wp, c, bias=bias, residual=res, out_scale=0.7071, emit=dict(em, s2d=s2d))
        y2b, yll = ops.conv3d_wf([x], wp, c, bias=bias, residual=res, out_scale=0.7071, emit=dict(em, s2d=s2d, ll=True), keep_y=False)
        assert torch.equal(y2, y2b)
        want = ops.haar_dwt2d(y, want_high=False, ll_scale=0.5)[0]
        assert_close(yll, want.cpu(), 1e-6, 5e-7, "LL output vs the Haar kernel on y")
        a_ = y[..., 0::2, 0::2] + y[..., 0::2, 1::2] + y[..., 1::2, 0::2] + y[..., 1::2, 1::2]
        assert_close(yll, (a_ * 0.25).cpu(), 1e-6, 5e-7, "LL output vs the definition")
    if cin % 32 == 0:      # with a folded res_conv on top (the in-network combination of the 32 -> 64 ResBlocks)
        xr, w1 = torch.randn(B, 32, 8, H, W, device="cuda"), torch.randn(c, 32, 1, 1, 1, device="cuda") / 32 ** 0.5
        y, y2 = ops.conv3d_wf([x], wp, c, bias=bias, res_conv=(xr, w1, 32), emit=em)
        y2b, yll = ops.conv3d_wf([x], wp, c, bias=bias, res_conv=(xr, w1, 32), emit=dict(em, ll=True), keep_y=False)
        assert torch.equal(y2, y2b)
        assert_close(yll, ops.haar_dwt2d(y, want_high=False, ll_scale=0.5)[0].cpu(), 1e-6, 5e-7, "LL output with a folded res_conv")
    with pytest.raises(ValueError):
        ops.conv3d_wf([x], wp, c, emit=dict(em, ll=True))                    # (the LL output replaces y: keep_y=False)
    # desc.y_hi: the WHOLE Haar transform of y instead of y -- LL / 2 through the emit prologue, LH, HL, HH -- against the Haar kernel
    # (with its LL prologue) on the y of an ordinary launch: what a down block that keeps its high bands makes of Conv_0's output
    y = ops.conv3d_wf([x], wp, c, bias=bias)
    got = ops.conv3d_wf([x], wp, c, bias=bias, emit=dict(em, dwt=True), keep_y=False)
    want = ops.haar_dwt2d(y, want_high=True, ll_scale=0.5, ll_prologue=dict(act=True, shift=sh2, scale=sc2))
    for nm, g_, w_ in zip(("LL'", "LH", "HL", "HH"), got, want):
        assert_close(g_, w_.cpu(), 2e-6, 1e-6, f"Haar output {nm} vs the Haar kernel on y")


@pytest.mark.parametrize("case", [(2, (32,), (8, 16, 16)), (3, (8, 16, 8), (4, 10, 7)), (8, (32,), (8, 64, 64))])
def test_prologue_backward_adds_another_gradient(ops, case):
    """tmdiff_conv3d_prologue_bwd_add: dx_i = add_i + dL/dx_i in one pass (the identity residual's gradient joins conv20's input
    gradient inside the kernel: tmdiff_amd.autograd._ResBlockId) -- bit for bit the separate sum, the added tensor untouched; the
    in-place form (accumulate) gives the same."""
    B, seg_c, shp = case
    cin = sum(seg_c)
    torch.manual_seed(cin + B)
    xs = [torch.randn(B, c, *shp, device="cuda") for c in seg_c]
    gp = torch.randn(B, cin, *shp, device="cuda")
    sh, sc = torch.randn(B, cin, device="cuda") * 0.3, torch.rand(B, cin, device="cuda") + 0.5
    adds = [torch.randn_like(x) for x in xs]
    keep = [a.clone() for a in adds]
    d = ops.make_conv_desc(xs, 0, 32, 3, torch.empty(B, 32, *shp, device="cuda"), in_shift=sh, in_scale=sc, in_act=True, drop=(77, 0.2))
    plain = [torch.empty_like(x) for x in xs]
    dsh0, dsc0 = ops.conv3d_prologue_bwd(d, gp, plain, [False] * len(xs), True, True)
    out = [torch.full_like(x, float("nan")) for x in xs]
    dsh1, dsc1 = ops.conv3d_prologue_bwd(d, gp, out, [False] * len(xs), True, True, add_segs=adds)
    assert torch.equal(dsh0, dsh1) and torch.equal(dsc0, dsc1)
    for o, p, a, k in zip(out, plain, adds, keep):
        assert torch.equal(o, a + p) and torch.equal(a, k)
    acc = [a.clone() for a in adds]
    ops.conv3d_prologue_bwd(d, gp, acc, [True] * len(xs), False, False)
    for o, c in zip(out, acc):
        assert torch.equal(o, c)
    with pytest.raises(ValueError):
        ops.conv3d_prologue_bwd(d, gp, acc, [True] * len(xs), False, False, add_segs=adds)


def test_dropout_seed_word_in_device_memory(ops):
    """tmdiff_conv3d_desc.drop_seed_dev (ABI v6): the in-kernel dropout seed is drop_seed + *drop_seed_dev, read when the
    kernel starts -- a launch recorded into a HIP graph draws a fresh mask on every replay once the word is bumped.  The
    prologue pass (conv3d_wf / staged), the fused direct kernel, the weight gradient's own prologue and the prologue
    backward all honour it: with word w and seed s they reproduce seed s + w without a word, bit for bit."""
    from tmdiff_amd import autograd as A
    torch.manual_seed(3)
    x = torch.randn(2, 32, 8, 16, 16, device="cuda")
    w = torch.randn(32, 32, 3, 3, 3, device="cuda") / 30
    sh = torch.randn(2, 32, device="cuda") * 0.3
    wd, wf = ops.pack_conv_weight(w), ops.pack_conv_weight_wino(w, mode=2, planes=6)

    def run(seed, word):
        ops.DROP_WORD = None if word is None else torch.tensor([word], dtype=torch.int64, device="cuda")
        try:
            xp = torch.empty_like(x)
            y_wf = ops.conv3d_wf([x], wf, 32, in_shift=sh, in_act=True, drop=(seed, 0.2), xp_out=xp)
            y_fused = ops.conv3d([x], wd, 32, 3, in_shift=sh, in_act=True, drop=(seed, 0.2), staged=False)
            g = torch.ones_like(y_wf)
            d = ops.make_conv_desc([x], 0, 32, 3, g, in_shift=sh, in_act=True, drop=(seed, 0.2))
            dw = ops.conv3d_wgrad(d, g, tuple(w.shape))
            dx = torch.empty_like(x)
            dsh, _ = ops.conv3d_prologue_bwd(d, g, [dx], [False], True, False)
            return xp, y_wf, y_fused, dw, dx, dsh
        finally:
            ops.DROP_WORD = None

    base = run(1000, None)
    assert 0.1 < float((base[0] == 0).float().mean()) < 0.3
    moved = run(1000, 7)
    assert not torch.equal(base[0], moved[0])                       # another word, another mask
    for a, b in zip(run(1007, None), moved):                        # ... namely the mask of seed + word
        assert torch.equal(a, b)
    for a, b in zip(run(1000, 0), base):
        assert torch.equal(a, b)
    # inside a graph: the word is bumped by a recorded kernel, every replay draws the next mask
    word = torch.tensor([50], dtype=torch.int64, device="cuda")
    ops.DROP_WORD = word
    try:
        xp = torch.empty_like(x)
        ops.conv3d_wf([x], wf, 32, in_shift=sh, in_act=True, drop=(1000, 0.2), xp_out=xp)     # warm-up (workspaces)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            word.add_(1)
            ops.conv3d_wf([x], wf, 32, in_shift=sh, in_act=True, drop=(1000, 0.2), xp_out=xp)
        g.replay()
        first = xp.clone()
        g.replay()
        assert int(word) == 52 and not torch.equal(first, xp)
    finally:
        ops.DROP_WORD = None
    assert torch.equal(first, run(1051, None)[0]) and torch.equal(xp, run(1052, None)[0])


def test_conv3d_large_plane_config3_shape(ops):
    """One level-0 convolution at the config-3 plane size (8 x 256 x 256 = 524288 positions per channel): offsets, tiling
    and zero padding on a large plane, for the fused, staged and bf16 kernels, against the CPU convolution."""
    torch.manual_seed(41)
    cin, cout = 16, 32
    x = torch.randn(1, cin, 8, 256, 256)
    w = torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5
    sc = torch.rand(1, cin) + 0.5
    xs = x * torch.sigmoid(x) * sc[:, :, None, None, None]
    want = F.conv3d(xs, w, padding=1)
    wp = ops.pack_conv_weight(cu(w))
    for staged in (False, True):
        y = ops.conv3d([cu(x)], wp, cout, 3, in_scale=cu(sc), in_act=True, staged=staged)
        assert_close(y, want, 2e-5, 2e-6, f"large plane, staged={staged}")
    y16 = ops.conv3d([cu(x)], ops.pack_conv_weight_bf16(cu(w)), cout, 3, math="bf16", in_scale=cu(sc), in_act=True)
    assert_close(y16, want, 3e-2, 1e-2, "large plane, bf16")
    # borders are where the padding logic lives: compare them separately at full precision budget
    for sl in (np.s_[..., 0, :, :], np.s_[..., -1, :, :], np.s_[..., :, 0, :], np.s_[..., :, -1, :], np.s_[..., 0], np.s_[..., -1]):
        assert_close(y.cpu()[sl], want[sl], 2e-5, 2e-6, "border")


@pytest.mark.parametrize("case", [
    dict(B=1, cin=256, cout=256, N=8, H=8, W=8),          # config 1, level 3: 16 workgroups without the split
    dict(B=1, cin=96, cout=32, N=8, H=32, W=32),          # 32-channel tiles
    dict(B=2, cin=64, cout=128, N=4, H=16, W=16),
    dict(B=1, cin=24, cout=64, N=3, H=10, W=12),          # ragged box, 6 chunks
])
def test_conv3d_split_k(ops, case):
    """Small grids are split over the input channels (deterministic two-kernel reduction): same result as the unsplit
    launch to fp32 rounding, both against fp64, the second output (consumer prologue) included; repeatable bit for bit."""
    import ctypes as C
    from tmdiff_amd._lib import lib
    torch.manual_seed(31)
    B, cin, cout, shp = case["B"], case["cin"], case["cout"], (case["N"], case["H"], case["W"])
    x = cu(torch.randn(B, cin, *shp))
    w = torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5
    wp = ops.pack_conv_weight(cu(w))
    bias, res = cu(torch.randn(cout)), cu(torch.randn(B, cout, *shp))
    sh2, sc2 = cu(torch.randn(B, cout)), cu(torch.rand(B, cout) + 0.5)
    outs = {}
    for mode in ("nosplit", "split", "split_again"):
        y, y2 = torch.empty(B, cout, *shp, device="cuda"), torch.empty(B, cout, *shp, device="cuda")
        d = ops.make_conv_desc([x], wp, cout, 3, y, bias=bias, residual=res, out_scale=0.5, in_act=True, y2=y2, y2_act=True,
                               y2_shift=sh2, y2_scale=sc2)
        need = lib.tmdiff_conv3d_fwd_splitk_workspace_bytes(C.byref(d))
        assert need > 0 and need % (B * cout * shp[0] * shp[1] * shp[2] * 4) == 0
        if mode != "nosplit":
            ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), need
        assert lib.tmdiff_conv3d_fwd(C.byref(d), None) == 0
        outs[mode] = (y, y2)
    assert torch.equal(outs["split"][0], outs["split_again"][0]) and torch.equal(outs["split"][1], outs["split_again"][1])
    xd = x.cpu().double()
    xd = xd * torch.sigmoid(xd)
    ref = (F.conv3d(xd, w.double(), bias.cpu().double(), padding=1) + res.cpu().double()) * 0.5
    t = ref + sh2.cpu().double()[:, :, None, None, None]
    ref2 = t * torch.sigmoid(t) * sc2.cpu().double()[:, :, None, None, None]
    for mode in ("nosplit", "split"):
        assert_close(outs[mode][0], ref.float(), 2e-5, 2e-6, f"{mode} y vs fp64")
        assert_close(outs[mode][1], ref2.float(), 2e-5, 2e-6, f"{mode} y2 vs fp64")
    # a grid that fills the chip is never split
    xb = cu(torch.randn(32, 32, 8, 64, 64))
    db = ops.make_conv_desc([xb], ops.pack_conv_weight(cu(torch.randn(32, 32, 3, 3, 3))), 32, 3, torch.empty_like(xb))
    assert lib.tmdiff_conv3d_fwd_splitk_workspace_bytes(C.byref(db)) == 0


def test_conv1_vectorised_equals_dword_kernel(ops, tmp_path):
    """The 16-byte 1x1x1 kernel (a lane owns four consecutive positions; MFMA columns permuted accordingly) against fp64
    and, bit for bit, against the dword kernel it replaces on large grids (child processes with TMDIFF_CONV1_VEC=1 /
    TMDIFF_CONV1_DWORD=1 force either kernel at these small sizes)."""
    import os, subprocess, sys
    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from tmdiff_amd import ops
outs = {}
for tag, (b, cin, cout, shp, segs) in {"a": (2, 32, 64, (8, 16, 16), [32]), "b": (3, 48, 32, (4, 8, 16), [16, 32]),
                                      "c": (1, 64, 128, (8, 32, 32), [64])}.items():
    g = torch.Generator().manual_seed(len(tag) + cin)
    xs = [torch.randn(b, c, *shp, generator=g).cuda() for c in segs]
    w = (torch.randn(cout, cin, 1, 1, 1, generator=g) / cin ** 0.5).cuda()
    kw = dict(bias=torch.randn(cout, generator=g).cuda(), residual=torch.randn(b, cout, *shp, generator=g).cuda(), out_scale=0.5,
              in_shift=torch.randn(b, cin, generator=g).cuda(), in_scale=(torch.rand(b, cin, generator=g) + 0.5).cuda(), in_act=True)
    outs[tag] = ops.conv3d(xs, ops.pack_conv_weight(w), cout, 1, **kw).cpu()
    outs[tag + "_in"] = (torch.cat([x.cpu() for x in xs], 1), w.cpu(), {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in kw.items()})
torch.save(outs, sys.argv[2])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode, env in (("vec", {"TMDIFF_CONV1_VEC": "1"}), ("dword", {"TMDIFF_CONV1_DWORD": "1"})):
        path = str(tmp_path / f"{mode}.pt")
        subprocess.run([sys.executable, "-c", code, root, path], check=True, env=dict(os.environ, **env), timeout=300)
        res[mode] = torch.load(path)
    for tag in ("a", "b", "c"):
        assert torch.equal(res["vec"][tag], res["dword"][tag]), tag
        x, w, kw = res["vec"][tag + "_in"]
        xd = x.double() + kw["in_shift"].double()[:, :, None, None, None]
        xd = xd * torch.sigmoid(xd) * kw["in_scale"].double()[:, :, None, None, None]
        ref = (F.conv3d(xd, w.double(), kw["bias"].double()) + kw["residual"].double()) * 0.5
        assert_close(res["vec"][tag], ref.float(), 2e-5, 2e-6, f"1x1x1 {tag} vs fp64")


@pytest.mark.parametrize("case", [(2, (64, 64, 128), 128, (8, 8, 8)), (3, (192,), 32, (4, 8, 12)), (32, (256, 256, 256), 128, (8, 8, 8)),
                                  (1, (128,), 64, (8, 16, 16))])
def test_conv1_small_planes_split_the_channels_over_the_waves(ops, case):
    """The 1x1x1 kernel on small grids (the 8x8 / 16x16 levels: fewer than 512 tiles of 256 positions, at least 128 input
    channels): the four waves of a workgroup take every fourth 16-channel group of the same 64 positions and their partial sums
    meet in LDS (conv1.hip, KS = 4) -- prologue, segments, bias, residual, scale against fp64."""
    B, seg_c, cout, shp = case
    cin = sum(seg_c)
    torch.manual_seed(cin + cout + B)
    xs = [torch.randn(B, c, *shp) for c in seg_c]
    w = torch.randn(cout, cin, 1, 1, 1) / cin ** 0.5
    bias, res = torch.randn(cout), torch.randn(B, cout, *shp)
    sh, sc = torch.randn(B, cin) * 0.3, torch.rand(B, cin) + 0.5
    wp = ops.pack_conv_weight(cu(w))
    x = torch.cat(xs, 1).double()
    want = F.conv3d(x, w.double(), bias.double())
    assert_close(ops.conv3d([cu(t) for t in xs], wp, cout, 1, bias=cu(bias)), want.float(), 2e-5, 2e-6, "1x1x1, channels over the waves")
    xd = x + sh.double()[:, :, None, None, None]
    xd = xd * torch.sigmoid(xd) * sc.double()[:, :, None, None, None]
    want = (F.conv3d(xd, w.double(), bias.double()) + res.double()) * 0.5
    y = ops.conv3d([cu(t) for t in xs], wp, cout, 1, bias=cu(bias), residual=cu(res), out_scale=0.5, in_shift=cu(sh), in_scale=cu(sc), in_act=True)
    assert_close(y, want.float(), 2e-5, 2e-6, "1x1x1, channels over the waves, prologue + residual")


@pytest.mark.parametrize("case", [(16, (32, 32, 32), 32, (8, 64, 32)), (32, (64, 64), 64, (8, 32, 32)), (16, (16, 48), 96, (4, 64, 64)),
                                  (32, (32, 32, 32), 32, (8, 36, 28)),       # a plane that ends inside a 512-position tile
                                  (40, (64, 64), 64, (4, 44, 36)),
                                  (32, (256, 256, 256), 128, (8, 8, 8)),     # small grids: the kernel that splits the channels over its waves
                                  (32, (128, 128, 128), 64, (8, 16, 16)), (3, (64, 64), 32, (4, 14, 20))])
def test_conv1_writes_the_prologue_of_its_input_on_the_side(ops, case):
    """desc.xp_* (ABI v6): the 1x1x1 bandwidth kernel also writes SiLU(x + shift) of its segmented input -- what a ResBlock's
    conv20 reads of the same concat (reference Hyper_unet_general.py:243-248) -- bit for bit the prologue pass it replaces;
    its own result is unchanged; routing.k1_side_xp agrees with the library about which launches can."""
    from tmdiff_amd import routing
    from tmdiff_amd._lib import lib
    import ctypes as C
    B, seg_c, cout, shp = case
    cin = sum(seg_c)
    torch.manual_seed(cin + cout)
    xs = [torch.randn(B, c, *shp, device="cuda") for c in seg_c]
    w = torch.randn(cout, cin, 1, 1, 1, device="cuda") / cin ** 0.5
    bias, sh = torch.randn(cout, device="cuda"), torch.randn(B, cin, device="cuda") * 0.5
    wp = ops.pack_conv_weight(w)
    assert routing.k1_side_xp(B, seg_c, cout, *shp)
    plain = ops.conv3d(xs, wp, cout, 1, bias=bias)
    for shift, act in ((sh, True), (None, True), (sh, False)):
        side = torch.full((B, cin, *shp), float("nan"), device="cuda")
        y = ops.conv3d(xs, wp, cout, 1, bias=bias, side_xp=dict(out=side, shift=shift, act=act))
        assert torch.equal(y, plain)
        v = torch.cat(xs, 1)
        if shift is not None:
            v = v + shift[:, :, None, None, None]
        want = (v.double() * torch.sigmoid(v.double())).float() if act else v
        assert_close(side, want.cpu(), 2e-6, 1e-6, "side prologue vs torch")
    # ... and the pass it replaces (conv3d_wf's prologue pass, kept through xp_out): the same bits
    w3 = torch.randn(32, cin, 3, 3, 3, device="cuda") / (cin * 27) ** 0.5
    kept = torch.empty(B, cin, *shp, device="cuda")
    ops.conv3d_wf(xs, ops.pack_conv_weight_wino(w3, mode=2, planes=6), 32, in_act=True, in_shift=sh, xp_out=kept)
    side = torch.empty_like(kept)
    ops.conv3d(xs, wp, cout, 1, bias=bias, side_xp=dict(out=side, shift=sh, act=True))
    assert torch.equal(side, kept)
    # launches neither form of the kernel takes are refused, and the rule says so beforehand: a small grid of fewer than 128 input
    # channels (the dword kernel without the channel split), a small grid with a prologue of its own
    small = [x[:1, :, :, :8, :8].contiguous() for x in xs]
    d = ops.make_conv_desc(small, wp, cout, 1, torch.empty(1, cout, shp[0], 8, 8, device="cuda"),
                           side_xp=dict(out=torch.empty(1, cin, shp[0], 8, 8, device="cuda"), act=True))
    assert bool(lib.tmdiff_conv3d_fwd_xp_supported(C.byref(d))) == routing.k1_side_xp(1, seg_c, cout, shp[0], 8, 8) == (cin >= 128)
    if cin < 128:
        with pytest.raises(ValueError):
            ops.conv3d(small, wp, cout, 1, side_xp=dict(out=torch.empty(1, cin, shp[0], 8, 8, device="cuda"), act=True))
    else:
        with pytest.raises(ValueError):
            ops.conv3d(small, wp, cout, 1, in_shift=sh[:1].contiguous(), side_xp=dict(out=torch.empty(1, cin, shp[0], 8, 8, device="cuda"), act=True))
    with pytest.raises(ValueError):
        ops.conv3d(xs, ops.pack_conv_weight(w3), 32, 3, side_xp=dict(out=side, act=True))


def test_multi_tensor_weight_packing_equals_single(ops):
    """tmdiff_conv3d_pack_weights_multi (every weight of a network, forward + data-gradient packing, one launch, LDS-tiled
    transposes) against tmdiff_conv3d_pack_weights tensor by tensor: identical buffers; refresh() re-packs on a version bump."""
    torch.manual_seed(41)
    shapes = [(32, 32, 3, 1), (64, 32, 3, 1), (32, 96, 3, 1), (128, 64, 1, 1), (96, 48, 3, 3), (24, 20, 3, 1), (8, 4, 1, 1),
              (256, 768, 3, 1)]
    ws = [(torch.nn.Parameter(cu(torch.randn(co, ci // g, k, k, k))), g) for co, ci, k, g in shapes]
    pk = ops.PackedWeights(ws).refresh()
    for w, g in ws:
        f, d = pk.lookup(w)
        assert torch.equal(f, ops.pack_conv_weight(w, groups=g, mode=0)), tuple(w.shape)
        assert torch.equal(d, ops.pack_conv_weight(w, groups=g, mode=1)), tuple(w.shape)
    with torch.no_grad():
        ws[2][0].mul_(2.0)                       # an optimizer step bumps the version
    assert pk.lookup(ws[2][0]) is None and pk.lookup(ws[1][0]) is not None
    pk.refresh()
    assert torch.equal(pk.lookup(ws[2][0])[0], ops.pack_conv_weight(ws[2][0], groups=1, mode=0))


def test_multi_tensor_winograd_weight_packing_equals_single(ops):
    """tmdiff_conv3d_wino_pack_weights_multi (every weight, forward + data-gradient form, one launch: the finetune step's
    re-pack) against the single-tensor packing, bit for bit; and its refresh / lookup bookkeeping."""
    torch.manual_seed(3)
    ws = [(torch.randn(64, 32, 3, 3, 3, device="cuda"), 1), (torch.randn(32, 32, 3, 3, 3, device="cuda"), 1),
          (torch.randn(96, 32, 3, 3, 3, device="cuda"), 3), (torch.randn(128, 64, 3, 3, 3, device="cuda"), 1)]
    pk = ops.WinoPackedWeights()
    first = {(i, mode): pk.get(w, g, mode).clone() for i, (w, g) in enumerate(ws) for mode in (2, 3)}    # packed singly, registered
    for i, (w, g) in enumerate(ws):
        for mode in (2, 3):
            assert torch.equal(first[(i, mode)], ops.pack_conv_weight_wino(w, groups=g, mode=mode, planes=6))
    for w, _ in ws:
        w.mul_(1.5)                                     # (an optimizer step: every version changes)
    pk.refresh()                                        # ONE launch re-packs all eight
    for i, (w, g) in enumerate(ws):
        for mode in (2, 3):
            got = pk.get(w, g, mode)
            assert torch.equal(got, ops.pack_conv_weight_wino(w, groups=g, mode=mode, planes=6)), (tuple(w.shape), g, mode)
            assert not torch.equal(got, first[(i, mode)])
    ws[1][0].mul_(2.0)                                  # changed after the refresh: re-packed on demand
    assert torch.equal(pk.get(ws[1][0], 1, 3), ops.pack_conv_weight_wino(ws[1][0], groups=1, mode=3, planes=6))
