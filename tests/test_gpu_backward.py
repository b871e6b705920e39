"""GPU parity of the differentiable path against the REFERENCE-generated fixtures that carry gradients
(tests/golden/blocks.npz, modconv.npz, haar.npz) and, elementwise, against the pinned oracle's CPU autograd:
every block forward + input / embedding / parameter gradients at N = 4 and 8, the modulated convolution, the Haar
transforms, a full-width (ch 32-256) forward + backward, dropout with host-supplied masks, the grad-mode dispatch of
``WavBEST.forward`` and the EMA update.  Reference: GeneralModel/Hyper_unet_general.py:51-77, :158-273, :334-414, :600-636;
DWT_IDWT/DWT_IDWT_Functions.py:47-112; utils/EmaUpdater.py:23-38."""
import copy

import numpy as np
import pytest
import torch

from conftest import assert_close, rel_err
from oracle import unet_ref as U
from oracle.make_golden import FULL, TINY, case_inputs, randn

pytestmark = pytest.mark.gpu
E = 128


def cu(t):
    return t.detach().cuda().contiguous()


def leaf(t):
    return cu(t).requires_grad_(True)


# ---- Haar DWT / IDWT gradients (haar.npz: *_gx, *_idwt_g*) --------------------------------------------------------------
@pytest.mark.parametrize("tag,shape", [("a", (2, 6, 8, 8)), ("b", (1, 16, 16, 12))])
def test_haar_gradients_vs_reference(golden, tag, shape):
    from tmdiff_amd import autograd as A
    g = golden("haar")
    x = leaf(randn(11, *shape))
    bands = A.haar_dwt2d(x, want_high=True)
    for n, v in zip(("ll", "lh", "hl", "hh"), bands):
        assert_close(v.detach().cpu(), g[f"{tag}_{n}"], 1e-6, 1e-6, f"dwt {n}")
    gout = [cu(randn(12 + i, *bands[0].shape)) for i in range(4)]
    torch.autograd.backward(bands, gout)
    assert_close(x.grad.cpu(), g[f"{tag}_gx"], 1e-6, 1e-6, "dwt input gradient")
    # IDWT through the paired operator the network uses: (h_up, x_up) = IDWT(h | x, stacked bands [B, 3C, N, h, w])
    b, c, h, w = bands[0].shape
    ins = [randn(20 + i, b, c, h, w) for i in range(4)]
    ll = leaf(ins[0].reshape(b, c, 1, h, w))
    other = leaf(torch.zeros(b, c, 1, h, w))
    stacked = leaf(torch.cat(ins[1:], dim=1).reshape(b, 3 * c, 1, h, w))
    y, _ = A.haar_idwt2d_pair(ll, other, stacked, in_scale=1.0)
    assert_close(y.detach().cpu().reshape(b, c, 2 * h, 2 * w), g[f"{tag}_idwt"], 1e-6, 1e-6, "idwt")
    y.backward(cu(randn(30, b, c, 2 * h, 2 * w)).reshape(y.shape))
    assert_close(ll.grad.cpu().reshape(b, c, h, w), g[f"{tag}_idwt_gll"], 1e-6, 1e-6, "idwt d/dll")
    gb = stacked.grad.cpu().reshape(b, 3, c, h, w)
    for k, n in enumerate(("lh", "hl", "hh")):
        assert_close(gb[:, k], g[f"{tag}_idwt_g{n}"], 1e-6, 1e-6, f"idwt d/d{n}")
    assert float(other.grad.abs().max()) == 0.0


# ---- modulated_conv3d gradients (modconv.npz: gx, gw, gs) ------------------------------------------------------------
@pytest.mark.parametrize("b,k", [(1, 1), (1, 3), (3, 1), (3, 3)])
def test_modconv_gradients_vs_reference(golden, b, k):
    from tmdiff_amd import autograd as A
    g = golden("modconv")
    x = leaf(randn(40, b, 5, 4, 6, 6))
    w = leaf(randn(41, 7, 5, k, k, k) / (5 * k ** 3) ** 0.5)
    s = leaf((1 + 0.3 * randn(42, b, 5, 1, 1)).reshape(b, 5))
    y = A.conv3d([x], w, None, scale=s)
    assert_close(y.detach().cpu(), g[f"b{b}k{k}_y"], 1e-5, 1e-5, "modconv y")
    y.backward(cu(randn(43, *y.shape)))
    assert_close(x.grad.cpu(), g[f"b{b}k{k}_gx"], 3e-5, 3e-5, "modconv gx")
    assert_close(w.grad.cpu(), g[f"b{b}k{k}_gw"], 3e-5, 3e-5, "modconv gw")
    assert_close(s.grad.cpu().reshape(b, 5, 1, 1), g[f"b{b}k{k}_gs"], 3e-5, 3e-5, "modconv gs")


# ---- every block, forward + all gradients (blocks.npz) ----------------------------------------------------------------
def _pair(ref_mod, hip_cls, *ctor, **kw):
    """Reference-shaped oracle block with the fixture's weights + the HIP block holding the same weights."""
    from tmdiff_amd import Hyper_unet_general as H
    U.fill_weights_(ref_mod, seed=7).eval()
    hip = getattr(H, hip_cls)(*ctor, **kw)
    hip.load_state_dict(ref_mod.state_dict())
    return ref_mod, hip.cuda().eval()


def _check_block(g, tag, ref_mod, hip_mod, args, seed=60, grad_names=None):
    """args: CPU leaf tensors (or lists of them).  Runs oracle and HIP block, compares outputs and input gradients with
    the REFERENCE fixture elementwise, parameter-gradient checksums with the fixture, parameter gradients elementwise
    with the oracle (itself pinned to the same fixture by tests/test_oracle_golden.py)."""
    flat = lambda a: [t for v in a for t in (v if isinstance(v, (list, tuple)) else [v])]
    dev_args = [[leaf(t) for t in v] if isinstance(v, (list, tuple)) else leaf(v) for v in args]
    out_r, out_h = ref_mod(*args), hip_mod(*dev_args)
    outs_r = [out_r] if torch.is_tensor(out_r) else [out_r[0]] + list(out_r[1])
    outs_h = [out_h] if torch.is_tensor(out_h) else [out_h[0]] + list(out_h[1])
    gouts = [randn(seed + i, *o.shape) for i, o in enumerate(outs_r)]
    torch.autograd.backward(outs_r, gouts)
    torch.autograd.backward(outs_h, [cu(t) for t in gouts])
    for i, (a, b) in enumerate(zip(outs_h, outs_r)):
        assert_close(a.detach().cpu(), g[f"{tag}_y{i}"], 1e-5, 1e-5, f"{tag} y{i} vs reference")
    names = grad_names or [f"gin{i}" for i in range(len(flat(args)))]
    for name, a, b in zip(names, flat(dev_args), flat(args)):
        key = f"{tag}_{name}"
        if key not in g.files:                       # the reference produced no gradient (flag=True ignores temb)
            assert b.grad is None and (a.grad is None or float(a.grad.abs().max()) == 0.0), key
            continue
        assert_close(a.grad.cpu(), g[key], 3e-5, 3e-5, f"{key} vs reference")
    n = 0
    for (k, p_h), (_, p_r) in zip(hip_mod.named_parameters(), ref_mod.named_parameters()):
        key = f"{tag}_gp_{k}"
        if p_r.grad is None:
            assert key not in g.files and p_h.grad is None, f"{k}: gradient-free in the reference"
            continue
        assert_close(p_h.grad.cpu(), p_r.grad, 3e-5, 3e-5, f"{tag} d/d{k} vs oracle autograd")
        if key in g.files:
            ref = g[key]
            got = torch.stack([p_h.grad.sum(), p_h.grad.abs().sum()]).cpu().numpy()
            assert abs(got[1] - ref[1]) <= 2e-4 * max(ref[1], 1e-6) and abs(got[0] - ref[0]) <= 2e-4 * max(ref[1], 1e-6), (k, got, ref)
            n += 1
    assert n > 0 or not any(k.startswith(f"{tag}_gp_") for k in g.files)   # (the fixture's `up` case stores no checksums)


@pytest.mark.parametrize("n", [4, 8])
def test_blocks_forward_backward_vs_reference(golden, n):
    g = golden("blocks")
    temb, pemb = randn(50, 2, E), randn(51, 2, E)
    mk = lambda seed, ch, hh: randn(seed, 2, ch, n, hh, hh).requires_grad_(True)
    emb = lambda: (temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True))
    _check_block(g, f"n{n}_adaption", *_pair(U.AdaptionModulateBEST(1, 4, E), "AdaptionModulateBEST", 1, 4, E),
                 (mk(52, 1, 16), *emb()))
    _check_block(g, f"n{n}_res", *_pair(U.ResBlockModulateBEST(4, 8, E), "ResBlockModulateBEST", 4, 8, E),
                 (mk(53, 4, 16), *emb()))
    _check_block(g, f"n{n}_res_same_flag", *_pair(U.ResBlockModulateBEST(8, 8, E, flag=True), "ResBlockModulateBEST", 8, 8, E,
                                                  flag=True), (mk(54, 8, 16), *emb()))
    _check_block(g, f"n{n}_down", *_pair(U.ResblockDownOneModulateBEST(4, 8, E), "ResblockDownOneModulateBEST", 4, 8, E),
                 (mk(55, 4, 16), *emb()))
    _check_block(g, f"n{n}_down_flag", *_pair(U.ResblockDownOneModulateBEST(4, 8, E, flag=True),
                                              "ResblockDownOneModulateBEST", 4, 8, E, flag=True), (mk(56, 4, 16), *emb()))
    te, pe = emb()
    skip = [mk(57 + i, 16, 8) for i in range(3)]
    _check_block(g, f"n{n}_up", *_pair(U.ResblockUpOneModulateBEST(16, 8, E), "ResblockUpOneModulateBEST", 16, 8, E),
                 (mk(61, 48, 8), te, skip, pe), seed=62, grad_names=["gin0", "gte", "gskip0", "gskip1", "gskip2", "gpe"])
    _check_block(g, f"n{n}_final", *_pair(U.FinalBlockModulateBEST(4, 1, E), "FinalBlockModulateBEST", 4, 1, E),
                 (mk(63, 12, 16), *emb()))


# ---- whole network, full width: forward + backward elementwise against the oracle's autograd ----------------------------
NAMED = ["embed.0.weight", "embed2.4.weight", "conv1.conv21.weight", "down1.conv20.conv20.weight", "down2_1.down.Conv_1.weight",
         "middle1.conv21.weight", "up1.up1.convH_0.0.weight", "up2.conv20.conv20.weight", "up3.up1.Dense_0.weight",
         "final.conv20.res_conv.weight", "final.conv24.weight", "down3.conv20.dense1.dense.bias", "conv2.conv20.weight"]


def test_full_width_forward_backward_vs_oracle():
    """ch 32-256 (BASELINE configs[1]/[3] widths), B = 2, 8 x 16 x 16, dropout off: output, d/dx_t and 13 named parameter
    gradients (stems, every block family, both embedding MLPs, a modulation Dense, the grouped skip conv, the head)
    elementwise against the oracle's CPU autograd; the other gradient-carrying tensors by checksum."""
    from tmdiff_amd.Hyper_unet_general import WavBEST
    ref = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    hip = WavBEST(channels=FULL)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda().eval()
    d = case_inputs(911, 2, 8, 16)
    t = torch.tensor([[37], [801]])
    x_r = d["x_t"].clone().requires_grad_(True)
    y_r = ref(x_r, t, d["PAN"], d["MS"], "WV3")
    gy = randn(912, *y_r.shape)
    y_r.backward(gy)
    x_h = leaf(d["x_t"])
    y_h = hip(x_h, t.cuda(), cu(d["PAN"]), cu(d["MS"]), "WV3")          # grad mode on -> the differentiable path
    assert y_h.requires_grad
    y_h.backward(cu(gy))
    m, l2 = rel_err(y_h.detach().cpu(), y_r.detach())
    print(f"full-width training forward: max-rel {m:.2e} rel-L2 {l2:.2e}")
    assert m <= 1e-4 and l2 <= 1e-5
    assert_close(x_h.grad.cpu(), x_r.grad, 1e-4, 2e-5, "d/dx_t")
    pr, ph = dict(ref.named_parameters()), dict(hip.named_parameters())
    for name in NAMED:
        m, l2 = rel_err(ph[name].grad.cpu(), pr[name].grad)
        print(f"  d/d{name}: max-rel {m:.2e} rel-L2 {l2:.2e}")
        assert m <= 1e-4 and l2 <= 2e-5, name
    n_free = 0
    for name, p in pr.items():
        if p.grad is None:
            assert ph[name].grad is None, name
            n_free += 1
            continue
        a, b = ph[name].grad.cpu().double(), p.grad.double()
        assert abs(float(a.sum() - b.sum())) <= 1e-4 * float(b.abs().sum()) + 1e-7, name
        assert abs(float(a.abs().sum() - b.abs().sum())) <= 1e-4 * float(b.abs().sum()) + 1e-7, name
    assert n_free == 56                                 # 56 tensors never receive a gradient


# ---- dropout: host-supplied masks on both sides ------------------------------------------------------------------------
class _FedDropout(torch.nn.Module):
    def __init__(self, feed):
        super().__init__()
        self.feed = feed

    def forward(self, x):
        return x * self.feed(tuple(x.shape))


def test_dropout_with_host_masks_vs_oracle():
    """train() mode, Dropout(0.2) active (ref :230, :243-246, :349, :403): both sides consume the same mask stream (drawn
    from one seeded CPU generator in module-call order); loss-free forward + gradients must agree."""
    from tmdiff_amd import Hyper_unet_general as H
    ref = U.fill_weights_(U.WavBESTRef(channels=TINY)).train()
    hip = H.WavBEST(channels=TINY)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda().train()

    def feeder():
        gen, shapes = torch.Generator().manual_seed(77), []

        def feed(shape):
            shapes.append(shape)
            return (torch.rand(shape, generator=gen) >= 0.2).float() / 0.8
        return feed, shapes

    feed_r, shapes_r = feeder()
    for mod in list(ref.modules()):
        for name, child in list(mod.named_children()):
            if isinstance(child, torch.nn.Dropout):
                setattr(mod, name, _FedDropout(feed_r))
    d = case_inputs(921, 2, 8, 16)
    t = torch.tensor([[5], [640]])
    y_r = ref(d["x_t"], t, d["PAN"], d["MS"], "GF2")
    gy = randn(922, *y_r.shape)
    y_r.backward(gy)
    feed_h, shapes_h = feeder()
    H.set_dropout_mask_fn(feed_h)
    try:
        y_h = hip(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), "GF2")
        y_h.backward(cu(gy))
    finally:
        H.set_dropout_mask_fn(None)
    assert shapes_h == shapes_r and len(shapes_r) == 2 * 14 + 9      # 14 ResBlocks x 2 + 9 wavelet blocks
    assert_close(y_h.detach().cpu(), y_r.detach(), 1e-4, 1e-5, "forward with dropout masks")
    for (k, p_r), (_, p_h) in zip(ref.named_parameters(), hip.named_parameters()):
        if p_r.grad is None:
            assert p_h.grad is None, k
        else:
            assert_close(p_h.grad.cpu(), p_r.grad, 1e-4, 3e-5, f"d/d{k} with dropout masks")
    # without a mask hook the device generator drives dropout: two runs differ, eval() is deterministic
    a = hip(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), "GF2").detach()
    b = hip(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), "GF2").detach()
    assert not torch.equal(a, b)


# ---- WavBEST.forward dispatch and the condition-cache key ----------------------------------------------------------------
def test_forward_grad_mode_dispatch_and_cache_key():
    from tmdiff_amd.Hyper_unet_general import WavBEST
    ref = U.fill_weights_(U.WavBESTRef(channels=TINY)).eval()
    hip = WavBEST(channels=TINY)
    hip.load_state_dict(ref.state_dict())
    hip = hip.cuda().eval()
    d = {k: cu(v) for k, v in case_inputs(931, 2, 8, 16).items()}
    t = torch.tensor([[9], [333]]).cuda()
    y_grad = hip(d["x_t"], t, d["PAN"], d["MS"], "WV3")
    assert y_grad.requires_grad and y_grad.grad_fn is not None          # the reference's forward is differentiable
    with torch.no_grad():
        y_inf = hip(d["x_t"], t, d["PAN"], d["MS"], "WV3")
    assert not y_inf.requires_grad
    assert_close(y_inf, y_grad.detach(), 1e-5, 1e-6, "inference path vs differentiable path")
    for p in hip.parameters():
        p.requires_grad_(False)
    assert not hip(d["x_t"], t, d["PAN"], d["MS"], "WV3").requires_grad    # nothing to differentiate: fast path
    x = d["x_t"].clone().requires_grad_(True)
    hip(x, t, d["PAN"], d["MS"], "WV3").sum().backward()                # frozen weights, guidance-style d/dx_t
    assert x.grad is not None and torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    # condition cache: keyed on storage + version, so an in-place update of MS is noticed
    with torch.no_grad():
        ms = d["MS"].clone()
        hip.begin_condition_cache(d["PAN"], ms, "WV3")
        try:
            y0 = hip(d["x_t"], t, d["PAN"], ms, "WV3")
            assert torch.equal(y0, y_inf)
            ms.mul_(0.5)
            y1 = hip(d["x_t"], t, d["PAN"], ms, "WV3")
        finally:
            hip.end_condition_cache()
        want = hip(d["x_t"], t, d["PAN"], ms.clone(), "WV3")
        assert torch.equal(y1, want) and not torch.equal(y1, y0)


def test_ema_update_refreshes_packed_weights(tmp_path):
    """EmaUpdater writes through .data / a raw kernel (no version bump); the EMA model must not keep serving
    the packed weights of its previous state (utils/EmaUpdater.py:23-38)."""
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.model import EmaUpdater, create_model
    opt = {"phase": "train", "gpu_ids": [0], "distributed": False, "path": {"resume": None},
           "model": {"unet": {"channel_multiplier": TINY}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
           "train": {"optimizer": {"lr": 1e-2}, "max_iter": 100}}
    torch.manual_seed(3)
    m = create_model(opt)
    ema = EmaUpdater(m, copy.deepcopy(m), decay=0.5)
    m.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, "train")
    d = case_inputs(941, 2, 8, 16)
    d["LR"] = d["MS"].clone()
    dc = {k: cu(v) for k, v in d.items()}
    t = torch.tensor([[4], [7]]).cuda()
    net_ema = ema.ema_model.netG.denoise_fn.eval()
    with torch.no_grad():
        before = net_ema(dc["x_t"], t, dc["PAN"], dc["MS"], "WV3")       # packs the EMA model's weights
    for it in range(1, 3):
        m.feed_data({k: v.clone() for k, v in d.items()})
        m.optimize_parameters("WV3")
        ema.update(it)
    with torch.no_grad():
        after = net_ema(dc["x_t"], t, dc["PAN"], dc["MS"], "WV3")
        fresh = WavBEST(channels=TINY)
        fresh.load_state_dict(net_ema.state_dict())
        want = fresh.cuda().eval()(dc["x_t"], t, dc["PAN"], dc["MS"], "WV3")
    assert not torch.equal(after, before)
    assert torch.equal(after, want)


def test_in_kernel_dropout_matches_explicit_mask():
    """Dropout inside the kernels (no mask tensor): the keep mask is a pure function of (seed, element index), so the
    forward (fused and staged kernels), the weight gradient and the prologue backward all see the same mask.  Recover the
    mask through an identity convolution, then compare every output of the DropSpec path with the same operator fed that
    mask as a tensor (the path that test_dropout_with_host_masks_vs_oracle pins to the oracle)."""
    from tmdiff_amd import autograd as A, ops
    torch.manual_seed(5)
    b, cin, cout, shp = 2, 8, 32, (4, 16, 16)
    seed, p = 123456789123, 0.2
    ident = torch.zeros(cin, cin, 3, 3, 3)
    for c in range(cin):
        ident[c, c, 1, 1, 1] = 1.0
    ones = torch.ones(b, cin, *shp, device="cuda")
    wp = ops.pack_conv_weight(cu(ident))
    m_fused = ops.conv3d([ones], wp, cin, 3, drop=(seed, p), staged=False)
    m_staged = ops.conv3d([ones], wp, cin, 3, drop=(seed, p), staged=True)
    assert torch.equal(m_fused, m_staged)
    vals = torch.unique(m_fused)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.25) < 1e-6
    keep = float((m_fused > 0).float().mean())
    assert abs(keep - 0.8) < 0.01, keep                      # 16k elements: sigma = 0.003
    per_plane = (m_fused > 0).float().flatten(2).mean(-1)
    assert float(per_plane.min()) > 0.7 and float(per_plane.max()) < 0.9
    assert torch.equal(ops.conv3d([ones], wp, cin, 3, drop=(seed, p)), m_fused)           # repeatable
    other = ops.conv3d([ones], wp, cin, 3, drop=(seed + 1, p))
    assert 0.55 < float(((other > 0) == (m_fused > 0)).float().mean()) < 0.8             # independent: 0.68 expected
    # full operator: forward and every gradient, DropSpec vs explicit mask tensor
    mk = lambda s, *shape: leaf(randn(s, *shape))
    outs = []
    for mask in (A.DropSpec(seed, p), m_fused.clone()):
        x, w, bias = mk(1, b, cin, *shp), mk(2, cout, cin, 3, 3, 3), mk(3, cout)
        shift, scale, res = mk(4, b, cin), leaf(1 + 0.3 * randn(5, b, cin)), mk(6, b, cout, *shp)
        y = A.conv3d([x], w, bias, shift=shift, scale=scale, act=True, mask=mask, residual=res)
        y.backward(cu(randn(7, *y.shape)))
        outs.append([y.detach()] + [t.grad for t in (x, w, bias, shift, scale, res)])
    for name, a_, b_ in zip(("y", "dx", "dw", "dbias", "dshift", "dscale", "dres"), *outs):
        assert_close(a_, b_, 1e-6, 1e-6, f"in-kernel dropout {name}")


@pytest.mark.parametrize("shape", [(2, 32, 64, 3, 1), (1, 24, 40, 3, 1), (2, 48, 96, 3, 3), (2, 32, 32, 1, 1)])
def test_wgrad_with_bias_gradient_on_the_side(shape):
    """tmdiff_conv3d_wgrad_bias: dw as tmdiff_conv3d_wgrad (bit for bit) and dbias = bias_scale * sum g against the
    channel-sum kernel / fp64."""
    from tmdiff_amd import ops
    b, cin, cout, k, groups = shape
    x, g = cu(randn(1, b, cin, 4, 8, 16)), cu(randn(2, b, cout, 4, 8, 16))
    sc = cu(torch.rand(b, cin) + 0.5)
    d = ops.make_conv_desc([x], 0, cout, k, g, groups=groups, in_scale=sc, in_act=True, bias_scale=2.0)
    wshape = (cout, cin // groups, k, k, k)
    dw0 = ops.conv3d_wgrad(d, g, wshape)
    dw1, db = ops.conv3d_wgrad(d, g, wshape, want_bias=True)
    assert torch.equal(dw0, dw1)
    want = 2.0 * g.double().sum(dim=(0, 2, 3, 4))
    assert_close(db.cpu(), want.float().cpu(), 1e-5, 1e-5, "dbias inside wgrad")
    assert_close(ops.channel_sum(g, 2.0).cpu(), want.float().cpu(), 1e-5, 1e-5, "channel_sum")


@pytest.mark.parametrize("case", [
    # B, segs, Cout, N, H, W, groups, prologue
    (2, (32,), 64, 8, 16, 16, 1, False),       # two band tiles, whole boxes, 2 x 1 channel tiles
    (1, (24,), 40, 8, 12, 20, 1, False),       # ragged h / w, channel counts that are no multiples of 32
    (2, (16, 8, 8), 32, 4, 16, 32, 1, True),   # N = 4 (one band tile), three segments + prologue (the x' pass first)
    (2, (16, 16, 16), 96, 8, 8, 16, 3, False), # groups = 3 on three plain segments
    (3, (64,), 64, 8, 8, 8, 1, False),         # 8-column planes: the 8 x 8 box
    (1, (8,), 8, 4, 8, 136, 1, False),         # three column chunks of the transform pass, ragged last chunk
    (4, (32,), 32, 8, 32, 32, 1, False),       # one channel tile, many boxes: split over positions
    (1, (4,), 4, 12, 8, 16, 1, False),         # three band tiles
    (2, (64,), 64, 8, 64, 64, 1, False),       # 128 boxes per plane class: several boxes per workgroup (double buffering)
    (1, (32,), 32, 8, 40, 48, 1, False),       # an odd number of boxes per workgroup
])
def test_weight_gradient_in_the_winograd_domain(case):
    """tmdiff_conv3d_wgrad_wino: F(3,4) along the bands (transform passes + per-plane MFMA accumulation + reduction with
    A'^T) against CPU autograd in fp64 and against the direct weight-gradient kernel; deterministic."""
    import torch.nn.functional as F
    from tmdiff_amd import ops
    B, segc, cout, N, H, W, groups, pro = case
    cin = sum(segc)
    torch.manual_seed(11 + cout + W)
    segs = [torch.randn(B, c, N, H, W) for c in segc]
    g = torch.randn(B, cout, N, H, W)
    sh, sc = torch.randn(B, cin) * 0.3, torch.rand(B, cin) + 0.5
    xs = torch.cat(segs, 1).double()
    if pro:
        xs = xs + sh[:, :, None, None, None].double()
        xs = xs * torch.sigmoid(xs) * sc[:, :, None, None, None].double()
    with torch.enable_grad():
        wd = torch.zeros(cout, cin // groups, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv3d(xs, wd, padding=1, groups=groups).backward(g.double())
    kw = dict(in_shift=cu(sh), in_scale=cu(sc), in_act=True) if pro else {}
    gd, xd = cu(g), [cu(s_) for s_ in segs]          # (the descriptor holds bare pointers: the tensors must outlive it)
    d = ops.make_conv_desc(xd, 0, cout, 3, gd, groups=groups, **kw)
    from tmdiff_amd import _lib
    import ctypes as C
    assert _lib.lib.tmdiff_conv3d_wgrad_wino_supported(C.byref(d)) == 1
    counts = ops.COUNTS
    ops.COUNTS = __import__("collections").Counter()
    try:
        dw = ops.conv3d_wgrad(d, gd, tuple(wd.shape))
        again = ops.conv3d_wgrad(d, gd, tuple(wd.shape))
        assert ops.COUNTS["conv3d_wgrad_wino"] == 2 and ops.COUNTS["conv3d_wgrad"] == 0
    finally:
        ops.COUNTS = counts
    assert torch.equal(dw, again)
    scale = float(wd.grad.abs().max())
    err = float((dw.cpu().double() - wd.grad).abs().max()) / scale
    l2 = float((dw.cpu().double() - wd.grad).norm() / wd.grad.norm())
    print(f"winograd wgrad {case}: max err / max |dw| {err:.2e}, rel-L2 {l2:.2e}")
    assert err <= 1e-5 and l2 <= 3e-6
    keep, ops.config.wgrad_wino = ops.config.wgrad_wino, False
    try:
        direct = ops.conv3d_wgrad(d, gd, tuple(wd.shape))
    finally:
        ops.config.wgrad_wino = keep
    l2d = float((direct.cpu().double() - wd.grad).norm() / wd.grad.norm())
    print(f"   direct kernel rel-L2 {l2d:.2e}")
    assert float((dw - direct).abs().max()) / scale <= 1e-5
    # with the bias gradient on the side (a channel sum beside the Winograd kernel)
    d.bias_scale = 2.0
    dw2, db = ops.conv3d_wgrad(d, gd, tuple(wd.shape), want_bias=True)
    assert torch.equal(dw2, dw)
    assert_close(db.cpu(), (2.0 * g.double().sum(dim=(0, 2, 3, 4))).float(), 1e-5, 1e-5, "dbias beside the Winograd weight gradient")


@pytest.mark.parametrize("shape", [(2, 8, 64, 4, 16, 16), (1, 6, 128, 3, 12, 20), (8, 16, 64, 8, 32, 64)])
def test_composed_conv_ll_gradients_vs_cpu_autograd(shape):
    """autograd.conv3d_ll = LL(conv3d(SiLU(x), w) + b) / 2 with the forward as ONE strided convolution on composed weights
    (down blocks whose high bands are dropped): output and the gradients w.r.t. x, w, b against CPU autograd of the
    reference's operator order (convolution at full resolution, then the Haar LL band, halved)."""
    import torch.nn.functional as F
    from tmdiff_amd import autograd as A
    B, cin, cout, N, H, W = shape
    torch.manual_seed(23)
    x = torch.randn(B, cin, N, H, W, requires_grad=True)
    w = (torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5).requires_grad_()
    b = torch.randn(cout, requires_grad=True)
    r = torch.randn(B, cout, N, H // 2, W // 2)
    full = F.conv3d(F.silu(x), w, b, padding=1)
    want = 0.25 * (full[..., 0::2, 0::2] + full[..., 0::2, 1::2] + full[..., 1::2, 0::2] + full[..., 1::2, 1::2])
    (want * r).sum().backward()
    xg, wg, bg = (t.detach().cuda().requires_grad_() for t in (x, w, b))
    from tmdiff_amd import ops
    counts, ops.COUNTS = ops.COUNTS, __import__("collections").Counter()
    try:
        got = A.conv3d_ll(xg, wg, bg, 0.5)
        ran = dict(ops.COUNTS)
    finally:
        ops.COUNTS = counts
    # (the large case is taken by the composed-LL mode of the Winograd kernel, the small ones by conv3d_ll)
    assert ran.get("conv3d_wfll_fwd", 0) + ran.get("conv3d_ll_fwd", 0) == 1 and (B < 8 or ran.get("conv3d_wfll_fwd", 0) == 1), ran
    (got * r.cuda()).sum().backward()
    assert_close(got.detach(), want.detach(), 2e-5, 2e-6, "conv3d_ll forward")
    assert_close(xg.grad, x.grad, 3e-5, 3e-6, "d/dx")
    assert_close(wg.grad, w.grad, 3e-5, 3e-6, "d/dw")
    assert_close(bg.grad, b.grad, 3e-5, 3e-6, "d/db")
