"""BASELINE.json configs exercised at their own sizes on the GPU (VERDICT r1 "configs_untested"):

  configs[1]  batch-32 8-ch 64x64 tiles, ch 32-256 (the benchmark workload): reference fixture + replication property;
  configs[2]  channel_multiplier [64,128,256,512] (config/general.json:52-54), 8-ch 64x64 and 256x256 forwards against
              REFERENCE-generated fixtures (tests/golden/unet_c3.npz) in fp32 and in the bf16 compute mode, and the
              21-NFE DPM-Solver++ run in bf16 against the oracle;
  configs[3]  finetune step at full width, local batch 8, 8-ch 64x64 (general_finetune.json:64-66);
  configs[4]  mixed GF-2 (4 bands) / WV-3 (8 bands) sub-batches with per-sample prompts at full width, tiled scene.
"""
import numpy as np
import os

import pytest
import torch

from conftest import assert_close, rel_err
from oracle import unet_ref as U
from oracle.diffusion_ref import GeneralDiffusionRef
from oracle.make_golden import FULL, WIDE, case_inputs, randn

pytestmark = pytest.mark.gpu


def cu(t):
    return t.detach().cuda().contiguous()


def _hip_net(channels, ref=None):
    from tmdiff_amd.Hyper_unet_general import WavBEST
    net = WavBEST(channels=channels)
    if ref is not None:
        net.load_state_dict(ref.state_dict())
    else:
        U.fill_weights_(net)
    return net.cuda().eval()


# ---- configs[1]: batch 32 -------------------------------------------------------------------------------------------
def test_config2_batch32_full_width(golden):
    """B = 32 selects other tile instantiations than B <= 4 at the 32x32 / 16x16 / 8x8 levels (conv3d.hip dispatch): a
    replicated reference tile must come back 32 times, bit-identical, and equal to the reference's output."""
    net = _hip_net(FULL)
    d = case_inputs(3407, 1, 8, 64)
    rep = lambda t: cu(t.repeat(32, *([1] * (t.dim() - 1))))
    with torch.no_grad():
        y = net(rep(d["x_t"]), torch.full((32, 1), 250).cuda(), rep(d["PAN"]), rep(d["MS"]), "WV3").cpu()
    for i in range(1, 32):
        assert torch.equal(y[i], y[0]), f"row {i} differs from row 0"
    m, l2 = rel_err(y[:1], golden("unet_full")["y"])
    print(f"B=32 full-width forward vs reference: max-rel {m:.3e} rel-L2 {l2:.3e}")
    assert m <= 1e-4 and l2 <= 1e-5
    # distinct tiles in one batch of 32 equal the same tiles run four at a time (batch independence at B = 32)
    d32 = {k: cu(v) for k, v in case_inputs(11, 32, 8, 64).items()}
    t = torch.arange(1, 33, dtype=torch.float32).reshape(32, 1).cuda() * 31
    with torch.no_grad():
        y32 = net(d32["x_t"], t, d32["PAN"], d32["MS"], "WV3")
        y4 = net(d32["x_t"][12:16].contiguous(), t[12:16].contiguous(), d32["PAN"][12:16].contiguous(),
                 d32["MS"][12:16].contiguous(), "WV3")
    # (other tile configurations and split-K factors at B = 4: a different fp32 summation order, not different maths)
    assert_close(y32[12:16], y4, 1e-5, 1e-5, "rows 12..15 of a batch of 32 vs a batch of 4")


@pytest.mark.parametrize("switch,what", [
    ("fuse_res_conv", "k1"),        # res_conv / Conv_2 folded into the consumer's epilogue: fewer 1x1x1 launches
    ("emit_ll", "dwt"),             # the ResBlock in front of a down block writes LL(y) / 2: no LL-only Haar pass
    ("emit_dwt", "dwt"),            # Conv_0 of the condition branch's down blocks writes the Haar transform: no DWT pass
    ("side_xp", "prologue"),        # res_conv also writes conv20's prologue output: no prologue pass
])
def test_config2_batch32_epilogue_fusions_switched_off(switch, what):
    """Every host-side fusion of the round-4 inference graph (ops.config: fuse_res_conv, emit_ll, emit_dwt, side_xp) replaces
    launches, never an operator's result: the batch-32 forward of the benchmark workload with one of them switched off equals the
    default graph's (another summation order where a folded convolution is involved: 1e-5); the folded 1x1x1 convolutions are
    launches of their own again, the other switches leave the convolution launches as they are (ops.COUNTS)."""
    import collections
    from tmdiff_amd import ops
    net = _hip_net(FULL)
    d = {k: cu(v) for k, v in case_inputs(23, 32, 8, 64).items()}
    t = torch.arange(1, 33, dtype=torch.float32).reshape(32, 1).cuda() * 29

    def run():
        ops.COUNTS = collections.Counter()
        try:
            with torch.no_grad():
                y = net(d["x_t"], t, d["PAN"], d["MS"], "WV3")
            return y, ops.COUNTS
        finally:
            ops.COUNTS = None

    y_on, c_on = run()
    with ops.config.override(**{switch: False}):
        y_off, c_off = run()
    assert_close(y_off, y_on, 1e-5, 2e-6, f"{switch} off vs on")
    k1 = lambda c: sum(v for k, v in c.items() if k.endswith("_k1"))
    print(switch, "on:", dict(c_on), "off:", dict(c_off))
    if what == "k1":        # the folded convolutions are launches of their own again
        assert k1(c_off) > k1(c_on)
    else:                   # (the Haar / prologue passes these switches remove are not convolution launches: the same convolutions run)
        assert c_off == c_on


# ---- configs[2]: the WorldView-3 network of config/general.json ---------------------------------------------------------
@pytest.fixture(scope="module")
def wide_net():
    return _hip_net(WIDE)


@pytest.mark.parametrize("key,seed,size,t", [("y64", 3408, 64, torch.tensor([[612]])), ("y256", 3409, 256, torch.tensor([431.7]))])
def test_config3_wide_network_vs_reference(golden, wide_net, key, seed, size, t):
    g = golden("unet_c3")
    d = {k: cu(v) for k, v in case_inputs(seed, 1, 8, size).items()}
    with torch.no_grad():
        y32 = wide_net(d["x_t"], t.cuda(), d["PAN"], d["MS"], "WV3").cpu()
        wide_net.set_compute_dtype("bf16")
        try:
            y16 = wide_net(d["x_t"], t.cuda(), d["PAN"], d["MS"], "WV3").cpu()
        finally:
            wide_net.set_compute_dtype("fp32")
    m32, l32 = rel_err(y32, g[key])
    m16, l16 = rel_err(y16, g[key])
    print(f"ch 64-512, 8x{size}x{size} vs reference: fp32 max-rel {m32:.2e} rel-L2 {l32:.2e}; bf16 max-rel {m16:.2e} rel-L2 {l16:.2e}")
    assert m32 <= 1e-4 and l32 <= 1e-5            # SURVEY 8(d) fp32 single-forward tolerance
    assert l16 <= 1e-2                            # SURVEY 8(d) bf16 tolerance (rel-L2)


def test_config3_dpmsolver20_bf16_vs_oracle(wide_net):
    """DPM-Solver++ 20 steps (21 NFE) at ch 64-512 in the bf16 compute mode against the fp32 CPU oracle with shared
    noise; SURVEY 8(d): PSNR >= 35 dB (bf16), >= 60 dB / 2e-3 (fp32)."""
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.util import psnr
    ref = U.fill_weights_(U.WavBESTRef(channels=WIDE)).eval()
    d = case_inputs(3410, 1, 8, 32)
    noise = randn(3411, 1, 8, 32, 32)
    ora = GeneralDiffusionRef(ref, "l1", noise_fn=lambda like: noise)
    ora.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    with torch.no_grad():
        want = ora.sample_by_dpmsolver(d, "WV3", steps=20)
    diff = GeneralDiffusion(wide_net, "l1", noise_fn=lambda like: noise).cuda()
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    dc = {k: cu(v) for k, v in d.items()}
    got32 = diff.sample_by_dpmsolver(dc, "WV3", steps=20).cpu()
    assert diff.last_solver.nfe == 21
    wide_net.set_compute_dtype("bf16")
    try:
        got16 = diff.sample_by_dpmsolver(dc, "WV3", steps=20).cpu()
    finally:
        wide_net.set_compute_dtype("fp32")
    p32, p16 = psnr(got32, want), psnr(got16, want)
    print(f"config 3, 21 NFE: PSNR vs oracle fp32 {p32:.1f} dB, bf16 {p16:.1f} dB")
    assert p32 >= 60.0 and (got32 - want).abs().max() <= 2e-3
    assert p16 >= 35.0


# ---- configs[3]: finetune step at full width, local batch 8 ---------------------------------------------------------------
def test_config4_finetune_step_full_size():
    """Local batch 8 of 8x64x64 tiles, ch 32-256.  Size-independent properties of p_losses_dynamic (ref :349-370):
    (1) dropout off: the gradient of the batch-8 mean-squared loss is the mean of the two half-batch gradients (same
    timesteps and noise) -- checks every backward kernel at the production tile configurations (the L2 loss, because the
    L1 loss's sign() is discontinuous: one element whose residual changes sign between the B = 8 and the B = 4 forward
    moves a cancellation-dominated gradient by ~1/sqrt(#elements), seen as 1e-3 on convH_0); (2) dropout on, L1: finite
    loss and gradients, 56 gradient-free tensors, and the step is reproducible from the seeds."""
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    net = _hip_net(FULL)
    diff = GeneralDiffusion(net, "l2").cuda()
    diff.set_loss("cuda")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    d = {k: cu(v) for k, v in case_inputs(3412, 8, 8, 64).items()}
    noise = cu(randn(3413, 8, 8, 64, 64))
    times = np.random.RandomState(5).randint(1, 1001, size=8)

    def grads(sl):
        state = {"lo": sl.start}
        diff.noise_fn = lambda like: noise[sl]
        orig = np.random.randint
        np.random.randint = lambda lo, hi, size: times[sl]
        try:
            net.zero_grad()
            loss = diff({k: v[sl].contiguous() for k, v in d.items()}, "WV3")
            loss.backward()
        finally:
            np.random.randint = orig
        return float(loss), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    net.eval()
    l8, g8 = grads(slice(0, 8))
    la, ga = grads(slice(0, 4))
    lb, gb = grads(slice(4, 8))
    assert abs(l8 - 0.5 * (la + lb)) <= 1e-5 * abs(l8)
    assert len(g8) == 272 - 56
    stats = []
    for k in g8:
        m, l2 = rel_err(g8[k], 0.5 * (ga[k] + gb[k]))
        stats.append((m, l2, k, float(g8[k].abs().max())))
    stats.sort(reverse=True)
    print("batch-8 gradient vs mean of half-batch gradients, worst tensors (max-rel, rel-L2, name, max|grad|):")
    for m, l2, k, mag in stats[:8]:
        print(f"  {m:.2e} {l2:.2e} {k} {mag:.2e}")
    # (fp32 sums over 8 x 32768 positions taken in two different orders)
    assert stats[0][0] <= 2e-5 and max(s_[1] for s_ in stats) <= 2e-5, stats[:3]
    net.train()
    diff.noise_fn = None
    diff.loss_type = "l1"
    diff.set_loss("cuda")
    outs = []
    for _ in range(2):
        np.random.seed(1); torch.manual_seed(2); torch.cuda.manual_seed(2)
        net.zero_grad()
        loss = diff(d, "WV3")
        loss.backward()
        outs.append((float(loss), net.final.conv20.conv20.weight.grad.clone()))
    assert np.isfinite(outs[0][0]) and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    net.eval()
    np.random.seed(1); torch.manual_seed(2); torch.cuda.manual_seed(2)
    assert abs(outs[0][0] - float(diff(d, "WV3"))) > 1e-6        # dropout really was active


# ---- configs[4]: mixed satellites, per-sample prompts, tiled scene ----------------------------------------------------------
def test_config5_mixed_satellites_full_width():
    """4-band GF-2 / QB and 8-band WV-3 / WV-4 sub-batches (the reference never mixes band counts inside a batch,
    general_sharpening...py:45-53) with per-sample prompts at full width: one DDPM step each against the oracle."""
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    ref = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    net = _hip_net(FULL, ref)
    diff = GeneralDiffusion(net, "l1").cuda()
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    ora = GeneralDiffusionRef(ref, "l1")
    ora.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    for bands, prompts, seed in ((4, ["GF2", "QB"], 3414), (8, ["WV3", "WV4"], 3415)):
        d = case_inputs(seed, 2, bands, 64)
        noise = randn(seed + 10, 2, bands, 64, 64)
        diff.noise_fn = lambda like: noise
        dc = {k: cu(v) for k, v in d.items()}
        got = diff.p_sample(dc["x_t"], 700, condition_x=dc, prompt=prompts).cpu()
        rows = []
        for i in range(2):
            ora.noise_fn = lambda like, i=i: noise[i:i + 1]
            with torch.no_grad():
                rows.append(ora.p_sample(d["x_t"][i:i + 1], 700, condition_x={k: v[i:i + 1] for k, v in d.items()},
                                         prompt=prompts[i]))
        want = torch.cat(rows)
        m, l2 = rel_err(got, want)
        print(f"{bands}-band sub-batch {prompts}: max-rel {m:.2e} rel-L2 {l2:.2e}")
        assert m <= 1e-4 and l2 <= 1e-5


def test_config5_four_band_batch_on_the_production_kernels():
    """the 4-band (GF-2 / QB) network had only been held to the oracle on the DIRECT kernels (B = 2 stays
    below the grid threshold) -- the kernels a batch of 32 four-band tiles runs on (conv3d_wf<1,16,16> and its composed-LL
    mode at N = 4) were checked against the direct kernels only.  Here the production family is forced onto a small
    four-band batch at full width: one forward and one DDPM step against the oracle, launch counts asserted
    (reference: general_sharpening_joint_random_batch_finetune.py:45-53, Hyper_unet_general.py:388-396)."""
    import collections
    from tmdiff_amd import ops
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    ref = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    net = _hip_net(FULL, ref)
    d = case_inputs(3430, 2, 4, 64)
    t = torch.tensor([[650], [12]])
    prompts = ["GF2", "QB"]
    with torch.no_grad():
        want = torch.cat([ref(d["x_t"][i:i + 1], t[i:i + 1], d["PAN"][i:i + 1], d["MS"][i:i + 1], prompts[i]) for i in range(2)])
    counts = collections.Counter()
    with ops.config.override(wino_min_blocks=1):
        ops.COUNTS = counts
        try:
            got = net(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), prompts).cpu()
            fwd_counts = dict(counts)
            noise = randn(3431, 2, 4, 64, 64)
            diff = GeneralDiffusion(net, "l1", noise_fn=lambda like: noise).cuda()
            diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
            dc = {k: cu(v) for k, v in d.items()}
            step = diff.p_sample(dc["x_t"], 700, condition_x=dc, prompt=prompts).cpu()
        finally:
            ops.COUNTS = None
    m, l2 = rel_err(got, want)
    print(f"4-band batch on the production kernels: max-rel {m:.2e} rel-L2 {l2:.2e}; launches {fwd_counts}")
    assert m <= 1e-4 and l2 <= 1e-5
    # 4 bands at 64 / 32 / 16 columns: every stride-1 3x3x3 convolution there is a conv3d_wf launch (16 x 16 tiles, one band
    # tile), the three Conv_0 + LL pairs of the main branch run in its composed-LL mode; the 8-column level (4 bands x 8 x 8
    # planes fill a quarter of a tile) stays on the other kernels
    assert fwd_counts.get("conv3d_wf_fwd", 0) >= 36 and fwd_counts.get("conv3d_wfll_fwd", 0) >= 2, fwd_counts
    ora = GeneralDiffusionRef(ref, "l1")
    ora.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    rows = []
    for i in range(2):
        ora.noise_fn = lambda like, i=i: noise[i:i + 1]
        with torch.no_grad():
            rows.append(ora.p_sample(d["x_t"][i:i + 1], 700, condition_x={k: v[i:i + 1] for k, v in d.items()}, prompt=prompts[i]))
    assert_close(step, torch.cat(rows), 1e-4, 1e-5, "DDPM step, 4-band batch, production kernels")


def test_config5_tiled_512_scene():
    """A 512x512 4-band scene cut into 64 tiles of 64x64 (config 5's tiling) sampled in batches of 32 at full width with a
    short DPM-Solver run: stitched shape, finiteness, and tile (3, 5) alone with the noise it saw gives the same pixels."""
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.tiling import sample_tiled, split_tiles
    net = _hip_net(FULL)
    diff = GeneralDiffusion(net, "l1", noise_fn=lambda like: torch.randn(like.shape)).cuda()
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    d = case_inputs(3416, 1, 4, 512)
    scene = {"MS": cu(d["MS"]), "PAN": cu(d["PAN"])}
    torch.manual_seed(6)
    fused = sample_tiled(diff, scene, "GF2", tile=64, method="dpmsolver", steps=3, max_batch=32)
    assert fused.shape == (1, 4, 512, 512) and torch.isfinite(fused).all()
    idx = 3 * 8 + 5                                        # row 3, col 5 -> batch 0 (tiles 0..31), slot 29
    torch.manual_seed(6)
    first = torch.randn(32, 4, 64, 64)
    tiles = {"MS": split_tiles(scene["MS"], 64, 64)[idx:idx + 1], "PAN": split_tiles(scene["PAN"], 64, 64)[idx:idx + 1]}
    tiles["Res"] = torch.zeros_like(tiles["MS"])
    diff.noise_fn = lambda like: first[idx:idx + 1]
    alone = diff.sample_by_dpmsolver(tiles, "GF2", steps=3)
    assert (alone[0] - fused[0, :, 192:256, 320:384]).abs().max() <= 1e-4


@pytest.mark.parametrize("channels,b,c,h,w", [([4, 8, 16, 32], 3, 4, 24, 40), (FULL, 1, 8, 24, 24), (FULL, 2, 4, 8, 40)])
def test_ragged_tile_sizes_vs_oracle(channels, b, c, h, w):
    """Planes that are not whole numbers of the kernels' 8x8 / 8x16 boxes at some level (24 -> 12 -> 6 -> 3; 40 -> 20 -> 10 -> 5),
    4- and 8-band inputs: every bounds path of the staged / fused / split-K / 1x1x1 / wavelet kernels against the oracle,
    forward (inference path) and one DDPM step."""
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    ref = U.fill_weights_(U.WavBESTRef(channels=channels)).eval()
    net = _hip_net(channels, ref)
    d = case_inputs(4000 + h + w, b, c, h, w)
    t = torch.arange(1, b + 1).reshape(b, 1) * 211
    with torch.no_grad():
        want = ref(d["x_t"], t, d["PAN"], d["MS"], "GF2" if c == 4 else "WV3")
        got = net(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), "GF2" if c == 4 else "WV3").cpu()
    m, l2 = rel_err(got, want)
    print(f"ragged {channels[0]}ch B={b} {c}x{h}x{w}: max-rel {m:.2e} rel-L2 {l2:.2e}")
    assert m <= 1e-4 and l2 <= 1e-5
    noise = randn(4100, b, c, h, w)
    ora = GeneralDiffusionRef(ref, "l1", noise_fn=lambda like: noise)
    ora.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 100}, "cpu")
    diff = GeneralDiffusion(net, "l1", noise_fn=lambda like: noise).cuda()
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 100}, "cuda")
    dc = {k: cu(v) for k, v in d.items()}
    with torch.no_grad():
        y_ref = ora.p_sample(d["x_t"], 57, condition_x=d, prompt="QB")
    y = diff.p_sample(dc["x_t"], 57, condition_x=dc, prompt="QB").cpu()
    assert_close(y, y_ref, 1e-4, 1e-5, "DDPM step on a ragged tile")


_C4_ORACLE = {}


def _c4_inputs():
    B = 8
    return case_inputs(3420, B, 8, 64), randn(3421, B, 8, 64, 64), np.random.RandomState(7).randint(1, 1001, size=B)


def _c4_oracle(d, noise):
    """The oracle's side of the full-size finetune step, once for all tests that use it (CPU autograd of a batch of 8 at
    full width; np.random.randint must already be patched to the shared timesteps)."""
    if _C4_ORACLE:
        return _C4_ORACLE
    ref_net = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    for loss_type in ("l1", "l2"):
        ora = GeneralDiffusionRef(ref_net, loss_type, noise_fn=lambda like: noise)
        ora.set_loss("cpu")
        ora.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
        ref_net.zero_grad()
        lo = ora(d, "WV3")
        if loss_type == "l2":
            lo.backward()
        _C4_ORACLE[loss_type] = float(lo)
    _C4_ORACLE["grads"] = {k: (None if p.grad is None else p.grad.clone()) for k, p in ref_net.named_parameters()}
    return _C4_ORACLE


def test_config4_finetune_step_captured_in_a_hip_graph_vs_oracle_autograd():
    """the finetune forward + backward RECORDED INTO A HIP GRAPH and replayed (tmdiff_amd.model.CapturedStep
    records exactly this call; reference model.py:40-47, diffusion_general.py:349-370): local batch 8 of 8x64x64 tiles at
    full width, timesteps / noise in fixed device tensors, the L2 loss and all 216 gradients of the REPLAY against the
    oracle's CPU autograd -- and a second replay on new inputs in the same tensors against an eager evaluation."""
    from tmdiff_amd import ops
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    d, noise, times = _c4_inputs()
    orig = np.random.randint
    np.random.randint = lambda lo, hi, size: times
    try:
        want = _c4_oracle(d, noise)
    finally:
        np.random.randint = orig
    net = _hip_net(FULL)
    diff = GeneralDiffusion(net, "l2").cuda()
    diff.set_loss("cuda")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    static = {k: cu(v) for k, v in d.items() if k in ("Res", "PAN", "MS")}
    t_dev = torch.from_numpy(times).view(8, 1).cuda()
    a_dev = torch.tensor(diff.sqrt_alphas_cumprod_prev[times], dtype=torch.float32).cuda()
    nz = cu(noise)
    for _ in range(2):                                     # eager warm-up: packed-weight sets, workspaces, allocator
        net.zero_grad(set_to_none=True)
        diff.p_losses_with(static, "WV3", t_dev, a_dev, nz).backward()
    net.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = diff.p_losses_with(static, "WV3", t_dev, a_dev, nz)
        loss.backward()
    for p in net.parameters():                             # recording ran nothing: poison what the replay must produce
        if p.grad is not None:
            p.grad.fill_(float("nan"))
    g.replay()
    assert abs(float(loss) - want["l2"]) <= 1e-5 * abs(want["l2"]), (float(loss), want["l2"])
    n = 0
    for k, p in net.named_parameters():
        if p.grad is None:
            assert want["grads"][k] is None, k
            continue
        n += 1
        m, l2 = rel_err(p.grad, want["grads"][k])
        assert l2 <= 1e-4, (k, m, l2)
    assert n == 272 - 56
    # new inputs through the same static tensors: the replay equals an eager call (same launches, same arithmetic)
    d2 = case_inputs(3440, 8, 8, 64)
    for k in static:
        static[k].copy_(cu(d2[k]))
    t_dev.copy_(torch.from_numpy(times[::-1].copy()).view(8, 1))
    a_dev.copy_(torch.tensor(diff.sqrt_alphas_cumprod_prev[times[::-1]], dtype=torch.float32))
    nz.copy_(cu(randn(3441, 8, 8, 64, 64)))
    g.replay()
    got_loss, got = float(loss), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    eager = diff.p_losses_with(static, "WV3", t_dev, a_dev, nz)
    eager.backward()
    assert got_loss == float(eager)
    off = []
    for k, p in net.named_parameters():
        if p.grad is not None and not torch.equal(got[k], p.grad):
            off.append((float((got[k] - p.grad).abs().max() / p.grad.abs().max().clamp_min(1e-30)), k))
    print("replay vs eager, tensors that are not bit-identical (max-rel difference, name):", sorted(off, reverse=True)[:10])
    # (same launches, same arithmetic: bit-identical except where a reduction's order depends on the launch stream's
    #  workspace -- nothing may differ beyond fp32 rounding of a sum)
    assert all(m <= 2e-6 for m, _ in off), sorted(off, reverse=True)[:5]


@pytest.mark.parametrize("switches", [{}, {"wgrad_wino": False}, {"train_ll_wino": False}, {"wf_pair": False}],
                         ids=["defaults", "wgrad_direct", "train_ll_direct", "no_pair_mode"])
def test_config4_finetune_step_vs_oracle_autograd(switches):
    """(also with each result-changing default switched OFF -- the direct weight gradient, the finetune
    forward's Conv_0 + LL on conv3d_ll, the 8x8 level without pair mode: the off paths are product code too.)
    VERDICT r2 weak #3: the full-size training check above is a property (batch-8 gradient == mean of half-batch
    gradients) and had to use the L2 loss.  This is the direct measurement: local batch 8 of 8x64x64 tiles, ch 32-256,
    dropout off, the SAME timesteps and noise on both sides -- the loss under the config's L1 (1e-5) and, under L2 (whose
    gradient is continuous in the residual), named parameter gradients ELEMENTWISE against the oracle's CPU autograd
    (reference Hyper_unet_general.py:334-414, diffusion_general.py:349-370).  At this size the forward and data-gradient
    convolutions run on the Winograd kernels, Conv_0 + LL on the composed kernel."""
    import collections
    from tmdiff_amd import ops
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    d, noise, times = _c4_inputs()
    names = ["up1.up1.convH_0.0.weight", "down1.down.Conv_0.weight", "final.conv24.weight", "conv2.conv21.weight",
             "down2_1.conv20.conv21.weight", "down3.down.Conv_1.weight", "middle1.conv20.weight", "up2.conv20.conv20.weight",
             "up3.up1.Conv_2.weight", "final.conv21.conv20.bias", "down1.conv20.dense1.dense.weight", "embed.2.weight",
             "final.dense2.dense.weight"]
    orig = np.random.randint
    np.random.randint = lambda lo, hi, size: times
    try:
        _c4_oracle(d, noise)
        net = _hip_net(FULL)
        net.eval()
        res = {}
        with ops.config.override(**switches):
            for loss_type in ("l1", "l2") if not switches else ("l2",):
                diff = GeneralDiffusion(net, loss_type, noise_fn=lambda like: cu(noise)).cuda()
                diff.set_loss("cuda")
                diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
                net.zero_grad()
                counts = ops.COUNTS = collections.Counter()
                try:
                    lh = diff({k: cu(v) for k, v in d.items()}, "WV3")
                    if loss_type == "l2":
                        lh.backward()
                finally:
                    ops.COUNTS = None
                res[loss_type] = (_C4_ORACLE[loss_type], float(lh), dict(counts))
    finally:
        np.random.randint = orig
    for lt, (lo, lh, _) in res.items():
        assert abs(lo - lh) <= 1e-5 * abs(lo), (lt, lo, lh)
    cnt = res["l2"][2]
    if ops.config.winograd and ops.config.ll_compose:   # (experiment switches off)
        assert (cnt.get("conv3d_wf_fwd", 0) + cnt.get("conv3d_wino4_fwd", 0) + cnt.get("conv3d_wino2_fwd", 0) >= 40 and
                cnt.get("conv3d_ll_fwd", 0) + cnt.get("conv3d_wfll_fwd", 0) == 3), cnt
    # the switch really selected the other path
    if switches.get("wgrad_wino") is False:
        assert cnt.get("conv3d_wgrad_wino", 0) == 0 and cnt.get("conv3d_wgrad", 0) > 40, cnt
    elif not switches:
        assert cnt.get("conv3d_wgrad_wino", 0) > 40, cnt
    if switches.get("train_ll_wino") is False:
        assert cnt.get("conv3d_wfll_fwd", 0) == 0 and cnt.get("conv3d_ll_fwd", 0) == 3, cnt
    elif not switches:
        assert cnt.get("conv3d_wfll_fwd", 0) == 3, cnt
    ref_g = _C4_ORACLE["grads"]
    hip_g = dict(net.named_parameters())
    report = []
    for k in names:
        m, l2 = rel_err(hip_g[k].grad, ref_g[k])
        report.append((m, l2, k))
    print("full-size L2 gradients vs the oracle's autograd (max-rel, rel-L2):")
    for m, l2, k in sorted(report, reverse=True):
        print(f"  {m:.2e} {l2:.2e} {k}")
    assert max(r[0] for r in report) <= 3e-5 and max(r[1] for r in report) <= 3e-5, sorted(report, reverse=True)[:3]
    # and every other gradient by its sums (all 216)
    n = 0
    for k, p in hip_g.items():
        if p.grad is None:
            assert ref_g[k] is None, k
            continue
        n += 1
        m, l2 = rel_err(p.grad, ref_g[k])
        assert l2 <= 1e-4, (k, m, l2)
    assert n == 272 - 56


def test_round3_kernels_at_full_size_properties():
    """The round-3 kernels at BASELINE sizes, through properties that need no CPU reference: the Winograd-domain weight
    gradient at the finetune batch (8 tiles of 8x64x64) is linear in the gradient, deterministic and agrees with the direct
    kernel; the composed Conv_0 + LL convolution with Winograd on top at the benchmark batch (32 tiles) agrees with conv3d_ll,
    its producer's space-to-depth output with the plain one, and an image gives the same bits wherever it sits in the batch
    (pair mode at the 8x8 level included)."""
    from tmdiff_amd import ops
    torch.manual_seed(31)
    # ---- weight gradient, 64 -> 64 at 8 x 8x64x64
    x = torch.randn(8, 64, 8, 64, 64, device="cuda")
    g1, g2 = torch.randn(8, 64, 8, 64, 64, device="cuda"), torch.randn(8, 64, 8, 64, 64, device="cuda")
    wshape = (64, 64, 3, 3, 3)
    dw = lambda g: ops.conv3d_wgrad(ops.make_conv_desc([x], 0, 64, 3, g), g, wshape)
    a, b, ab = dw(g1), dw(g2), dw(g1 + 0.5 * g2)
    assert torch.equal(a, dw(g1))
    scale = float(ab.abs().max())
    assert float((ab - (a + 0.5 * b)).abs().max()) <= 1e-5 * scale                 # linearity (sums of 262 144 fp32 products)
    keep, ops.config.wgrad_wino = ops.config.wgrad_wino, False
    try:
        direct = dw(g1)
    finally:
        ops.config.wgrad_wino = keep
    assert float((a - direct).norm() / direct.norm()) <= 5e-6                      # both accumulate 262 144 products in fp32
    del g1, g2, a, b, ab, direct
    # ---- Conv_0 + LL, 64 -> 64 at 32 x 8x64x64: producer (space-to-depth second output) + composed convolution
    xin = torch.randn(32, 64, 8, 64, 64, device="cuda")
    w21 = torch.randn(64, 64, 3, 3, 3, device="cuda") / (64 * 27) ** 0.5
    w0, bias = torch.randn(64, 64, 3, 3, 3, device="cuda") / (64 * 27) ** 0.5, torch.randn(64, device="cuda")
    wp = ops.pack_conv_weight_wino(w21, groups=1, mode=2, planes=6)
    assert ops.wf_route(32, 64, 64, 8, 64, 64) == (True, 1) and ops.wfll_route(32, 64, 64, 8, 64, 64)
    y, plain = ops.conv3d_wf([xin], wp, 64, emit=dict(act=True))
    y_, s2d = ops.conv3d_wf([xin], wp, 64, emit=dict(act=True, s2d=True))
    assert torch.equal(y, y_)
    assert torch.equal(s2d, plain.view(32, 64, 8, 32, 2, 32, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(32, 256, 8, 32, 32))
    out = ops.conv3d_wf_ll(s2d, ops.pack_conv_weight_wfll(w0, 0.5), 64, 0.5, bias=bias)
    old = ops.conv3d_ll(plain, ops.pack_conv_weight_ll(w0, 0.5), 64, 0.5, bias=bias)
    assert float((out - old).norm() / old.norm()) <= 2e-6
    # ---- batch position: image 5 alone in a batch of two identical images, and at the 8x8 level in pair mode
    for shape in ((8, 64, 64), (8, 8, 8)):
        c = 64 if shape[1] == 64 else 256
        xb = torch.randn(32, c, *shape, device="cuda")
        wq = ops.pack_conv_weight_wino(torch.randn(c, c, 3, 3, 3, device="cuda") / (c * 27) ** 0.5, groups=1, mode=2, planes=6)
        full = ops.conv3d_wf([xb], wq, c)
        shuffled = ops.conv3d_wf([xb.flip(0).contiguous()], wq, c).flip(0)
        assert torch.equal(full, shuffled), shape
