"""GPU parity of the finetune path: backward kernels against PyTorch CPU autograd, the whole training loss and
its parameter gradients against the reference fixture (tests/golden/train.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import unet_ref as U
from oracle.diffusion_ref import GeneralDiffusionRef
from oracle.make_golden import TINY, case_inputs, randn

pytestmark = pytest.mark.gpu


def cu(t):
    return t.detach().cuda().contiguous()


CASES = [  # B, Cin, Cout, N, H, W, k, groups
    (2, 8, 8, 4, 8, 8, 3, 1), (1, 5, 7, 4, 6, 6, 3, 1), (2, 32, 64, 8, 16, 16, 3, 1), (3, 4, 12, 8, 2, 2, 3, 1),
    (2, 12, 8, 3, 10, 12, 3, 1), (2, 16, 32, 8, 16, 16, 1, 1), (1, 5, 3, 4, 6, 6, 1, 1), (2, 24, 12, 8, 8, 8, 3, 3),
    (1, 40, 96, 4, 8, 8, 3, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_fused_conv_backward(case):
    from tmdiff_amd import autograd as A
    b, ci, co, n, h, w, k, g = case
    nseg = 3 if (ci % 3 == 0 and g == 3) else (2 if ci % 2 == 0 and g == 1 else 1)
    cs = [ci // nseg] * nseg
    segs = [randn(10 + i, b, c, n, h, w).requires_grad_(True) for i, c in enumerate(cs)]
    wt = (randn(2, co, ci // g, k, k, k) / (ci // g * k ** 3) ** 0.5).requires_grad_(True)
    bias = randn(3, co).requires_grad_(True)
    shift = randn(4, b, ci).requires_grad_(True)
    scale = (1 + 0.3 * randn(5, b, ci)).requires_grad_(True)
    res = randn(6, b, co, n, h, w).requires_grad_(True)
    mask = (torch.rand(b, ci, n, h, w, generator=torch.Generator().manual_seed(7)) > 0.2).float() / 0.8
    gy = randn(8, b, co, n, h, w)
    xp = U.silu(torch.cat(segs, 1) + shift[:, :, None, None, None]) * scale[:, :, None, None, None] * mask
    y = (F.conv3d(xp, wt, None, 1, k // 2, 1, g) + 2.0 * bias[None, :, None, None, None] + res) * 0.5
    y.backward(gy)
    dsegs = [cu(s).requires_grad_(True) for s in segs]
    dw, db, dsh, dsc, dres = (cu(t).requires_grad_(True) for t in (wt, bias, shift, scale, res))
    yd = A.conv3d(dsegs, dw, db, bias_scale=2.0, shift=dsh, scale=dsc, act=True, mask=cu(mask), residual=dres,
                  groups=g, out_scale=0.5)
    assert_close(yd.detach().cpu(), y.detach(), 1e-5, 1e-5, "fused conv forward")
    yd.backward(cu(gy))
    for name, got, want in [("dw", dw, wt), ("dbias", db, bias), ("dshift", dsh, shift), ("dscale", dsc, scale),
                            ("dres", dres, res)] + [(f"dx{i}", a, s) for i, (a, s) in enumerate(zip(dsegs, segs))]:
        assert_close(got.grad.cpu(), want.grad, 3e-5, 3e-5, f"{name} {case}")


def test_small_op_backward():
    from tmdiff_amd import autograd as A
    # Haar DWT (all bands / LL only) and the paired IDWT
    x = randn(1, 2, 3, 4, 8, 8).requires_grad_(True)
    from oracle.haar_ref import haar_dwt2d, haar_idwt2d
    bands = haar_dwt2d(x)
    gs = [randn(2 + i, *bands[0].shape) for i in range(4)]
    torch.autograd.backward([0.5 * bands[0], *bands[1:]], gs)
    xd = cu(x).requires_grad_(True)
    bd = A.haar_dwt2d(xd, want_high=True, ll_scale=0.5)
    torch.autograd.backward(bd, [cu(g) for g in gs])
    assert_close(xd.grad.cpu(), x.grad, 1e-6, 1e-6, "dwt backward")
    x.grad = None
    (0.5 * haar_dwt2d(x)[0]).backward(gs[0])
    xd = cu(x).requires_grad_(True)
    A.haar_dwt2d(xd, want_high=False, ll_scale=0.5)[0].backward(cu(gs[0]))
    assert_close(xd.grad.cpu(), x.grad, 1e-6, 1e-6, "LL-only dwt backward")
    b, c, n, h, w = 2, 3, 4, 4, 4
    hb, xb = randn(10, b, c, n, h, w).requires_grad_(True), randn(11, b, c, n, h, w).requires_grad_(True)
    st = randn(12, b, 3 * c, n, h, w).requires_grad_(True)
    fold = lambda v: v.reshape(b, -1, h, w)
    hi = [fold(st[:, i * c:(i + 1) * c]) for i in range(3)]
    o1 = haar_idwt2d(2.0 * fold(hb), *hi).reshape(b, c, n, 2 * h, 2 * w)
    o2 = haar_idwt2d(2.0 * fold(xb), *hi).reshape(b, c, n, 2 * h, 2 * w)
    g1, g2 = randn(13, *o1.shape), randn(14, *o2.shape)
    torch.autograd.backward([o1, o2], [g1, g2])
    hd, xd, sd = (cu(t).requires_grad_(True) for t in (hb, xb, st))
    p1, p2 = A.haar_idwt2d_pair(hd, xd, sd, in_scale=2.0)
    assert_close(p1.detach().cpu(), o1.detach(), 1e-6, 1e-6, "paired idwt")
    torch.autograd.backward([p1, p2], [cu(g1), cu(g2)])
    for name, got, want in (("dh", hd, hb), ("dx", xd, xb), ("dbands", sd, st)):
        assert_close(got.grad.cpu(), want.grad, 1e-6, 1e-6, f"paired idwt {name}")
    # stem, head, linear
    d = case_inputs(5, 2, 8, 16)
    wt, bias = randn(20, 6, 1, 1, 1, 1).requires_grad_(True), randn(21, 6).requires_grad_(True)
    cond = (d["PAN"].repeat(1, 8, 1, 1) - d["MS"]).unsqueeze(1)
    y = U.silu(F.conv3d(cond, wt, bias))
    gy = randn(22, *y.shape)
    y.backward(gy)
    wd, bd_ = cu(wt).requires_grad_(True), cu(bias).requires_grad_(True)
    yd = A.stem(wd, bd_, pan=cu(d["PAN"]), ms=cu(d["MS"]))
    yd.backward(cu(gy))
    assert_close(wd.grad.cpu(), wt.grad, 1e-5, 1e-5, "stem dw")
    assert_close(bd_.grad.cpu(), bias.grad, 1e-5, 1e-5, "stem dbias")
    x5 = randn(23, 2, 6, 8, 16, 16).requires_grad_(True)
    w24 = randn(24, 1, 6, 1, 1, 1).requires_grad_(True)
    sc = (1 + 0.2 * randn(25, 2, 6)).requires_grad_(True)
    y = U.modconv3d(U.silu(x5), w24, sc[:, :, None, None], 0).squeeze(1)
    gy = randn(26, *y.shape)
    y.backward(gy)
    xd, wd, sd = (cu(t).requires_grad_(True) for t in (x5, w24, sc))
    A.head(xd, wd, sd).backward(cu(gy))
    for name, got, want in (("dx", xd, x5), ("dw", wd, w24), ("dscale", sd, sc)):
        assert_close(got.grad.cpu(), want.grad, 1e-5, 1e-5, f"head {name}")
    for act in (False, True):
        x, wl, bl = randn(30, 5, 128).requires_grad_(True), (randn(31, 300, 128) / 11).requires_grad_(True), randn(32, 300).requires_grad_(True)
        y = F.linear(x, wl, bl)
        y = U.silu(y) if act else y
        gy = randn(33, *y.shape)
        y.backward(gy)
        xd, wd, bd_ = (cu(t).requires_grad_(True) for t in (x, wl, bl))
        A.linear(xd, wd, bd_, act=act).backward(cu(gy))
        for name, got, want in (("dx", xd, x), ("dw", wd, wl), ("db", bd_, bl)):
            assert_close(got.grad.cpu(), want.grad, 1e-5, 1e-5, f"linear act={act} {name}")


def test_training_loss_and_gradients_vs_reference(golden):
    """p_losses_dynamic end to end (dropout off via eval(), as in the fixture): loss value and per-parameter
    gradient checksums against the real reference; 56 parameters must stay gradient-free."""
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    g = golden("train")
    ref_net = U.fill_weights_(U.WavBESTRef(channels=TINY)).eval()
    net = WavBEST(channels=TINY)
    net.load_state_dict(ref_net.state_dict())
    net = net.cuda().eval()
    diff = GeneralDiffusion(net, "l1", noise_fn=lambda like: torch.randn(like.shape)).cuda()
    diff.set_loss("cuda")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    for b in (3, 1):
        d = {k: cu(v) for k, v in case_inputs(132 + b, b, 8, 16).items()}
        np.random.seed(5)
        torch.manual_seed(6)
        net.zero_grad()
        loss = diff(d, "WV3")
        loss.backward()
        assert abs(float(loss) - float(g[f"loss_b{b}"])) <= 1e-4 * abs(float(g[f"loss_b{b}"]))
        ref = g[f"gsum_b{b}"]
        n_nograd = 0
        for i, (name, p) in enumerate(net.named_parameters()):
            if p.grad is None:
                assert np.isnan(ref[i, 0]), name
                n_nograd += 1
                continue
            assert not np.isnan(ref[i, 0]), f"{name}: gradient here but none in the reference"
            got = np.array([float(p.grad.sum()), float(p.grad.abs().sum())])
            assert np.allclose(got, ref[i], rtol=2e-3, atol=2e-3 * max(ref[i, 1], 1e-6)), (name, got, ref[i])
        assert n_nograd == 56


def test_training_step_with_dropout_runs_and_descends():
    """train() mode: dropout masks active; a few AdamW steps on one batch reduce the loss."""
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.util import fill_weights_
    torch.manual_seed(0)
    net = fill_weights_(WavBEST(channels=TINY)).cuda().train()
    diff = GeneralDiffusion(net, "l1").cuda()
    diff.set_loss("cuda")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    opt = torch.optim.AdamW([p for p in net.parameters()], lr=1e-3, weight_decay=1e-4)
    d = {k: cu(v) for k, v in case_inputs(1, 4, 8, 16).items()}
    losses = []
    for it in range(8):
        np.random.seed(it % 2)
        opt.zero_grad()
        loss = diff(d, "WV3").sum()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    assert min(losses[4:]) < losses[0]


def test_packed_weights_follow_a_deep_copy_and_ignore_foreign_weights():
    """(1) copy.deepcopy of a network that has already trained must not keep the source's Winograd pack -- its
    device table held the SOURCE network's raw addresses, and the copy's first refresh re-packed the source's memory (a
    use-after-free once the source is gone).  forward_train, deepcopy, delete the original, forward_train on the copy: same
    loss and gradients as a fresh network with the same weights.  (2) a convolution on a weight that does not belong to the
    network whose pack is current is packed on the spot and never registered."""
    import copy
    import gc
    from tmdiff_amd import autograd as A
    from tmdiff_amd import ops
    from tmdiff_amd.Hyper_unet_general import WavBEST
    FULLC = [32, 64, 128, 256]
    d = case_inputs(991, 8, 8, 32)          # B = 8 at 32x32: the 32x32 / 16x16 levels run conv3d_wf (multi-tensor pack in use)
    t = torch.arange(1, 9).reshape(8, 1) * 100

    def step(net):
        net.zero_grad()
        out = net(cu(d["x_t"]), t.cuda(), cu(d["PAN"]), cu(d["MS"]), "WV3")
        out.square().mean().backward()
        return float(out.square().mean()), net.down1.conv20.conv21.weight.grad.clone(), net.up2.up1.Conv_1.weight.grad.clone()

    net = U.fill_weights_(WavBEST(channels=FULLC)).cuda().train()
    net.requires_grad_(True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    want = step(net)
    pack = net.__dict__["_train_pack_wino"]
    assert len(pack.items) >= 20                              # the pack is really in use at this size
    twin = copy.deepcopy(net)
    assert "_train_pack_wino" not in twin.__dict__ and "_train_pack" not in twin.__dict__ and twin._prep is None
    del net, pack
    ops.PACKED = ops.WINO_PACKED = None
    gc.collect()
    torch.cuda.empty_cache()
    junk = torch.full((64 << 20,), float("nan"), device="cuda")       # whatever reuses the freed memory is poison
    got = step(twin)
    del junk
    assert got[0] == want[0] and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    tpack = twin.__dict__["_train_pack_wino"].refresh()      # (the step after the learning one builds the device table)
    n_items = len(tpack.items)
    key0 = tpack._table_key
    assert n_items >= 20 and key0 is not None
    # (2) a foreign weight while the twin's pack is the current one
    w = torch.randn(32, 32, 3, 3, 3, device="cuda", requires_grad=True)
    x = torch.randn(8, 32, 8, 32, 32, device="cuda")
    with ops.config.override(wino_min_blocks=1):
        y = A.conv3d([x], w, None)
        y.sum().backward()
    assert len(tpack.items) == n_items and tpack._table_key == key0
    tpack.refresh()
    assert len(tpack.items) == n_items and tpack._table_key == key0      # nothing to rebuild, nothing kept alive
    got2 = step(twin)
    assert got2[0] == want[0] and torch.equal(got2[1], want[1])


def test_fused_adamw_is_torch_adamw():
    """tmdiff_amd.optim.FusedAdamW (one multi-tensor HIP launch, learning rate and step count read from device scalars) against
    torch.optim.AdamW (reference model.py:30-31) on the same parameters and gradients: ragged tensor sizes, a parameter
    without a gradient, a learning-rate schedule, eight steps -- and the state_dict of either continues in the other."""
    from tmdiff_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(4)
    shapes = [(64, 32, 3, 3, 3), (7,), (33, 5), (1,), (128, 128), (16385,), (3, 3)]
    base = [torch.randn(s, generator=g) for s in shapes]
    mk = lambda: [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pa, pb = mk(), mk()
    oa = torch.optim.AdamW(pa, lr=1e-3, weight_decay=1e-2)
    ob = FusedAdamW(pb, lr=torch.tensor(1e-3, device="cuda"), weight_decay=1e-2)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lambda k: 1.0 / (1 + k))
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda k: 1.0 / (1 + k))

    def steps(opt_a, opt_b, sch_a, sch_b, params_a, params_b, n, seed):
        gg = torch.Generator().manual_seed(seed)
        for it in range(n):
            for k, (x, y) in enumerate(zip(params_a, params_b)):
                if k == 6:                      # this one never gets a gradient (56 WavBEST tensors never do)
                    continue
                gr = torch.randn(x.shape, generator=gg).cuda() * (10.0 if it == 2 else 1.0)
                x.grad, y.grad = gr.clone(), gr.clone()
            opt_a.step(); opt_b.step(); sch_a.step(); sch_b.step()
        return max(float((x - y).abs().max() / x.abs().max()) for x, y in zip(params_a, params_b))

    worst = steps(oa, ob, sa, sb, pa, pb, 8, 1)
    assert worst <= 2e-6, worst
    assert torch.equal(pa[6], pb[6]) and pb[6] not in ob.state or "exp_avg" not in ob.state[pb[6]]
    assert float(ob.state[pb[0]]["step"]) == 8.0 and ob.state[pb[0]]["step"] is ob.state[pb[3]]["step"]
    # checkpoints interchange: each optimizer continues from the other's state
    pc, pd = [torch.nn.Parameter(x.detach().clone()) for x in pa], [torch.nn.Parameter(y.detach().clone()) for y in pb]
    oc = FusedAdamW(pc, lr=torch.tensor(1e-3, device="cuda"), weight_decay=1e-2)
    od = torch.optim.AdamW(pd, lr=1e-3, weight_decay=1e-2)
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    sd_b["param_groups"][0]["lr"] = float(sd_b["param_groups"][0]["lr"])
    sd_b["param_groups"][0]["initial_lr"] = float(sd_b["param_groups"][0]["initial_lr"])
    sd_a["param_groups"][0]["lr"] = torch.tensor(sd_a["param_groups"][0]["lr"], device="cuda")
    oc.load_state_dict(sd_a)
    od.load_state_dict(sd_b)
    sc = torch.optim.lr_scheduler.LambdaLR(oc, lambda k: 1.0 / (1 + k), last_epoch=7)
    sdd = torch.optim.lr_scheduler.LambdaLR(od, lambda k: 1.0 / (1 + k), last_epoch=7)
    worst2 = steps(od, oc, sdd, sc, pd, pc, 3, 2)
    assert worst2 <= 4e-6 and float(oc.state[pc[0]]["step"]) == 11.0, (worst2, oc.state[pc[0]]["step"])
    print(f"FusedAdamW vs torch.optim.AdamW: {worst:.2e} after 8 steps, {worst2:.2e} after swapping the state_dicts")


def test_captured_finetune_step_follows_the_eager_trainer():
    """tmdiff_amd.model.DDPM with train.hip_graph: after two eager warm-up steps per (prompt, batch shape) the step -- forward,
    backward, AdamW -- is ONE HIP-graph launch (reference model.py:40-47).  (1) dropout off, same timesteps and noise: the
    parameters after six steps follow the eager trainer's (capturable AdamW computes its bias corrections on the device, so
    agreement is to rounding, not to the bit); (2) dropout on: replays of the recorded launches draw fresh masks (the
    per-step seed word lives in device memory), the loss stays finite and decreases over a few steps of a fixed batch."""
    import copy
    from tmdiff_amd import ops
    from tmdiff_amd.model import DDPM
    from tmdiff_amd.util import fill_weights_, synthetic_tile_batch
    base = {"phase": "train", "gpu_ids": [0], "distributed": False, "path": {"resume": None},
            "model": {"unet": {"channel_multiplier": [32, 64, 128, 256]}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
            "train": {"optimizer": {"lr": 1e-4}, "max_iter": 1000}}
    d = synthetic_tile_batch(4100, 4, 8, 32, device=torch.device("cuda"))
    d["LR"] = d["MS"]
    gen = torch.Generator().manual_seed(9)
    noises = [torch.randn(4, 8, 32, 32, generator=gen) for _ in range(6)]

    def run(graph, dropout):
        opt = copy.deepcopy(base)
        opt["train"]["hip_graph"] = graph
        m = DDPM(opt)
        fill_weights_(m.netG.denoise_fn)
        m.netG.denoise_fn.invalidate_prepared()
        m.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "train")
        if not dropout:
            for mod in m.netG.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
        it = iter(noises)
        m.netG.noise_fn = lambda like: next(it)
        np.random.seed(77)
        torch.manual_seed(5)
        losses = []
        for _ in range(6):
            m.feed_data(dict(d))
            m.optimize_parameters("WV3")
            losses.append(float(m.get_current_log()["l_pix"]))
        return m, losses

    eager, le = run(False, False)
    graph, lg = run(True, False)
    step = next(iter(graph._captured.values()))
    assert graph.use_graph and len(graph._captured) == 1 and step.replays == 4          # 2 eager warm-up steps + 4 replays
    assert np.allclose(le, lg, rtol=2e-5), (le, lg)
    worst = 0.0
    for (k, p), (_, q) in zip(eager.netG.named_parameters(), graph.netG.named_parameters()):
        worst = max(worst, float((p - q).abs().max() / p.abs().max().clamp_min(1e-12)))
    assert worst <= 2e-4, worst                      # six AdamW steps of lr <= 6e-6 (warm-up schedule): tiny updates, tiny differences
    print(f"captured vs eager trainer after 6 steps: losses {lg}, worst relative parameter difference {worst:.2e}")
    del eager, graph
    torch.cuda.empty_cache()
    # (2) dropout on: fresh masks per replay
    m, losses = run(True, True)
    step = next(iter(m._captured.values()))
    assert all(np.isfinite(losses))
    word0 = int(ops.DROP_WORD)
    step.replay()
    step.replay()
    assert int(ops.DROP_WORD) == word0 + 2 and np.isfinite(float(step.loss))     # the graph itself bumps the seed word
    # (that the kernels add the word to their seeds: tests/test_gpu_kernels.py::test_dropout_seed_word_in_device_memory)


def test_trainer_wrapper_roundtrip(tmp_path):
    """DDPM wrapper (reference model.py API): train steps, EMA, save -> load into a fresh wrapper, test()."""
    import copy
    from tmdiff_amd.model import DDPM, EmaUpdater, create_model
    opt = {"phase": "train", "gpu_ids": [0], "distributed": False,
           "path": {"resume": None, "checkpoint": str(tmp_path)},
           "model": {"unet": {"channel_multiplier": TINY}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
           "train": {"optimizer": {"lr": 1e-3}, "max_iter": 100}}
    torch.manual_seed(1)
    m = create_model(opt)
    ema = EmaUpdater(m, copy.deepcopy(m), decay=0.9)
    m.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, "train")
    d = case_inputs(3, 2, 8, 16)
    d["LR"] = d["MS"].clone()
    for it in range(3):
        m.feed_data({k: v.clone() for k, v in d.items()})
        m.optimize_parameters("WV3")
        ema.update(it)
    assert np.isfinite(float(m.get_current_log()["l_pix"])) and m.get_current_log()["lr"] > 0
    p, pe = next(m.netG.denoise_fn.parameters()), next(ema.ema_model.netG.denoise_fn.parameters())
    assert not torch.equal(p, pe)
    m.save_network(3)
    opt2 = dict(opt, path={"resume": str(tmp_path / "I3"), "checkpoint": str(tmp_path)})
    m2 = DDPM(opt2)
    assert m2.begin_step == 3
    sd2 = m2.netG.state_dict()           # (no schedule buffers yet: they are not module state until a schedule is set)
    n_cmp = 0
    for k, a in m.netG.state_dict().items():
        if k.startswith("denoise_fn."):
            assert torch.equal(a.cpu(), sd2[k].cpu()), k
            n_cmp += 1
    assert n_cmp == 272
    m2.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, "val")
    m2.feed_data({k: v.clone() for k, v in d.items()})
    m2.test(continous=False, prompt="WV3")
    vis = m2.get_current_visuals()
    assert vis["SR"].shape[1:] == (8, 16, 16) and torch.isfinite(vis["SR"]).all()


def test_finetune_driver_end_to_end(tmp_path):
    """python -m tmdiff_amd.train on tiny npz datasets: random dataset choice, optimizer steps, validation with .mat
    output, checkpoint; then a val-phase run that resumes from that checkpoint (ref driver :56-180)."""
    import json
    import scipy.io as scio
    from tmdiff_amd import train

    def dataset(name, n, c, scale):
        g = np.random.default_rng(len(name))
        np.savez(str(tmp_path / name), gt=g.integers(0, scale, (n, c, 16, 16)), lms=g.integers(0, scale, (n, c, 16, 16)),
                 ms=g.integers(0, scale, (n, c, 4, 4)), pan=g.integers(0, scale, (n, 1, 16, 16)))
        return str(tmp_path / name)

    tr = dict(batch_size=2, num_workers=0, use_shuffle=True, data_len=-1)
    opt = {"name": "t", "phase": "train", "gpu_ids": [0],
           "path": {"log": "logs", "results": "results", "checkpoint": "checkpoint", "resume": None},
           "datasets": {"train_qb": dict(tr, dataroot=dataset("train_qb.npz", 4, 4, 2047)),
                        "train_gf2": dict(tr, dataroot=dataset("train_gf2.npz", 4, 4, 1023)),
                        "train_wv3": dict(tr, dataroot=dataset("train_wv3.npz", 6, 8, 2047)),
                        "val_GF2": dict(dataroot=dataset("test_gf2.npz", 1, 4, 1023), data_len=-1),
                        "val_WV3": dict(dataroot=dataset("test_wv3.npz", 2, 8, 2047), data_len=-1)},
           "model": {"beta_schedule": {"train": {"schedule": "cosine", "n_timestep": 20},
                                       "val": {"schedule": "cosine", "n_timestep": 4}},
                     "unet": {"channel_multiplier": [8, 16, 32, 64]}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
           "train": {"val_freq": 4, "save_checkpoint_freq": 4, "print_freq": 2, "max_iter": 4, "optimizer": {"lr": 1e-4}}}
    cfg = tmp_path / "opt.json"
    cfg.write_text(json.dumps(opt))
    root = str(tmp_path / "exp")
    assert train.main(["-c", str(cfg), "-p", "train", "--root", root]) == 4
    run = os.path.join(root, sorted(os.listdir(root))[-1])
    mats = sorted(os.listdir(os.path.join(run, "results", "WV3")))
    assert mats == ["output_mulExm_0.mat", "output_mulExm_1.mat"]
    sr = scio.loadmat(os.path.join(run, "results", "WV3", mats[0]))["sr"]
    assert sr.shape == (16, 16, 8) and np.isfinite(sr).all() and sr.min() >= 0 and sr.max() <= 2047.0
    assert scio.loadmat(os.path.join(run, "results", "GF2", "output_mulExm_0.mat"))["sr"].shape == (16, 16, 4)
    ckpt = os.path.join(run, "checkpoint", "I4")
    assert os.path.exists(ckpt + "_gen.pth") and os.path.exists(ckpt + "_opt.pth")
    # validation-only run from that checkpoint
    opt["path"]["resume"] = ckpt
    opt["model"]["beta_schedule"]["val"]["n_timestep"] = 3
    cfg.write_text(json.dumps(opt))
    scores = train.main(["-c", str(cfg), "-p", "val", "--root", root])
    assert set(scores) >= {"ssim_WV3", "sam_WV3", "ssim_GF2", "sam_GF2"} and all(np.isfinite(v) for v in scores.values())


def test_bench_train_two_ranks_share_the_gpu_over_gloo():
    """`bench.py --gpus 2 --mode train` as its own launcher: two ranks (sharing the one GPU, gloo instead of RCCL) run the
    finetune leg with the bucketed all-reduce started from backward hooks; every bucket must go out from a hook and the
    exchange-free variant must run too.  (RCCL itself needs >= 2 GPUs: the driver's multi-GPU run is its first execution.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TMDIFF_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "2",
                        "--warmup", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    ar = line["train_step"]["allreduce"]
    assert line["n_gpus"] == 2 and line["train_step"]["global_batch"] == 16
    assert ar["buckets"] >= 3 and ar["launched_from_backward_hooks"] == ar["buckets"]
    assert 100 < ar["payload_mb"] < 130                      # 216 gradient tensors of the ch 32-256 network, fp32
    assert np.isfinite(line["train_step"]["loss"]) and line["train_step"]["ms_per_step"] > 0


_RCCL_ONE_RANK = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tmdiff_amd import dist as tdist
from tmdiff_amd.Hyper_unet_general import WavBEST
from tmdiff_amd.diffusion_general import GeneralDiffusion
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch
torch.cuda.set_device(0)                           # (WORLD_SIZE=1: tdist.init_from_env would leave a single process alone)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
def run(reduce):
    torch.manual_seed(3); torch.cuda.manual_seed(3)
    import numpy as np; np.random.seed(3)
    net = fill_weights_(WavBEST(channels=(4, 8, 16, 32))).to(dev).train()
    diff = GeneralDiffusion(net, "l1").to(dev)
    diff.set_loss(dev)
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 100}, dev)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    red = tdist.GradReducer(diff, op="sum", bucket_bytes=64 << 10, min_world=1) if reduce else None
    d = synthetic_tile_batch(11, 2, 8, 16, device=dev)
    stats = []
    for step in range(3):
        loss = diff(d, "WV3").sum()
        loss.backward()
        n = red.finish() if red else 0
        launched = red.launched if red else 0
        opt.step()
        if red: red.zero_grad()
        else: opt.zero_grad(set_to_none=True)
        stats.append((float(loss), n, launched))
    return stats, [p.detach().clone() for p in net.parameters()]
s0, w0 = run(False)
s1, w1 = run(True)
same = all(torch.equal(a, b) for a, b in zip(w0, w1))
print(json.dumps({"plain": s0, "reduced": s1, "same_weights": same, "backend": dist.get_backend()}))
dist.barrier(); dist.destroy_process_group()
"""


def test_grad_reducer_over_rccl_one_rank():
    """The RCCL code path on the one GPU this box has: a one-rank `nccl` group (RCCL refuses two ranks on one device), the
    finetune step of a narrow network through GradReducer(min_world=1) -- flat buckets, all_reduce(async_op=True) started
    from the backward hooks on GPU tensors, handle.wait() before AdamW.  A one-rank SUM is the identity, so three steps
    must leave exactly the weights of the same three steps without the reducer."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["same_weights"]
    assert [s[0] for s in out["plain"]] == [s[0] for s in out["reduced"]]
    n1, launched1 = out["reduced"][1][1], out["reduced"][1][2]
    assert n1 >= 3 and launched1 == n1                       # several 64 KB buckets, every one sent from a backward hook
