#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    python oracle/make_golden.py            # writes tests/golden/*.npz

What is stored is data only: seeds/shapes of the inputs and the reference's outputs
(fp32 arrays).  Inputs and weights are re-created on the test side from the same seeds
(``case_inputs`` below, ``oracle.unet_ref.fill_weights_``), so no weight files are kept.
Each file records the torch version whose CPU kernels produced it.

The reference is imported with the stand-ins of ``oracle/ref_shims.py`` for the
third-party modules that are not installed here (pywt, ml_collections, torchvision,
cv2) and for the CLIP text encoder whose weights are not available offline.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from oracle.unet_ref import fill_weights_, synthetic_text_embeddings  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TINY = [4, 8, 16, 32]
FULL = [32, 64, 128, 256]
WIDE = [64, 128, 256, 512]


# ----------------------------------------------------------------------------------------
# shared input recipes (imported by the tests as well)
# ----------------------------------------------------------------------------------------
def case_inputs(seed, b, c, h, w=None):
    """SURVEY 8(d) synthetic tile: MS, PAN, HR ~ U[0,1); Res = HR - MS; x_t ~ N(0,1)."""
    w = h if w is None else w
    g = torch.Generator(device="cpu").manual_seed(seed)
    ms = torch.rand(b, c, h, w, generator=g)
    pan = torch.rand(b, 1, h, w, generator=g)
    hr = torch.rand(b, c, h, w, generator=g)
    x_t = torch.randn(b, c, h, w, generator=g)
    return {"MS": ms, "PAN": pan, "HR": hr, "Res": hr - ms, "x_t": x_t}


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator(device="cpu").manual_seed(seed))


def gauss_model(ns, std=0.5):
    """Closed-form optimal noise prediction for data ~ N(0, std^2 I): eps = sigma_t x / (alpha_t^2 std^2 + sigma_t^2),
    as a "noise"-type model taking the discrete model time (t_cont - 1/N) * 1000.  A contraction towards the data
    manifold, so solver outputs are well conditioned (unlike the `toy` drift, whose solution is a 1e-4 remainder of
    cancellations).  `ns` is the caller's NoiseScheduleVP (reference, oracle or build)."""
    def model(x, t_in):
        t = t_in.reshape(-1)[:1].float().cpu() / 1000.0 + 1.0 / 1000
        a, sg = float(ns.marginal_alpha(t)[0]), float(ns.marginal_std(t)[0])
        return x * (sg / (a * a * std * std + sg * sg))
    return model


ONLY = set(a for a in sys.argv[1:] if not a.startswith("-"))   # e.g. `make_golden.py attnpp`: rewrite just that file


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    arrays["torch_version"] = np.asarray(torch.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays) - 1} arrays")


def main():
    emb = synthetic_text_embeddings()
    ref_shims.install(emb)
    torch.set_grad_enabled(True)

    from GeneralModel import Hyper_unet_general as RU
    from GeneralModel import diffusion_general as RD
    from DWT_IDWT.DWT_IDWT_layer import DWT_2D, IDWT_2D
    from core import dpm_solver_pytorch as RS
    from core import Attention as RA

    # ---- (1) schedule tables --------------------------------------------------------------
    arrs = {}
    for sched in ("cosine", "linear"):
        for T in (10, 50, 1000):
            d = RD.GeneralDiffusion(denoise_fn=None)
            d.set_new_noise_schedule({"schedule": sched, "n_timestep": T}, "cpu")
            for k, v in d.state_dict().items():
                arrs[f"{sched}_{T}_{k}"] = v
            arrs[f"{sched}_{T}_sqrt_alphas_cumprod_prev"] = d.sqrt_alphas_cumprod_prev
    save("schedules", **arrs)

    # ---- (2) timestep embedding -----------------------------------------------------------
    t_int = torch.tensor([1, 2, 17, 500, 1000])
    t_frac = torch.tensor([0.0, 0.37, 12.25, 998.9990])
    save("gamma_embedding", t_int=t_int, e_int=RU.gamma_embedding(t_int, 32),
         t_frac=t_frac, e_frac=RU.gamma_embedding(t_frac, 32), e_odd=RU.gamma_embedding(t_frac, 33))

    # ---- (3) Haar DWT / IDWT fwd + bwd -------------------------------------------------------
    arrs = {}
    for tag, shape in (("a", (2, 6, 8, 8)), ("b", (1, 16, 16, 12))):
        x = randn(11, *shape).requires_grad_(True)
        bands = DWT_2D("haar")(x)
        gout = [randn(12 + i, *bands[0].shape) for i in range(4)]
        torch.autograd.backward(bands, gout)
        for n, v in zip(("ll", "lh", "hl", "hh"), bands):
            arrs[f"{tag}_{n}"] = v
        arrs[f"{tag}_gx"] = x.grad
        ins = [randn(20 + i, *bands[0].shape).requires_grad_(True) for i in range(4)]
        y = IDWT_2D("haar")(*ins)
        gy = randn(30, *y.shape)
        y.backward(gy)
        arrs[f"{tag}_idwt"] = y
        for n, v in zip(("ll", "lh", "hl", "hh"), ins):
            arrs[f"{tag}_idwt_g{n}"] = v.grad
        arrs[f"{tag}_recon"] = IDWT_2D("haar")(*[b_.detach() for b_ in bands])
    save("haar", **arrs)

    # ---- (4) modulated_conv3d fwd + grads -------------------------------------------------------
    arrs = {}
    for b in (1, 3):
        for k in (1, 3):
            x = randn(40, b, 5, 4, 6, 6).requires_grad_(True)
            w = (randn(41, 7, 5, k, k, k) / (5 * k ** 3) ** 0.5).requires_grad_(True)
            s = (1 + 0.3 * randn(42, b, 5, 1, 1)).requires_grad_(True)
            y = RU.modulated_conv3d(x=x, w=w, s=s, stride=(1, 1, 1), padding=(k // 2,) * 3)
            y.backward(randn(43, *y.shape))
            arrs.update({f"b{b}k{k}_y": y, f"b{b}k{k}_gx": x.grad, f"b{b}k{k}_gw": w.grad, f"b{b}k{k}_gs": s.grad})
    save("modconv", **arrs)

    # ---- (5) blocks fwd + bwd -------------------------------------------------------------
    arrs = {}
    E = 128
    temb, pemb = randn(50, 2, E), randn(51, 2, E)

    def run(tag, mod, args, seed=60):
        fill_weights_(mod, seed=7)
        mod.eval()
        out = mod(*args)
        outs = [out] if torch.is_tensor(out) else ([out[0]] + list(out[1]))
        torch.autograd.backward(outs, [randn(seed + i, *o.shape) for i, o in enumerate(outs)])
        for i, o in enumerate(outs):
            arrs[f"{tag}_y{i}"] = o
        for i, a in enumerate(a for a in args if torch.is_tensor(a) and a.requires_grad):
            if a.grad is not None:              # flag=True blocks never touch the time embedding
                arrs[f"{tag}_gin{i}"] = a.grad
        for k, p in mod.named_parameters():
            if p.grad is not None:
                arrs[f"{tag}_gp_{k}"] = torch.stack([p.grad.sum(), p.grad.abs().sum()])

    for n in (4, 8):
        mk = lambda seed, ch, hh: randn(seed, 2, ch, n, hh, hh).requires_grad_(True)
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        run(f"n{n}_adaption", RU.AdaptionModulateBEST(1, 4, E), (mk(52, 1, 16), te, pe))
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        run(f"n{n}_res", RU.ResBlockModulateBEST(4, 8, E), (mk(53, 4, 16), te, pe))
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        run(f"n{n}_res_same_flag", RU.ResBlockModulateBEST(8, 8, E, flag=True), (mk(54, 8, 16), te, pe))
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        run(f"n{n}_down", RU.ResblockDownOneModulateBEST(4, 8, E), (mk(55, 4, 16), te, pe))
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        run(f"n{n}_down_flag", RU.ResblockDownOneModulateBEST(4, 8, E, flag=True), (mk(56, 4, 16), te, pe))
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        skip = [mk(57 + i, 16, 8) for i in range(3)]
        blk = RU.ResblockUpOneModulateBEST(16, 8, E)
        fill_weights_(blk, seed=7)
        blk.eval()
        xin = mk(61, 48, 8)
        y = blk(xin, te, skip, pe)
        y.backward(randn(62, *y.shape))
        arrs[f"n{n}_up_y0"] = y
        arrs[f"n{n}_up_gin0"] = xin.grad
        for i in range(3):
            arrs[f"n{n}_up_gskip{i}"] = skip[i].grad
        arrs[f"n{n}_up_gte"], arrs[f"n{n}_up_gpe"] = te.grad, pe.grad
        te, pe = temb.clone().requires_grad_(True), pemb.clone().requires_grad_(True)
        run(f"n{n}_final", RU.FinalBlockModulateBEST(4, 1, E), (mk(63, 12, 16), te, pe))
    save("blocks", **arrs)

    # ---- (6) whole UNet, tiny width ---------------------------------------------------------
    torch.set_grad_enabled(False)
    arrs = {}
    net = fill_weights_(RU.WavBEST(channels=TINY)).eval()
    for c in (4, 8):
        d = case_inputs(100 + c, 2, c, 16)
        for prompt in ("QB", "WV3", "GF2", "WV2", "WV4"):
            arrs[f"c{c}_{prompt}_int"] = net(d["x_t"], torch.tensor([[3], [977]]), d["PAN"], d["MS"], prompt)
        arrs[f"c{c}_WV3_frac"] = net(d["x_t"], torch.tensor([0.25, 731.4]), d["PAN"], d["MS"], "WV3")
    d = case_inputs(120, 1, 8, 32, 16)          # non-square tile
    arrs["nonsquare"] = net(d["x_t"], torch.tensor([[500]]), d["PAN"], d["MS"], "WV3")
    save("unet_tiny", **arrs)

    # ---- (7) whole UNet, full width, one 8-ch 64x64 tile ---------------------------------------
    netF = fill_weights_(RU.WavBEST(channels=FULL)).eval()
    d = case_inputs(3407, 1, 8, 64)
    save("unet_full", y=netF(d["x_t"], torch.tensor([[250]]), d["PAN"], d["MS"], "WV3"))
    del netF

    # ---- (7b) BASELINE configs[2] network: channel_multiplier [64,128,256,512] (config/general.json:52-54) ------------
    if not ONLY or "unet_c3" in ONLY:
        netW = fill_weights_(RU.WavBEST(channels=WIDE)).eval()
        d = case_inputs(3408, 1, 8, 64)
        arrs = {"y64": netW(d["x_t"], torch.tensor([[612]]), d["PAN"], d["MS"], "WV3")}
        d = case_inputs(3409, 1, 8, 256)             # the config's tile size, one forward (fractional DPM-Solver time)
        arrs["y256"] = netW(d["x_t"], torch.tensor([431.7]), d["PAN"], d["MS"], "WV3")
        save("unet_c3", **arrs)
        del netW

    # ---- (8) q_sample / training loss -------------------------------------------------------
    torch.set_grad_enabled(True)
    arrs = {}
    net = fill_weights_(RU.WavBEST(channels=TINY)).eval()        # dropout off: RNG-free forward
    diff = RD.GeneralDiffusion(net, loss_type="l1")
    diff.set_loss("cpu")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    d = case_inputs(130, 3, 8, 16)
    a = torch.tensor([0.9, 0.5, 0.1]).view(-1, 1, 1, 1)
    arrs["q_sample"] = diff.q_sample(d["Res"], a, noise=randn(131, *d["Res"].shape))
    for b in (3, 1):
        d = case_inputs(132 + b, b, 8, 16)
        np.random.seed(5)
        torch.manual_seed(6)
        net.zero_grad()
        loss = diff(d, "WV3")
        loss.backward()
        arrs[f"loss_b{b}"] = loss
        arrs[f"gsum_b{b}"] = torch.stack([torch.stack([p.grad.sum(), p.grad.abs().sum()]) if p.grad is not None
                                          else torch.full((2,), float("nan")) for p in net.parameters()])
    save("train", **arrs)

    # ---- (9) DDPM ancestral sampling ------------------------------------------------------------
    torch.set_grad_enabled(False)
    arrs = {}
    for T in (10, 50):
        diff = RD.GeneralDiffusion(net, loss_type="l1")
        diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": T}, "cpu")
        d = case_inputs(140 + T, 2, 8, 16)
        torch.manual_seed(T)
        stack = diff.super_resolution(d, False, "WV3", 3.0)       # prompt slip: runs "QB", returns the stack
        arrs[f"T{T}_stack_shape"] = np.asarray(stack.shape)
        arrs[f"T{T}_final"] = stack[-2:]
        arrs[f"T{T}_mid"] = stack[2:6]
        torch.manual_seed(T)
        arrs[f"T{T}_last_only"] = diff.p_sample_loop(d, continous=False, prompt="WV3")
    save("ddpm", **arrs)

    # ---- (9a) one reverse step through p_mean_variance and its x0-parameterised twin p_mean_variance_xo (:154-190) ------
    diff = RD.GeneralDiffusion(net, loss_type="l1")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 50}, "cpu")
    d = case_inputs(160, 2, 8, 16)
    arrs = {}
    for t in (0, 17, 49):
        m, lv = diff.p_mean_variance(d["x_t"].clone(), t, clip_denoised=True, x_in=d, prompt="WV3")
        arrs[f"t{t}_mean"], arrs[f"t{t}_logvar"] = m, lv
        m, lv = diff.p_mean_variance_xo(d["x_t"].clone(), t, clip_denoised=True, x_in=d, prompt="WV3")
        arrs[f"t{t}_xo_mean"], arrs[f"t{t}_xo_logvar"] = m, lv
    m, _ = diff.p_mean_variance_xo(d["x_t"].clone() * 3.0, 17, clip_denoised=False, x_in=d, prompt="GF2")
    arrs["t17_xo_mean_unclipped"] = m
    save("ddpm_xo", **arrs)

    # ---- (9b) the full T = 1000 chain of BASELINE config 2 (one tile; ~2 min of reference CPU time) -----------------
    if not ONLY or "ddpm1000" in ONLY:
        diff = RD.GeneralDiffusion(net, loss_type="l1")
        diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
        d = case_inputs(77, 1, 8, 16)
        torch.manual_seed(1000)
        save("ddpm1000", last_only=diff.p_sample_loop(d, continous=False, prompt="WV3"))

    # ---- (9c) FULL-width chains (ch 32-256): the widths at which the production kernels of the HIP path engage (Winograd
    # along the bands needs Cout % 32 == 0, the composed Conv_0 + LL convolution Cout % 64 == 0 -- the TINY-width chains
    # above can only reach the direct kernels).  T = 50 on two 8x32x32 tiles, T = 1000 and the 31-NFE DPM-Solver++ on one
    # 8x16x16 tile: ~4 min of reference CPU time. ------------------------------------------------------------------------
    if not ONLY or "chains_full" in ONLY:
        arrs = {}
        netF = fill_weights_(RU.WavBEST(channels=FULL)).eval()
        diff = RD.GeneralDiffusion(netF, loss_type="l1")
        diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 50}, "cpu")
        d = case_inputs(171, 2, 8, 32)
        torch.manual_seed(50)
        arrs["T50_last_only"] = diff.p_sample_loop(d, continous=False, prompt="WV3")
        torch.manual_seed(50)
        stack = diff.super_resolution(d, False, "WV3", 3.0)       # prompt slip: runs "QB", returns the stack
        arrs["T50_stack_shape"], arrs["T50_stack_final"] = np.asarray(stack.shape), stack[-2:]
        print("chains_full: T=50 done", flush=True)
        diff = RD.GeneralDiffusion(netF, loss_type="l1")
        diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
        d = case_inputs(172, 1, 8, 16)
        torch.manual_seed(1001)
        arrs["T1000_last_only"] = diff.p_sample_loop(d, continous=False, prompt="WV3")
        print("chains_full: T=1000 done", flush=True)
        d = case_inputs(173, 1, 8, 16)
        torch.manual_seed(11)
        arrs["dpm_out"] = diff.sample_by_dpmsolver(d, "WV3")
        save("chains_full", **arrs)
        del netF

    # ---- (10) DPM-Solver++ ------------------------------------------------------------------
    arrs = {}
    diff = RD.GeneralDiffusion(net, loss_type="l1")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    d = case_inputs(150, 1, 8, 16)      # B=1: the reference's x_start conversion (dpm_solver_pytorch.py:302-306)
    calls = []                          # broadcasts alpha_t[B] against W and fails for B>1 (unless B == W)
    orig_forward = net.forward
    net.forward = lambda x, t, **kw: (calls.append(t.detach().clone()), orig_forward(x, t, **kw))[1]
    torch.manual_seed(9)
    arrs["dpm_out"] = diff.sample_by_dpmsolver(d, "WV3")
    arrs["dpm_model_times"] = torch.stack([c[0] for c in calls])
    arrs["dpm_nfe"] = np.asarray(len(calls))
    net.forward = orig_forward
    ns = RS.NoiseScheduleVP("discrete", betas=diff.betas)
    solver = RS.DPM_Solver(lambda x, t: x, ns, algorithm_type="dpmsolver++", correcting_x0_fn="dynamic_thresholding")
    for steps in (20, 30, 31, 32):
        outer, orders = solver.get_orders_and_timesteps_for_singlestep_solver(steps, 3, "logSNR", 1.0, 1e-3, "cpu")
        arrs[f"grid_{steps}"], arrs[f"orders_{steps}"] = outer, np.asarray(orders)
    tq = torch.tensor([1e-3, 0.0137, 0.25, 0.5, 0.999, 1.0])
    arrs["ns_t"] = tq
    arrs["ns_log_alpha"], arrs["ns_std"], arrs["ns_lambda"] = ns.marginal_log_mean_coeff(tq), ns.marginal_std(tq), ns.marginal_lambda(tq)
    arrs["ns_inv_lambda"] = ns.inverse_lambda(ns.marginal_lambda(tq))
    x0 = randn(151, 3, 8, 16, 16)
    x0[0, 0, 0, :5] = torch.tensor([9.0, -7.0, 5.0, 30.0, -2.5])
    x0[1] *= 0.2
    arrs["thresh_in_seed"] = np.asarray(151)
    arrs["thresh_out"] = solver.dynamic_thresholding_fn(x0, None)
    # other solver families on a closed-form "model" (cheap, exercises every update rule)
    toy = lambda x, t: 0.3 * x + 0.1 * torch.tanh(x) * t.view(-1, 1, 1, 1)   # Lipschitz: no chaotic error growth
    xT = randn(152, 2, 4, 8, 8)
    for algo in ("dpmsolver", "dpmsolver++"):
        for method, order, skip, stype in (("singlestep", 3, "logSNR", "dpmsolver"), ("singlestep", 2, "time_uniform", "taylor"),
                                           ("singlestep", 3, "time_quadratic", "taylor"), ("multistep", 2, "time_uniform", "dpmsolver"),
                                           ("multistep", 3, "logSNR", "dpmsolver"), ("multistep", 2, "logSNR", "taylor"),
                                           ("singlestep_fixed", 2, "logSNR", "dpmsolver"), ("adaptive", 3, "logSNR", "dpmsolver")):
            s2 = RS.DPM_Solver(RS.model_wrapper(toy, ns, model_type="noise"), ns, algorithm_type=algo)
            arrs[f"toy_{algo}_{method}_{order}_{skip}_{stype}"] = s2.sample(xT, steps=9, order=order, skip_type=skip,
                                                                          method=method, solver_type=stype)
    save("dpm_solver", **arrs)

    # ---- (10b) adaptive driver, add_noise, inverse (dpm_solver_pytorch.py:982-1079) ---------------------------------
    # The adaptive `toy_*` cases above are ill conditioned: a 1e-7 relative change of x_T moves the REFERENCE's own
    # output by O(1) relative (stored below as evidence), so they cannot serve as parity vectors for an implementation
    # whose arithmetic differs in the last bit.  The Gaussian-denoiser model is the parity problem for the driver.
    arrs = {}
    gm = gauss_model(ns)
    xg = randn(153, 2, 4, 8, 8)
    for algo in ("dpmsolver", "dpmsolver++"):
        mk = lambda model: RS.DPM_Solver(RS.model_wrapper(model, ns, model_type="noise"), ns, algorithm_type=algo)
        for order in (2, 3):
            for tag, kw in (("", {}), ("_tight", dict(atol=1e-4, rtol=1e-3))):
                run = lambda x0: mk(gm).sample(x0, order=order, method="adaptive", skip_type="logSNR", **kw)
                base = run(xg)
                arrs[f"gauss_{algo}_adaptive_{order}{tag}"] = base
                # how far the reference moves from itself under 1e-6 .. 1e-5 relative input changes (the controller's
                # accept / reject decisions flip): the yardstick for an implementation with different last bits
                sens = [float(((run(xg * (1 + e)) / (1 + e)) - base).norm() / base.norm()) for e in (1e-6, -1e-6, 3e-6, 1e-5)]
                arrs[f"gauss_{algo}_adaptive_{order}{tag}_sensitivity"] = np.asarray(sens)
        arrs[f"toy_{algo}_adaptive_3_perturbed"] = mk(toy).sample(xT * (1 + 1e-7), steps=9, order=3, skip_type="logSNR",
                                                                 method="adaptive", solver_type="dpmsolver")
        # inverse (data -> noise along the ODE) and back, fixed-step multistep order 2
        data = 0.5 * randn(154, 2, 4, 8, 8)
        z = mk(gm).inverse(data, steps=12, order=2, skip_type="time_uniform", method="multistep")
        arrs[f"gauss_{algo}_inverse"] = z
        arrs[f"gauss_{algo}_inverse_back"] = mk(gm).sample(z, steps=12, order=2, skip_type="time_uniform", method="multistep")
    sol = RS.DPM_Solver(RS.model_wrapper(gm, ns, model_type="noise"), ns)
    xn = randn(155, 2, 4, 8, 8)
    arrs["add_noise_t1"] = sol.add_noise(xn, torch.tensor([0.3]), noise=randn(156, 1, 2, 4, 8, 8))
    arrs["add_noise_t2"] = sol.add_noise(xn, torch.tensor([0.1, 0.9]), noise=randn(157, 2, 2, 4, 8, 8))
    save("dpm_adaptive", **arrs)

    # ---- (11) core/Attention.py standalone ops ---------------------------------------------------
    arrs = {}
    torch.manual_seed(0)
    for c, hw in ((64, 8), (128, 16)):
        m = fill_weights_(RA.SpatialSelfAttention(c), seed=3).eval()
        arrs[f"ssa_c{c}"] = m(randn(162, 2, c, hw, hw))
    m = fill_weights_(RA.CrossAttention(128, context_dim=768, heads=8, dim_head=16), seed=3).eval()
    arrs["cross"] = m(randn(163, 2, 256, 128), context=randn(164, 2, 77, 768))
    mask = torch.ones(2, 77, dtype=torch.bool); mask[0, 40:] = False; mask[1, 5:9] = False
    arrs["cross_masked"] = m(randn(163, 2, 256, 128), context=randn(164, 2, 77, 768), mask=mask)
    m = fill_weights_(RA.CrossAttention(128, heads=4, dim_head=32), seed=3).eval()
    arrs["self"] = m(randn(165, 2, 64, 128))
    # the CLIP-context shape of the reference's defaults (CrossAttention(query_dim, context_dim=768, heads=8, dim_head=64),
    # core/Attention.py:165-170) at a small query count: 77 keys x d_head 64 -- the small-context kernel's shape
    m = fill_weights_(RA.CrossAttention(128, context_dim=768, heads=2, dim_head=64), seed=3).eval()
    arrs["cross_d64"] = m(randn(171, 2, 200, 128), context=randn(164, 2, 77, 768))
    arrs["cross_d64_masked"] = m(randn(171, 2, 200, 128), context=randn(164, 2, 77, 768), mask=mask)
    m = fill_weights_(RA.BasicTransformerBlock(128, 8, 16, context_dim=768, checkpoint=False), seed=3).eval()
    arrs["block"] = m(randn(166, 2, 64, 128), context=randn(167, 2, 77, 768))
    m = RA.SpatialTransformer(128, 8, 16, depth=1, context_dim=768, use_checkpoint=False)
    fill_weights_(m, seed=3).eval()       # note: this also un-zeroes proj_out so the block is observable
    arrs["spatial_transformer"] = m(randn(168, 2, 128, 16, 16), context=randn(169, 2, 77, 768))
    ff = fill_weights_(RA.FeedForward(64, glu=True), seed=3).eval()
    arrs["geglu_ff"] = ff(randn(170, 3, 10, 64))
    save("attention", **arrs)

    # ---- (12) AttnBlockpp / NIN (Hyper_unet_general.py:471-515; never instantiated by WavBEST) ------
    # `channels` must be the FOLDED channel count C*N (the block folds 'b c n h w -> b (c n) h w' before its GroupNorm).
    arrs = {}
    for tag, (b, c, n, hw, rescale) in {"a": (2, 16, 4, 8, True), "b": (1, 8, 8, 16, False)}.items():
        m = fill_weights_(RU.AttnBlockpp(c * n, skip_rescale=rescale), seed=5).eval()
        arrs[f"{tag}_y"] = m(randn(180, b, c, n, hw, hw))
    save("attnpp", **arrs)

    # ---- (13) SAM (core/metrics.py:91-112), the one validation metric whose arithmetic is NumPy's and therefore reproducible
    # here: as the driver calls it (general_sharpening_joint_random_batch_finetune.py:145), on H x W x C float32 arrays in
    # [0, 1]; also float64 inputs, 4 bands, and pixels that produce NaN (a zero spectrum: 0 / 0) or arccos of a ratio that
    # rounds above 1 (identical spectra) -- both of which the reference turns into 0 / NaN -> 0.  Inputs are stored (small).
    if not ONLY or "metrics" in ONLY:
        ref_shims.install_metrics_stand_ins()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from core import metrics as RM
        arrs = {}
        rng = np.random.default_rng(2024)
        cases = {"wv3_f32": (24, 20, 8, np.float32), "gf2_f32": (16, 16, 4, np.float32), "wv3_f64": (12, 10, 8, np.float64)}
        for tag, (h, w, c, dt) in cases.items():
            hr = rng.random((h, w, c)).astype(dt)
            sr = np.clip(hr + 0.05 * rng.standard_normal((h, w, c)), 0.0, 1.0).astype(dt)
            sr[0, 0] = 0                      # a zero spectrum in the prediction: 0 / 0 = NaN -> counted as angle 0
            hr[1, 1] = 0                      # ... and in the target
            sr[2, 2] = hr[2, 2]               # identical spectra: ratio ~ 1 (may round above 1 -> NaN -> 0)
            sr[3, 3] = 3 * hr[3, 3]           # parallel spectra
            with np.errstate(all="ignore"):
                arrs[f"{tag}_sam"] = np.float64(RM.SAM_numpy(sr, hr))          # argument order of the driver
                arrs[f"{tag}_sam_swapped"] = np.float64(RM.SAM_numpy(hr, sr))
            arrs[f"{tag}_hr"], arrs[f"{tag}_sr"] = hr, sr
        save("metrics", **arrs)


if __name__ == "__main__":
    main()
