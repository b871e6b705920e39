"""CPU-only tests of the product's host side: the C-ABI library loads and exports what the header
declares, the module tree / checkpoint contract, schedule tables, DPM-Solver host maths, factory
options and error behaviour.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from oracle import unet_ref as U
from oracle.make_golden import TINY


def test_library_exports_every_declared_symbol():
    from tmdiff_amd import _lib
    header = open(os.path.join(ROOT, "include", "tmdiff_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tmdiff_[a-z0-9_]+)\s*\(", header))
    declared -= {"tmdiff_conv3d_desc"}
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/tmdiff_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.ABI_VERSION == 1
    assert ctypes.sizeof(_lib.Conv3dDesc) % 8 == 0


def test_state_dict_contract_and_clip_keys_ignored():
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion, GaussianDiffusion
    assert GaussianDiffusion is GeneralDiffusion
    ref = U.fill_weights_(U.WavBESTRef(channels=TINY))
    net = WavBEST(channels=TINY)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert len(net.state_dict()) == 272
    for k, v in ref.state_dict().items():
        assert net.state_dict()[k].shape == v.shape, k
    sd = dict(ref.state_dict())
    sd["clip_text_model.transformer.text_model.embeddings.token_embedding.weight"] = torch.zeros(3)
    net.load_state_dict(sd, strict=True)              # reference checkpoints carry the CLIP weights
    diff = GeneralDiffusion(net, "l1")
    diff.set_new_noise_schedule({"schedule": "linear", "n_timestep": 10}, "cpu")
    keys = list(diff.state_dict().keys())
    assert keys[:12] == ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                         "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                         "sqrt_recip_alphas_cumprod_1", "sqrt_recipm1_alphas_cumprod_1", "posterior_variance",
                         "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    assert keys[12] == "denoise_fn.embed.0.weight" and len(keys) == 284


def test_schedule_tables_bitwise(golden):
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    g = golden("schedules")
    for sched in ("cosine", "linear"):
        for T in (10, 50, 1000):
            d = GeneralDiffusion(None)
            d.set_new_noise_schedule({"schedule": sched, "n_timestep": T}, "cpu")
            for k, v in d.state_dict().items():
                assert np.array_equal(v.numpy(), g[f"{sched}_{T}_{k}"]), (sched, T, k)
            assert np.array_equal(d.sqrt_alphas_cumprod_prev, g[f"{sched}_{T}_sqrt_alphas_cumprod_prev"])
            assert d.num_timesteps == T and len(d._step_coef) == T
            i = T // 2
            assert d._step_coef[i][0] == float(d.sqrt_recip_alphas_cumprod_1[i])
            assert abs(d._step_coef[i][4] - float((0.5 * d.posterior_log_variance_clipped[i]).exp())) < 1e-12


def test_error_behaviour_matches_reference():
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.Hyper_unet_general import WavBEST
    d = GeneralDiffusion(None, loss_type="huber")
    with pytest.raises(NotImplementedError):
        d.set_loss("cpu")
    with pytest.raises(NotImplementedError):
        d.set_new_noise_schedule({"schedule": "sigmoid", "n_timestep": 10}, "cpu")
    net = WavBEST(channels=TINY)
    x = torch.zeros(1, 4, 16, 16)
    with pytest.raises(RuntimeError, match="HIP kernels only"):     # no CPU fallback behind the product
        net(x, torch.tensor([[1]]), torch.zeros(1, 1, 16, 16), x, "QB")


def test_dpm_solver_host_math(golden):
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.dpm_solver import DPM_Solver, NoiseScheduleVP
    g = golden("dpm_solver")
    d = GeneralDiffusion(None)
    d.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    ns = NoiseScheduleVP("discrete", betas=d.betas)
    tq = torch.tensor(g["ns_t"])
    assert_close(ns.marginal_log_mean_coeff(tq), g["ns_log_alpha"], 1e-6, 1e-6)
    assert_close(ns.marginal_std(tq), g["ns_std"], 1e-6, 1e-6)
    assert_close(ns.marginal_lambda(tq), g["ns_lambda"], 1e-6, 1e-6)
    assert_close(ns.inverse_lambda(ns.marginal_lambda(tq)), g["ns_inv_lambda"], 1e-5, 1e-5)
    solver = DPM_Solver(lambda x, t: x, ns, algorithm_type="dpmsolver++", correcting_x0_fn="dynamic_thresholding")
    for steps in (20, 30, 31, 32):
        outer, orders = solver.get_orders_and_timesteps_for_singlestep_solver(steps, 3, "logSNR", 1.0, 1e-3)
        assert list(orders) == list(g[f"orders_{steps}"])
        assert_close(outer, g[f"grid_{steps}"], 1e-5, 1e-5, "outer grid")
    with pytest.raises(ValueError):
        NoiseScheduleVP("sigmoid")
    with pytest.raises(ValueError):
        solver.get_time_steps("bogus", 1.0, 1e-3, 5)
    for sched in ("linear", "cosine"):                # continuous schedules: lambda round trip
        nc = NoiseScheduleVP(sched)
        t = torch.tensor([0.05, 0.3, 0.9])
        assert_close(nc.inverse_lambda(nc.marginal_lambda(t)), t, 1e-4, 1e-4, sched)


def test_define_general_reads_reference_options():
    from tmdiff_amd import networks
    opt = {"model": {"unet": {"channel_multiplier": TINY}, "diffusion": {"loss_type": "l2"}, "init_type": "orthogonal"},
           "phase": "train", "gpu_ids": None, "distributed": False}
    net = networks.define_General(opt)
    assert net.loss_type == "l2" and net.denoise_fn.channels == TINY
    assert float(net.denoise_fn.down1.conv20.conv20.bias.abs().sum()) == 0.0      # orthogonal init zeroes biases
    with pytest.raises(NotImplementedError):
        networks.init_weights(net, "xavier")
    opt["phase"] = "val"
    emb = {k: torch.ones(1, 768) * i for i, k in enumerate(("QB", "WV3", "GF2", "WV2", "WV4"))}
    opt["model"]["text_embeddings"] = emb
    net = networks.define_General(opt)
    assert float(net.denoise_fn.get_embeding("GF2")[0, 0]) == 2.0
    assert net.denoise_fn.get_embeding("LANDSAT") is None


def test_psnr_and_res2img():
    from tmdiff_amd.util import img2res, psnr, res2img
    a = torch.rand(2, 4, 8, 8)
    b = a + 0.01
    assert abs(psnr(a, b) - 40.0) < 1e-3          # fp32 (a+0.01)-a is 0.01 only to ~1e-6
    assert psnr(a, a) > 200
    assert torch.equal(img2res(res2img(a, b), b), (a + b) - b)
