"""Oracle (test infrastructure): the diffusion process restated in plain PyTorch/NumPy, CPU.

Follows GeneralModel/diffusion_general.py of the reference:
    make_beta_schedule :29-43      betas_for_alpha_bar  :46-63
    set_new_noise_schedule :86-132 q_posterior          :134-138
    p_mean_variance    :154-171    dynamic_clip         :192-200
    p_sample           :203-208    p_sample_loop        :210-225
    sample_by_dpmsolver:227-255    super_resolution     :337-339
    q_sample           :341-347    p_losses_dynamic     :349-370
    predict_start_from_noise :376-378
and utils/util.py:135-142 (res2img / img2res).

Quirks kept on purpose: ``super_resolution`` passes ``prompt`` into the
``continous`` slot, so sampling always runs with prompt "QB" and returns the whole stack;
the network is trained to predict x_0 but sampled as if it predicted noise.

``noise_fn`` is an oracle-only hook: parity tests inject CPU-generated noise through it
(device RNG streams differ between CPU and HIP).
"""
import math

import numpy as np
import torch
from torch import nn

from .dpm_solver_ref import DPM_Solver, NoiseScheduleVP, model_wrapper


def res2img(res, lms):
    return res + lms


def img2res(img, lms):
    return img - lms


def make_beta_schedule(schedule, n_timestep):
    if schedule == "linear":
        scale = 1000 / n_timestep
        return np.linspace(scale * 1e-6, scale * 1e-2, n_timestep, dtype=np.float64)
    if schedule == "cosine":
        abar = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        n = n_timestep
        return np.array([min(1 - abar((i + 1) / n) / abar(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(schedule)


class GeneralDiffusionRef(nn.Module):
    def __init__(self, denoise_fn, loss_type="l1", noise_fn=None):
        super().__init__()
        self.denoise_fn = denoise_fn
        self.loss_type = loss_type
        self.noise_fn = noise_fn or (lambda like: torch.randn_like(like))

    def set_loss(self, device):
        table = {"l1": nn.L1Loss, "l2": nn.MSELoss, "smooth_l1": nn.SmoothL1Loss}
        if self.loss_type not in table:
            raise NotImplementedError()
        self.loss_func = table[self.loss_type]().to(device)

    def set_new_noise_schedule(self, schedule_opt, device):
        f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        betas = make_beta_schedule(schedule_opt["schedule"], schedule_opt["n_timestep"])
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.sqrt_alphas_cumprod_prev = np.sqrt(np.append(1.0, ac))     # float64, length T+1
        self.num_timesteps = int(betas.shape[0])
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        for name, val in (
            ("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
            ("sqrt_alphas_cumprod", np.sqrt(ac)),
            ("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - ac)),
            ("log_one_minus_alphas_cumprod", np.log(1.0 - ac)),
            ("sqrt_recip_alphas_cumprod_1", np.sqrt(1.0 / ac)),
            ("sqrt_recipm1_alphas_cumprod_1", np.sqrt(1.0 / ac - 1)),
            ("posterior_variance", post_var),
            ("posterior_log_variance_clipped", np.log(np.maximum(post_var, 1e-20))),
            ("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1.0 - ac)),
            ("posterior_mean_coef2", (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
        ):
            self.register_buffer(name, f32(val))

    # ---- reverse process -------------------------------------------------------------
    def predict_start_from_noise(self, x_t, t, noise):
        return self.sqrt_recip_alphas_cumprod_1[t] * x_t - self.sqrt_recipm1_alphas_cumprod_1[t] * noise

    def q_posterior(self, x_start, x_t, t):
        mean = self.posterior_mean_coef1[t] * x_start + self.posterior_mean_coef2[t] * x_t
        return mean, self.posterior_log_variance_clipped[t]

    @torch.no_grad()
    def p_mean_variance(self, x, t, clip_denoised=True, x_in=None, prompt="QB", guidance=1.0):
        b = x.shape[0]
        time_in = torch.full((b, 1), t + 1, device=x.device, dtype=torch.long)
        eps = self.denoise_fn(x, time_in, x_in["PAN"], x_in["MS"], prompt)
        x0 = self.predict_start_from_noise(x, t, eps)
        if clip_denoised:
            x0 = x0.clamp(-1.0, 1.0)
        return self.q_posterior(x0, x, t)

    @torch.no_grad()
    def p_mean_variance_xo(self, x, t, clip_denoised=True, x_in=None, prompt="QB", guidance=1.0):
        """x0-parameterised twin (reference diffusion_general.py:173-190; called by nothing there): the network output is
        x_0 itself -- clamp, then the posterior."""
        b = x.shape[0]
        time_in = torch.full((b, 1), t + 1, device=x.device, dtype=torch.long)
        x0 = self.denoise_fn(x, time_in, x_in["PAN"], x_in["MS"], prompt)
        if clip_denoised:
            x0 = x0.clamp(-1.0, 1.0)
        return self.q_posterior(x0, x, t)

    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=True, condition_x=None, prompt="QB", guidance=1.0):
        mean, logvar = self.p_mean_variance(x, t, clip_denoised, condition_x, prompt, guidance)
        noise = self.noise_fn(x) if t > 0 else torch.zeros_like(x)
        return mean + noise * (0.5 * logvar).exp()

    @torch.no_grad()
    def p_sample_loop(self, x_in, continous=False, prompt="QB", guidance=1.0):
        inter = 1 | (self.num_timesteps // 10)
        img = self.noise_fn(x_in["Res"])
        ret = res2img(img, x_in["MS"])
        for i in reversed(range(self.num_timesteps)):
            img = self.p_sample(img, i, condition_x=x_in, prompt=prompt, guidance=guidance)
            if i % inter == 0:
                ret = torch.cat([ret, res2img(img, x_in["MS"])], dim=0)
        return ret if continous else ret[-1]

    @torch.no_grad()
    def super_resolution(self, x_in, continous, prompt, guidance):
        return self.p_sample_loop(x_in, prompt)       # positional slip kept (ref :339)

    @torch.no_grad()
    def sample_by_dpmsolver(self, x_in, prompt, steps=30, return_trace=False):
        x_T = self.noise_fn(x_in["Res"])
        ns = NoiseScheduleVP("discrete", betas=self.betas)
        fn = model_wrapper(self.denoise_fn, ns, model_type="x_start",
                           model_kwargs={"PAN": x_in["PAN"], "MS": x_in["MS"], "prompt": prompt})
        solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn="dynamic_thresholding")
        x = solver.sample(x_T, steps=steps, order=3, skip_type="logSNR", method="singlestep", denoise_to_zero=True)
        out = res2img(x, x_in["MS"])
        return (out, solver) if return_trace else out

    # ---- forward process / loss --------------------------------------------------------
    def q_sample(self, x_start, continuous_sqrt_alpha_cumprod, noise=None):
        noise = self.noise_fn(x_start) if noise is None else noise
        a = continuous_sqrt_alpha_cumprod
        return a * x_start + (1 - a ** 2).sqrt() * noise

    def p_losses_dynamic(self, x_in, prompt=None):
        x0 = x_in["Res"]
        b = x0.shape[0]
        time_in = torch.from_numpy(np.random.randint(1, self.num_timesteps + 1, size=b))
        a = torch.FloatTensor(np.atleast_1d(self.sqrt_alphas_cumprod_prev[time_in.numpy()])).to(x0.device)
        noise = self.noise_fn(x0)
        x_noisy = self.q_sample(x0, a.view(-1, 1, 1, 1), noise)
        x_recon = self.denoise_fn(x_noisy, time_in.to(x0.device).view(b, -1), x_in["PAN"], x_in["MS"], prompt)
        return self.loss_func(x0, x_recon)

    def forward(self, x, *args, **kwargs):
        return self.p_losses_dynamic(x, *args, **kwargs)
