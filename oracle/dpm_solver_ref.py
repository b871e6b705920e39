"""Oracle (test infrastructure): DPM-Solver / DPM-Solver++ restated compactly, CPU torch.

Follows core/dpm_solver_pytorch.py of the reference (itself the public DPM-Solver file,
Lu et al. 2022):
    NoiseScheduleVP        :6-181      model_wrapper           :184-348
    DPM_Solver.__init__    :352-428    dynamic_thresholding_fn :430-439
    data_prediction_fn     :447-456    get_time_steps          :467-495
    singlestep order plan  :497-555    first/second/third singlestep updates :563-815
    multistep 2nd/3rd      :817-927    adaptive :982-1043      sample :1081-1289
    interpolate_fn         :1296-1335
Everything stays in fp32 torch tensors exactly like the reference (scalars are 1-element
tensors), so the time grids and coefficients agree to rounding.

Written as one update routine per *family* (exponential-integrator coefficients are
computed once, then a linear combination of x and model values is formed) rather than
the reference's per-branch formulas; the golden vectors in tests/golden pin the two
against each other.
"""
import math

import torch


def interp1d(x, xp, yp):
    """Piecewise-linear y(x) through keypoints (xp[1,K] ascending, yp[1,K]); x is [N,1].
    Linear extrapolation from the outermost segment on both sides (ref :1296-1335)."""
    k = xp.shape[1]
    xs, ys = xp[0], yp[0]
    q = x[:, 0]
    # index of the left keypoint of the segment used for each query
    pos = torch.searchsorted(xs, q.contiguous(), right=False)   # #keypoints strictly below q
    left = (pos - 1).clamp(0, k - 2)
    x0, x1 = xs[left], xs[left + 1]
    y0, y1 = ys[left], ys[left + 1]
    return (y0 + (q - x0) * (y1 - y0) / (x1 - x0)).reshape(-1, 1)


class NoiseScheduleVP:
    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None,
                 continuous_beta_0=0.1, continuous_beta_1=20.0, dtype=torch.float32):
        if schedule not in ("discrete", "linear", "cosine"):
            raise ValueError(f"Unsupported noise schedule {schedule}. "
                             "The schedule needs to be 'discrete' or 'linear' or 'cosine'")
        self.schedule = schedule
        if schedule == "discrete":
            log_alphas = 0.5 * torch.log(1 - betas).cumsum(dim=0) if betas is not None \
                else 0.5 * torch.log(alphas_cumprod)
            self.total_N = len(log_alphas)
            self.T = 1.0
            self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].reshape(1, -1).to(dtype)
            self.log_alpha_array = log_alphas.reshape(1, -1).to(dtype)
        else:
            self.total_N = 1000
            self.beta_0, self.beta_1 = continuous_beta_0, continuous_beta_1
            self.cosine_s, self.cosine_beta_max = 0.008, 999.0
            self.cosine_t_max = math.atan(self.cosine_beta_max * (1 + self.cosine_s) / math.pi) * 2 \
                * (1 + self.cosine_s) / math.pi - self.cosine_s
            self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1 + self.cosine_s) * math.pi / 2))
            self.T = 0.9946 if schedule == "cosine" else 1.0

    def marginal_log_mean_coeff(self, t):
        if self.schedule == "discrete":
            return interp1d(t.reshape(-1, 1), self.t_array.to(t.device), self.log_alpha_array.to(t.device)).reshape(-1)
        if self.schedule == "linear":
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.log(torch.cos((t + self.cosine_s) / (1 + self.cosine_s) * math.pi / 2)) - self.cosine_log_alpha_0

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        la = self.marginal_log_mean_coeff(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))

    def inverse_lambda(self, lamb):
        zero = torch.zeros((1,)).to(lamb)
        if self.schedule == "linear":
            tmp = 2.0 * (self.beta_1 - self.beta_0) * torch.logaddexp(-2.0 * lamb, zero)
            return tmp / (torch.sqrt(self.beta_0 ** 2 + tmp) + self.beta_0) / (self.beta_1 - self.beta_0)
        log_alpha = -0.5 * torch.logaddexp(zero, -2.0 * lamb)
        if self.schedule == "discrete":
            return interp1d(log_alpha.reshape(-1, 1), torch.flip(self.log_alpha_array.to(lamb.device), [1]),
                            torch.flip(self.t_array.to(lamb.device), [1])).reshape(-1)
        return torch.arccos(torch.exp(log_alpha + self.cosine_log_alpha_0)) * 2 * (1 + self.cosine_s) / math.pi \
            - self.cosine_s


def bcast(v, dims):
    return v[(...,) + (None,) * (dims - 1)]


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond",
                  condition=None, unconditional_condition=None, guidance_scale=1.0, classifier_fn=None,
                  classifier_kwargs={}):
    assert model_type in ("noise", "x_start", "v", "score")
    assert guidance_type in ("uncond", "classifier", "classifier-free")
    ns = noise_schedule

    def model_time(t):
        return (t - 1.0 / ns.total_N) * 1000.0 if ns.schedule == "discrete" else t

    def noise_pred(x, t, cond=None):
        out = model(x, model_time(t), **model_kwargs) if cond is None else model(x, model_time(t), cond, **model_kwargs)
        if model_type == "noise":
            return out
        # The reference multiplies by alpha_t[B] / sigma_t[B] without expanding dims (:302-312), which only
        # works for B == 1 (all entries equal); the oracle expands them so that any batch size is valid.
        a, sd = bcast(ns.marginal_alpha(t), x.dim()), bcast(ns.marginal_std(t), x.dim())
        if model_type == "x_start":
            return (x - a * out) / sd
        if model_type == "v":
            return a * out + sd * x
        return -sd * out

    def model_fn(x, t):
        if guidance_type == "uncond":
            return noise_pred(x, t)
        if guidance_type == "classifier":
            assert classifier_fn is not None
            with torch.enable_grad():
                xg = x.detach().requires_grad_(True)
                grad = torch.autograd.grad(classifier_fn(xg, model_time(t), condition, **classifier_kwargs).sum(), xg)[0]
            return noise_pred(x, t) - guidance_scale * ns.marginal_std(t) * grad
        if guidance_scale == 1.0 or unconditional_condition is None:
            return noise_pred(x, t, cond=condition)
        e_un, e_c = noise_pred(torch.cat([x] * 2), torch.cat([t] * 2),
                               cond=torch.cat([unconditional_condition, condition])).chunk(2)
        return e_un + guidance_scale * (e_c - e_un)

    return model_fn


class DPM_Solver:
    def __init__(self, model_fn, noise_schedule, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1.0, dynamic_thresholding_ratio=0.995):
        assert algorithm_type in ("dpmsolver", "dpmsolver++")
        self.model = lambda x, t: model_fn(x, t.expand(x.shape[0]))
        self.noise_schedule = noise_schedule
        self.algorithm_type = algorithm_type
        self.correcting_x0_fn = self.dynamic_thresholding_fn if correcting_x0_fn == "dynamic_thresholding" \
            else correcting_x0_fn
        self.correcting_xt_fn = correcting_xt_fn
        self.dynamic_thresholding_ratio = dynamic_thresholding_ratio
        self.thresholding_max_val = thresholding_max_val
        self.nfe = 0
        self.trace = []          # (continuous t) of every model evaluation, for parity tests

    # -- model views ----------------------------------------------------------------------
    def dynamic_thresholding_fn(self, x0, t):
        s = torch.quantile(x0.abs().reshape(x0.shape[0], -1), self.dynamic_thresholding_ratio, dim=1)
        s = bcast(torch.maximum(s, self.thresholding_max_val * torch.ones_like(s)), x0.dim())
        return torch.clamp(x0, -s, s) / s

    def noise_prediction_fn(self, x, t):
        self.nfe += 1
        self.trace.append(float(t.reshape(-1)[0]))
        return self.model(x, t)

    def data_prediction_fn(self, x, t):
        ns = self.noise_schedule
        x0 = (x - ns.marginal_std(t) * self.noise_prediction_fn(x, t)) / ns.marginal_alpha(t)
        return self.correcting_x0_fn(x0, t) if self.correcting_x0_fn is not None else x0

    def model_fn(self, x, t):
        return self.data_prediction_fn(x, t) if self.algorithm_type == "dpmsolver++" else self.noise_prediction_fn(x, t)

    def denoise_to_zero_fn(self, x, s):
        return self.data_prediction_fn(x, s)

    # -- grids ------------------------------------------------------------------------------
    def get_time_steps(self, skip_type, t_T, t_0, N, device):
        ns = self.noise_schedule
        if skip_type == "logSNR":
            lam_T = ns.marginal_lambda(torch.tensor(t_T).to(device))
            lam_0 = ns.marginal_lambda(torch.tensor(t_0).to(device))
            return ns.inverse_lambda(torch.linspace(lam_T.cpu().item(), lam_0.cpu().item(), N + 1).to(device))
        if skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1).to(device)
        if skip_type == "time_quadratic":
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2).to(device)
        raise ValueError(f"Unsupported skip_type {skip_type}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'")

    def get_orders_and_timesteps_for_singlestep_solver(self, steps, order, skip_type, t_T, t_0, device):
        if order == 3:
            k = steps // 3 + 1
            orders = {0: [3] * (k - 2) + [2, 1], 1: [3] * (k - 1) + [1], 2: [3] * (k - 1) + [2]}[steps % 3]
        elif order == 2:
            k = steps // 2 + steps % 2
            orders = [2] * (steps // 2) + [1] * (steps % 2)
        elif order == 1:
            k, orders = 1, [1] * steps
        else:
            raise ValueError("'order' must be '1' or '2' or '3'.")
        if skip_type == "logSNR":
            outer = self.get_time_steps(skip_type, t_T, t_0, k, device)
        else:
            outer = self.get_time_steps(skip_type, t_T, t_0, steps, device)[
                torch.cumsum(torch.tensor([0] + orders), 0).to(device)]
        return outer, orders

    # -- exponential-integrator building blocks ----------------------------------------------
    def _lin(self, x, s, t, model_s, h=None):
        """First-order transfer s->t: returns (x_lin, g, h) with x_t(1st order) = x_lin and
        g = alpha_t*phi_1 (++) or sigma_t*phi_1, the factor of the higher-order differences.
        ``h`` = lambda_t - lambda_s; intermediate points pass their nominal r*h like the reference."""
        ns = self.noise_schedule
        h = ns.marginal_lambda(t) - ns.marginal_lambda(s) if h is None else h
        if self.algorithm_type == "dpmsolver++":
            g = torch.exp(ns.marginal_log_mean_coeff(t)) * torch.expm1(-h)
            return ns.marginal_std(t) / ns.marginal_std(s) * x - g * model_s, g, h
        g = ns.marginal_std(t) * torch.expm1(h)
        return torch.exp(ns.marginal_log_mean_coeff(t) - ns.marginal_log_mean_coeff(s)) * x - g * model_s, g, h

    def dpm_solver_first_update(self, x, s, t, model_s=None, return_intermediate=False):
        model_s = self.model_fn(x, s) if model_s is None else model_s
        x_t = self._lin(x, s, t, model_s)[0]
        return (x_t, {"model_s": model_s}) if return_intermediate else x_t

    def singlestep_dpm_solver_second_update(self, x, s, t, r1=0.5, model_s=None, return_intermediate=False,
                                            solver_type="dpmsolver"):
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError(f"'solver_type' must be either 'dpmsolver' or 'taylor', got {solver_type}")
        r1 = 0.5 if r1 is None else r1
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        lam_s = ns.marginal_lambda(s)
        h = ns.marginal_lambda(t) - lam_s
        s1 = ns.inverse_lambda(lam_s + r1 * h)
        model_s = self.model_fn(x, s) if model_s is None else model_s
        model_s1 = self.model_fn(self._lin(x, s, s1, model_s, r1 * h)[0], s1)
        x_lin, g, h = self._lin(x, s, t, model_s, h)
        if solver_type == "dpmsolver":
            x_t = x_lin - (0.5 / r1) * g * (model_s1 - model_s)
        else:
            coef = ns.marginal_alpha(t) * (torch.expm1(-h) / h + 1.0) if pp else -ns.marginal_std(t) * (torch.expm1(h) / h - 1.0)
            x_t = x_lin + (1.0 / r1) * coef * (model_s1 - model_s)
        return (x_t, {"model_s": model_s, "model_s1": model_s1}) if return_intermediate else x_t

    def singlestep_dpm_solver_third_update(self, x, s, t, r1=1.0 / 3.0, r2=2.0 / 3.0, model_s=None, model_s1=None,
                                           return_intermediate=False, solver_type="dpmsolver"):
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError(f"'solver_type' must be either 'dpmsolver' or 'taylor', got {solver_type}")
        r1 = 1.0 / 3.0 if r1 is None else r1
        r2 = 2.0 / 3.0 if r2 is None else r2
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        sgn = -1.0 if pp else 1.0                       # exponent sign: ++ integrates exp(-lambda)
        lam_s = ns.marginal_lambda(s)
        h = ns.marginal_lambda(t) - lam_s
        s1, s2 = ns.inverse_lambda(lam_s + r1 * h), ns.inverse_lambda(lam_s + r2 * h)
        amp = (lambda u: ns.marginal_alpha(u)) if pp else (lambda u: ns.marginal_std(u))
        phi_1 = torch.expm1(sgn * h)
        phi_22 = torch.expm1(sgn * r2 * h) / (r2 * h) - sgn
        phi_2 = phi_1 / h - sgn
        phi_3 = phi_2 / h - 0.5
        model_s = self.model_fn(x, s) if model_s is None else model_s
        if model_s1 is None:
            model_s1 = self.model_fn(self._lin(x, s, s1, model_s, r1 * h)[0], s1)
        x_s2 = self._lin(x, s, s2, model_s, r2 * h)[0] - sgn * (r2 / r1) * (amp(s2) * phi_22) * (model_s1 - model_s)
        model_s2 = self.model_fn(x_s2, s2)
        x_lin = self._lin(x, s, t, model_s, h)[0]
        if solver_type == "dpmsolver":
            x_t = x_lin - sgn * (1.0 / r2) * (amp(t) * phi_2) * (model_s2 - model_s)
        else:
            d10, d11 = (1.0 / r1) * (model_s1 - model_s), (1.0 / r2) * (model_s2 - model_s)
            d1 = (r2 * d10 - r1 * d11) / (r2 - r1)
            d2 = 2.0 * (d11 - d10) / (r2 - r1)
            x_t = x_lin - sgn * (amp(t) * phi_2) * d1 - (amp(t) * phi_3) * d2
        if return_intermediate:
            return x_t, {"model_s": model_s, "model_s1": model_s1, "model_s2": model_s2}
        return x_t

    def multistep_dpm_solver_second_update(self, x, model_prev_list, t_prev_list, t, solver_type="dpmsolver"):
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError(f"'solver_type' must be either 'dpmsolver' or 'taylor', got {solver_type}")
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        m1, m0 = model_prev_list[-2], model_prev_list[-1]
        l1, l0, lt = (ns.marginal_lambda(u) for u in (t_prev_list[-2], t_prev_list[-1], t))
        h0, h = l0 - l1, lt - l0
        d10 = (1.0 / (h0 / h)) * (m0 - m1)
        x_lin, g, _ = self._lin(x, t_prev_list[-1], t, m0)
        if solver_type == "dpmsolver":
            return x_lin - 0.5 * g * d10
        if pp:
            return x_lin + (ns.marginal_alpha(t) * (torch.expm1(-h) / h + 1.0)) * d10
        return x_lin - (ns.marginal_std(t) * (torch.expm1(h) / h - 1.0)) * d10

    def multistep_dpm_solver_third_update(self, x, model_prev_list, t_prev_list, t, solver_type="dpmsolver"):
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        sgn = -1.0 if pp else 1.0
        m2, m1, m0 = model_prev_list
        l2, l1, l0, lt = (ns.marginal_lambda(u) for u in (*t_prev_list, t))
        h1, h0, h = l1 - l2, l0 - l1, lt - l0
        r0, r1 = h0 / h, h1 / h
        d10, d11 = (1.0 / r0) * (m0 - m1), (1.0 / r1) * (m1 - m2)
        d1 = d10 + (r0 / (r0 + r1)) * (d10 - d11)
        d2 = (1.0 / (r0 + r1)) * (d10 - d11)
        amp_t = ns.marginal_alpha(t) if pp else ns.marginal_std(t)
        phi_1 = torch.expm1(sgn * h)
        phi_2 = phi_1 / h - sgn
        phi_3 = phi_2 / h - 0.5
        x_lin = self._lin(x, t_prev_list[-1], t, m0)[0]
        return x_lin - sgn * (amp_t * phi_2) * d1 - (amp_t * phi_3) * d2

    def singlestep_dpm_solver_update(self, x, s, t, order, return_intermediate=False, solver_type="dpmsolver",
                                     r1=None, r2=None):
        if order == 1:
            return self.dpm_solver_first_update(x, s, t, return_intermediate=return_intermediate)
        if order == 2:
            return self.singlestep_dpm_solver_second_update(x, s, t, return_intermediate=return_intermediate,
                                                            solver_type=solver_type, r1=r1)
        if order == 3:
            return self.singlestep_dpm_solver_third_update(x, s, t, return_intermediate=return_intermediate,
                                                           solver_type=solver_type, r1=r1, r2=r2)
        raise ValueError(f"Solver order must be 1 or 2 or 3, got {order}")

    def multistep_dpm_solver_update(self, x, model_prev_list, t_prev_list, t, order, solver_type="dpmsolver"):
        if order == 1:
            return self.dpm_solver_first_update(x, t_prev_list[-1], t, model_s=model_prev_list[-1])
        if order == 2:
            return self.multistep_dpm_solver_second_update(x, model_prev_list, t_prev_list, t, solver_type)
        if order == 3:
            return self.multistep_dpm_solver_third_update(x, model_prev_list, t_prev_list, t, solver_type)
        raise ValueError(f"Solver order must be 1 or 2 or 3, got {order}")

    def dpm_solver_adaptive(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5,
                            solver_type="dpmsolver"):
        ns = self.noise_schedule
        s = t_T * torch.ones((1,)).to(x)
        lam_s, lam_0 = ns.marginal_lambda(s), ns.marginal_lambda(t_0 * torch.ones_like(s))
        h = h_init * torch.ones_like(s)
        x_prev, nfe = x, 0
        if order == 2:
            lower = lambda x, s, t: self.dpm_solver_first_update(x, s, t, return_intermediate=True)
            higher = lambda x, s, t, **kw: self.singlestep_dpm_solver_second_update(x, s, t, r1=0.5,
                                                                                   solver_type=solver_type, **kw)
        elif order == 3:
            lower = lambda x, s, t: self.singlestep_dpm_solver_second_update(x, s, t, r1=1 / 3, return_intermediate=True,
                                                                            solver_type=solver_type)
            higher = lambda x, s, t, **kw: self.singlestep_dpm_solver_third_update(x, s, t, r1=1 / 3, r2=2 / 3,
                                                                                  solver_type=solver_type, **kw)
        else:
            raise ValueError(f"For adaptive step size solver, order must be 2 or 3, got {order}")
        while torch.abs(s - t_0).mean() > t_err:
            t = ns.inverse_lambda(lam_s + h)
            x_lo, kw = lower(x, s, t)
            x_hi = higher(x, s, t, **kw)
            delta = torch.max(torch.ones_like(x) * atol, rtol * torch.max(x_lo.abs(), x_prev.abs()))
            err = torch.sqrt(torch.square(((x_hi - x_lo) / delta).reshape(x.shape[0], -1)).mean(-1, keepdim=True)).max()
            if torch.all(err <= 1.0):
                x, s, x_prev = x_hi, t, x_lo
                lam_s = ns.marginal_lambda(s)
            h = torch.min(theta * h * torch.float_power(err, -1.0 / order).float(), lam_0 - lam_s)
            nfe += order
        return x

    def add_noise(self, x, t, noise=None):
        ns = self.noise_schedule
        noise = torch.randn((t.shape[0], *x.shape), device=x.device) if noise is None else noise
        x = x.reshape(-1, *x.shape)
        xt = bcast(ns.marginal_alpha(t), x.dim()) * x + bcast(ns.marginal_std(t), x.dim()) * noise
        return xt.squeeze(0) if t.shape[0] == 1 else xt

    def inverse(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
                lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
                return_intermediate=False):
        t_0 = 1.0 / self.noise_schedule.total_N if t_start is None else t_start
        t_T = self.noise_schedule.T if t_end is None else t_end
        return self.sample(x, steps=steps, t_start=t_0, t_end=t_T, order=order, skip_type=skip_type, method=method,
                           lower_order_final=lower_order_final, denoise_to_zero=denoise_to_zero,
                           solver_type=solver_type, atol=atol, rtol=rtol, return_intermediate=return_intermediate)

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
               return_intermediate=False):
        ns = self.noise_schedule
        t_0 = 1.0 / ns.total_N if t_end is None else t_end
        t_T = ns.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0
        fixed = method in ("multistep", "singlestep", "singlestep_fixed")
        assert fixed or not (return_intermediate or self.correcting_xt_fn is not None)
        device, inter = x.device, []
        fix = (lambda x, t, k: self.correcting_xt_fn(x, t, k)) if self.correcting_xt_fn is not None else (lambda x, t, k: x)
        with torch.no_grad():
            step = 0
            if method == "adaptive":
                x = self.dpm_solver_adaptive(x, order=order, t_T=t_T, t_0=t_0, atol=atol, rtol=rtol, solver_type=solver_type)
            elif method == "multistep":
                assert steps >= order
                ts = self.get_time_steps(skip_type, t_T, t_0, steps, device)
                t_prev, m_prev = [ts[0]], [self.model_fn(x, ts[0])]
                x = fix(x, ts[0], 0)
                inter.append(x)
                for step in range(1, order):
                    x = fix(self.multistep_dpm_solver_update(x, m_prev, t_prev, ts[step], step, solver_type), ts[step], step)
                    inter.append(x)
                    t_prev.append(ts[step])
                    m_prev.append(self.model_fn(x, ts[step]))
                for step in range(order, steps + 1):
                    k = min(order, steps + 1 - step) if (lower_order_final and steps < 10) else order
                    x = fix(self.multistep_dpm_solver_update(x, m_prev, t_prev, ts[step], k, solver_type), ts[step], step)
                    inter.append(x)
                    t_prev = t_prev[1:] + [ts[step]]
                    m_prev = m_prev[1:] + ([self.model_fn(x, ts[step])] if step < steps else [m_prev[-1]])
            elif method in ("singlestep", "singlestep_fixed"):
                if method == "singlestep":
                    outer, orders = self.get_orders_and_timesteps_for_singlestep_solver(steps, order, skip_type, t_T, t_0, device)
                else:
                    orders = [order] * (steps // order)
                    outer = self.get_time_steps(skip_type, t_T, t_0, steps // order, device)
                for step, k in enumerate(orders):
                    s, t = outer[step], outer[step + 1]
                    lam = ns.marginal_lambda(self.get_time_steps(skip_type, s.item(), t.item(), k, device))
                    hh = lam[-1] - lam[0]
                    r1 = None if k <= 1 else (lam[1] - lam[0]) / hh
                    r2 = None if k <= 2 else (lam[2] - lam[0]) / hh
                    x = fix(self.singlestep_dpm_solver_update(x, s, t, k, solver_type=solver_type, r1=r1, r2=r2), t, step)
                    inter.append(x)
            else:
                raise ValueError(f"Got wrong method {method}")
            if denoise_to_zero:
                t = torch.ones((1,)).to(device) * t_0
                x = fix(self.denoise_to_zero_fn(x, t), t, step + 1)
                inter.append(x)
        return (x, inter) if return_intermediate else x
