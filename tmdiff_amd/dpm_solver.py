"""DPM-Solver / DPM-Solver++ for the HIP sampling path.

API mirror of the reference's core/dpm_solver_pytorch.py (``NoiseScheduleVP`` :6-181,
``model_wrapper`` :184-348, ``DPM_Solver`` :351-1289).  Division of labour:

  * everything that is scalar per step -- log-alpha interpolation, lambda and its inverse, time
    grids, order plans, exponential-integrator coefficients -- is evaluated on the HOST in fp32
    (1-element CPU tensors, the same arithmetic the reference performs on its 1-element tensors);
  * everything that touches an image tensor is a launch of libtmdiff_hip.so:
    ``tmdiff_axpby`` for the linear-combination updates (:563-927), ``tmdiff_x0_from_model`` for
    the x_start->noise->x0 conversion (:302-306, :447-456) and ``tmdiff_abs_quantile_clamp`` for
    dynamic thresholding (:430-439).

One deliberate difference: the reference multiplies by alpha_t[B] / sigma_t[B] without expanding
dims (:302-312), which breaks for batch sizes other than 1; here the time is one scalar per call,
so any batch size works and B = 1 reproduces the reference.
"""
import math

import torch

from . import ops


def _interp(x, xp, yp):
    """Piecewise-linear f(x) through ascending keypoints, linear extrapolation outside (:1296-1335).
    x: [N] fp32 CPU tensor; xp, yp: [K]."""
    k = xp.shape[0]
    left = (torch.searchsorted(xp, x.contiguous()) - 1).clamp(0, k - 2)
    x0, x1, y0, y1 = xp[left], xp[left + 1], yp[left], yp[left + 1]
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0)


class NoiseScheduleVP:
    """Host-side VP noise schedule.  All methods take and return fp32 CPU tensors."""

    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20.0, dtype=torch.float32):
        if schedule not in ("discrete", "linear", "cosine"):
            raise ValueError("Unsupported noise schedule {}. The schedule needs to be 'discrete' or 'linear' or "
                             "'cosine'".format(schedule))
        self.schedule = schedule
        if schedule == "discrete":
            src = betas if betas is not None else alphas_cumprod
            assert src is not None
            src = src.detach().to("cpu", torch.float32)
            log_alphas = 0.5 * torch.log(1 - src).cumsum(dim=0) if betas is not None else 0.5 * torch.log(src)
            self.total_N = len(log_alphas)
            self.T = 1.0
            self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].to(dtype)
            self.log_alpha_array = log_alphas.to(dtype)
            self._rev_la = torch.flip(self.log_alpha_array, [0]).contiguous()
            self._rev_t = torch.flip(self.t_array, [0]).contiguous()
        else:
            self.total_N = 1000
            self.beta_0, self.beta_1 = continuous_beta_0, continuous_beta_1
            self.cosine_s, self.cosine_beta_max = 0.008, 999.0
            self.cosine_t_max = math.atan(self.cosine_beta_max * (1.0 + self.cosine_s) / math.pi) * 2.0 * (
                1.0 + self.cosine_s) / math.pi - self.cosine_s
            self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1.0 + self.cosine_s) * math.pi / 2.0))
            self.T = 0.9946 if schedule == "cosine" else 1.0

    @staticmethod
    def _cpu(t):
        t = torch.as_tensor(t, dtype=torch.float32)
        return t.detach().cpu().reshape(-1)

    def marginal_log_mean_coeff(self, t):
        t = self._cpu(t)
        if self.schedule == "discrete":
            return _interp(t, self.t_array, self.log_alpha_array)
        if self.schedule == "linear":
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.log(torch.cos((t + self.cosine_s) / (1.0 + self.cosine_s) * math.pi / 2.0)) - self.cosine_log_alpha_0

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        la = self.marginal_log_mean_coeff(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))

    def inverse_lambda(self, lamb):
        lamb = self._cpu(lamb)
        zero = torch.zeros(1)
        if self.schedule == "linear":
            tmp = 2.0 * (self.beta_1 - self.beta_0) * torch.logaddexp(-2.0 * lamb, zero)
            return tmp / (torch.sqrt(self.beta_0 ** 2 + tmp) + self.beta_0) / (self.beta_1 - self.beta_0)
        log_alpha = -0.5 * torch.logaddexp(zero, -2.0 * lamb)
        if self.schedule == "discrete":
            return _interp(log_alpha, self._rev_la, self._rev_t)
        return torch.arccos(torch.exp(log_alpha + self.cosine_log_alpha_0)) * 2.0 * (1.0 + self.cosine_s) / math.pi \
            - self.cosine_s


class _WrappedModel:
    """Callable returned by :func:`model_wrapper`; keeps the pieces so the solver can fuse the
    x_start -> x0 conversion into one kernel."""

    def __init__(self, model, ns, model_type, model_kwargs, guidance_type, condition, unconditional_condition,
                 guidance_scale, classifier_fn, classifier_kwargs):
        self.model, self.ns, self.model_type, self.model_kwargs = model, ns, model_type, model_kwargs
        self.guidance_type, self.condition, self.unconditional_condition = guidance_type, condition, unconditional_condition
        self.guidance_scale, self.classifier_fn, self.classifier_kwargs = guidance_scale, classifier_fn, classifier_kwargs

    def model_time(self, t, batch, device):
        """continuous t (host scalar) -> the [B] time tensor the network expects (:285-294)."""
        t = NoiseScheduleVP._cpu(t)
        t_in = (t - 1.0 / self.ns.total_N) * 1000.0 if self.ns.schedule == "discrete" else t
        return t_in.expand(batch).to(device)

    def raw(self, x, t, cond=None):
        t_in = self.model_time(t, x.shape[0], x.device)
        if cond is None:
            return self.model(x, t_in, **self.model_kwargs)
        return self.model(x, t_in, cond, **self.model_kwargs)

    def noise_pred(self, x, t, cond=None):
        out = self.raw(x, t, cond)
        if self.model_type == "noise":
            return out
        a, s = float(self.ns.marginal_alpha(t)[0]), float(self.ns.marginal_std(t)[0])
        if self.model_type == "x_start":
            return ops.axpby([x, out], [1.0 / s, -a / s])
        if self.model_type == "v":
            return ops.axpby([out, x], [a, s])
        return ops.axpby([out], [-s])          # "score"

    def __call__(self, x, t):
        if self.guidance_type == "uncond":
            return self.noise_pred(x, t)
        if self.guidance_type == "classifier":
            assert self.classifier_fn is not None
            t_in = self.model_time(t, x.shape[0], x.device)
            with torch.enable_grad():
                xg = x.detach().requires_grad_(True)
                lp = self.classifier_fn(xg, t_in, self.condition, **self.classifier_kwargs)
                grad = torch.autograd.grad(lp.sum(), xg)[0]
            s = float(self.ns.marginal_std(t)[0])
            return ops.axpby([self.noise_pred(x, t), grad.contiguous()], [1.0, -self.guidance_scale * s])
        if self.guidance_scale == 1.0 or self.unconditional_condition is None:
            return self.noise_pred(x, t, cond=self.condition)
        both = self.noise_pred(torch.cat([x] * 2), t, cond=torch.cat([self.unconditional_condition, self.condition]))
        e_un, e_c = both[: x.shape[0]].contiguous(), both[x.shape[0]:].contiguous()
        return ops.axpby([e_un, e_c], [1.0 - self.guidance_scale, self.guidance_scale])


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond", condition=None,
                  unconditional_condition=None, guidance_scale=1.0, classifier_fn=None, classifier_kwargs={}):
    assert model_type in ["noise", "x_start", "v", "score"]
    assert guidance_type in ["uncond", "classifier", "classifier-free"]
    return _WrappedModel(model, noise_schedule, model_type, model_kwargs, guidance_type, condition,
                         unconditional_condition, guidance_scale, classifier_fn, classifier_kwargs)


class DPM_Solver:
    def __init__(self, model_fn, noise_schedule, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1.0, dynamic_thresholding_ratio=0.995):
        assert algorithm_type in ["dpmsolver", "dpmsolver++"]
        self.model = model_fn
        self.noise_schedule = noise_schedule
        self.algorithm_type = algorithm_type
        self._dynamic = correcting_x0_fn == "dynamic_thresholding"
        self.correcting_x0_fn = self.dynamic_thresholding_fn if self._dynamic else correcting_x0_fn
        self.correcting_xt_fn = correcting_xt_fn
        self.dynamic_thresholding_ratio = dynamic_thresholding_ratio
        self.thresholding_max_val = thresholding_max_val
        self.nfe = 0
        self.trace = []      # continuous time of every network evaluation

    # ---- model views --------------------------------------------------------------------------------
    def dynamic_thresholding_fn(self, x0, t=None):
        x0 = x0.clone() if not x0.is_contiguous() else x0
        ops.abs_quantile_clamp_(x0, self.dynamic_thresholding_ratio, self.thresholding_max_val)
        return x0

    def _count(self, t):
        self.nfe += 1
        self.trace.append(float(NoiseScheduleVP._cpu(t)[0]))

    def noise_prediction_fn(self, x, t):
        self._count(t)
        return self.model(x, t)

    def data_prediction_fn(self, x, t):
        ns = self.noise_schedule
        a, s = float(ns.marginal_alpha(t)[0]), float(ns.marginal_std(t)[0])
        m = self.model
        if isinstance(m, _WrappedModel) and m.guidance_type == "uncond" and m.model_type in ("x_start", "noise"):
            self._count(t)       # one kernel: x_start -> noise -> x0 in the reference's rounding order
            x0 = ops.x0_from_model(x, m.raw(x, t), a, s, model_is_x_start=m.model_type == "x_start")
        else:
            x0 = ops.x0_from_model(x, self.noise_prediction_fn(x, t), a, s, model_is_x_start=False)
        return self.correcting_x0_fn(x0, t) if self.correcting_x0_fn is not None else x0

    def model_fn(self, x, t):
        return self.data_prediction_fn(x, t) if self.algorithm_type == "dpmsolver++" else self.noise_prediction_fn(x, t)

    def denoise_to_zero_fn(self, x, s):
        return self.data_prediction_fn(x, s)

    # ---- grids ----------------------------------------------------------------------------------------
    def get_time_steps(self, skip_type, t_T, t_0, N, device=None):
        ns = self.noise_schedule
        if skip_type == "logSNR":
            lam_T, lam_0 = ns.marginal_lambda(torch.tensor(t_T)), ns.marginal_lambda(torch.tensor(t_0))
            return ns.inverse_lambda(torch.linspace(lam_T.item(), lam_0.item(), N + 1))
        if skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        if skip_type == "time_quadratic":
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
        raise ValueError("Unsupported skip_type {}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'"
                         .format(skip_type))

    def get_orders_and_timesteps_for_singlestep_solver(self, steps, order, skip_type, t_T, t_0, device=None):
        if order == 3:
            k = steps // 3 + 1
            orders = [[3] * (k - 2) + [2, 1], [3] * (k - 1) + [1], [3] * (k - 1) + [2]][steps % 3]
        elif order == 2:
            k = steps // 2 + steps % 2
            orders = [2] * (steps // 2) + [1] * (steps % 2)
        elif order == 1:
            k, orders = 1, [1] * steps
        else:
            raise ValueError("'order' must be '1' or '2' or '3'.")
        if skip_type == "logSNR":
            outer = self.get_time_steps(skip_type, t_T, t_0, k)
        else:
            outer = self.get_time_steps(skip_type, t_T, t_0, steps)[torch.cumsum(torch.tensor([0] + orders), 0)]
        return outer, orders

    # ---- exponential-integrator pieces: host coefficients, device axpby ---------------------------------
    def _coef(self, s, t, h=None):
        """(c_x, g, h): first-order transfer x_t = c_x * x - g * model_s; g also scales the corrections."""
        ns = self.noise_schedule
        h = ns.marginal_lambda(t) - ns.marginal_lambda(s) if h is None else h
        if self.algorithm_type == "dpmsolver++":
            return ns.marginal_std(t) / ns.marginal_std(s), torch.exp(ns.marginal_log_mean_coeff(t)) * torch.expm1(-h), h
        return (torch.exp(ns.marginal_log_mean_coeff(t) - ns.marginal_log_mean_coeff(s)),
                ns.marginal_std(t) * torch.expm1(h), h)

    @staticmethod
    def _f(v):
        return float(v.reshape(-1)[0]) if torch.is_tensor(v) else float(v)

    def _lin(self, x, s, t, model_s, h=None, extra=()):
        """x_t = c_x*x - g*model_s + sum_k coef_k * tensor_k  in one launch (left-to-right accumulation)."""
        c_x, g, _ = self._coef(s, t, h)
        tensors, coefs = [x, model_s], [self._f(c_x), -self._f(g)]
        for c, ten in extra:
            tensors.append(ten)
            coefs.append(self._f(c))
        return ops.axpby(tensors, coefs)

    def dpm_solver_first_update(self, x, s, t, model_s=None, return_intermediate=False):
        model_s = self.model_fn(x, s) if model_s is None else model_s
        x_t = self._lin(x, s, t, model_s)
        return (x_t, {"model_s": model_s}) if return_intermediate else x_t

    def singlestep_dpm_solver_second_update(self, x, s, t, r1=0.5, model_s=None, return_intermediate=False,
                                            solver_type="dpmsolver"):
        if solver_type not in ["dpmsolver", "taylor"]:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        r1 = 0.5 if r1 is None else r1
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        lam_s = ns.marginal_lambda(s)
        h = ns.marginal_lambda(t) - lam_s
        s1 = ns.inverse_lambda(lam_s + r1 * h)
        model_s = self.model_fn(x, s) if model_s is None else model_s
        model_s1 = self.model_fn(self._lin(x, s, s1, model_s, r1 * h), s1)
        diff = ops.axpby([model_s1, model_s], [1.0, -1.0])
        _, g, _ = self._coef(s, t, h)
        if solver_type == "dpmsolver":
            c = -(0.5 / r1) * g
        elif pp:
            c = (1.0 / r1) * (ns.marginal_alpha(t) * (torch.expm1(-h) / h + 1.0))
        else:
            c = -(1.0 / r1) * (ns.marginal_std(t) * (torch.expm1(h) / h - 1.0))
        x_t = self._lin(x, s, t, model_s, h, extra=[(c, diff)])
        return (x_t, {"model_s": model_s, "model_s1": model_s1}) if return_intermediate else x_t

    def singlestep_dpm_solver_third_update(self, x, s, t, r1=1.0 / 3.0, r2=2.0 / 3.0, model_s=None, model_s1=None,
                                           return_intermediate=False, solver_type="dpmsolver"):
        if solver_type not in ["dpmsolver", "taylor"]:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        r1 = 1.0 / 3.0 if r1 is None else r1
        r2 = 2.0 / 3.0 if r2 is None else r2
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        sgn = -1.0 if pp else 1.0
        amp = ns.marginal_alpha if pp else ns.marginal_std
        lam_s = ns.marginal_lambda(s)
        h = ns.marginal_lambda(t) - lam_s
        s1, s2 = ns.inverse_lambda(lam_s + r1 * h), ns.inverse_lambda(lam_s + r2 * h)
        phi_1 = torch.expm1(sgn * h)
        phi_22 = torch.expm1(sgn * r2 * h) / (r2 * h) - sgn
        phi_2 = phi_1 / h - sgn
        phi_3 = phi_2 / h - 0.5
        model_s = self.model_fn(x, s) if model_s is None else model_s
        if model_s1 is None:
            model_s1 = self.model_fn(self._lin(x, s, s1, model_s, r1 * h), s1)
        d1 = ops.axpby([model_s1, model_s], [1.0, -1.0])
        x_s2 = self._lin(x, s, s2, model_s, r2 * h, extra=[(-sgn * (r2 / r1) * (amp(s2) * phi_22), d1)])
        model_s2 = self.model_fn(x_s2, s2)
        d2 = ops.axpby([model_s2, model_s], [1.0, -1.0])
        if solver_type == "dpmsolver":
            x_t = self._lin(x, s, t, model_s, h, extra=[(-sgn * (1.0 / r2) * (amp(t) * phi_2), d2)])
        else:
            # D1 = (r2*D1_0 - r1*D1_1)/(r2-r1), D2 = 2*(D1_1 - D1_0)/(r2-r1), D1_k = diff_k / r_k
            i1, i2, den = 1.0 / r1, 1.0 / r2, (r2 - r1)
            big_d1 = ops.axpby([d1, d2], [self._f(r2 * i1 / den), -self._f(r1 * i2 / den)])
            big_d2 = ops.axpby([d2, d1], [self._f(2.0 * i2 / den), -self._f(2.0 * i1 / den)])
            x_t = self._lin(x, s, t, model_s, h, extra=[(-sgn * (amp(t) * phi_2), big_d1), (-(amp(t) * phi_3), big_d2)])
        if return_intermediate:
            return x_t, {"model_s": model_s, "model_s1": model_s1, "model_s2": model_s2}
        return x_t

    def multistep_dpm_solver_second_update(self, x, model_prev_list, t_prev_list, t, solver_type="dpmsolver"):
        if solver_type not in ["dpmsolver", "taylor"]:
            raise ValueError("'solver_type' must be either 'dpmsolver' or 'taylor', got {}".format(solver_type))
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        m1, m0 = model_prev_list[-2], model_prev_list[-1]
        l1, l0, lt = (ns.marginal_lambda(u) for u in (t_prev_list[-2], t_prev_list[-1], t))
        h0, h = l0 - l1, lt - l0
        inv_r0 = 1.0 / (h0 / h)
        d10 = ops.axpby([m0, m1], [self._f(inv_r0), -self._f(inv_r0)])
        _, g, _ = self._coef(t_prev_list[-1], t)
        if solver_type == "dpmsolver":
            c = -0.5 * g
        elif pp:
            c = ns.marginal_alpha(t) * (torch.expm1(-h) / h + 1.0)
        else:
            c = -(ns.marginal_std(t) * (torch.expm1(h) / h - 1.0))
        return self._lin(x, t_prev_list[-1], t, m0, extra=[(c, d10)])

    def multistep_dpm_solver_third_update(self, x, model_prev_list, t_prev_list, t, solver_type="dpmsolver"):
        ns, pp = self.noise_schedule, self.algorithm_type == "dpmsolver++"
        sgn = -1.0 if pp else 1.0
        m2, m1, m0 = model_prev_list
        l2, l1, l0, lt = (ns.marginal_lambda(u) for u in (*t_prev_list, t))
        h1, h0, h = l1 - l2, l0 - l1, lt - l0
        r0, r1 = h0 / h, h1 / h
        d10 = ops.axpby([m0, m1], [self._f(1.0 / r0), -self._f(1.0 / r0)])
        d11 = ops.axpby([m1, m2], [self._f(1.0 / r1), -self._f(1.0 / r1)])
        w = r0 / (r0 + r1)
        big_d1 = ops.axpby([d10, d10, d11], [1.0, self._f(w), -self._f(w)])
        big_d2 = ops.axpby([d10, d11], [self._f(1.0 / (r0 + r1)), -self._f(1.0 / (r0 + r1))])
        amp_t = ns.marginal_alpha(t) if pp else ns.marginal_std(t)
        phi_1 = torch.expm1(sgn * h)
        phi_2 = phi_1 / h - sgn
        phi_3 = phi_2 / h - 0.5
        return self._lin(x, t_prev_list[-1], t, m0, extra=[(-sgn * (amp_t * phi_2), big_d1), (-(amp_t * phi_3), big_d2)])

    def singlestep_dpm_solver_update(self, x, s, t, order, return_intermediate=False, solver_type="dpmsolver", r1=None,
                                     r2=None):
        if order == 1:
            return self.dpm_solver_first_update(x, s, t, return_intermediate=return_intermediate)
        if order == 2:
            return self.singlestep_dpm_solver_second_update(x, s, t, return_intermediate=return_intermediate,
                                                            solver_type=solver_type, r1=r1)
        if order == 3:
            return self.singlestep_dpm_solver_third_update(x, s, t, return_intermediate=return_intermediate,
                                                           solver_type=solver_type, r1=r1, r2=r2)
        raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))

    def multistep_dpm_solver_update(self, x, model_prev_list, t_prev_list, t, order, solver_type="dpmsolver"):
        if order == 1:
            return self.dpm_solver_first_update(x, t_prev_list[-1], t, model_s=model_prev_list[-1])
        if order == 2:
            return self.multistep_dpm_solver_second_update(x, model_prev_list, t_prev_list, t, solver_type=solver_type)
        if order == 3:
            return self.multistep_dpm_solver_third_update(x, model_prev_list, t_prev_list, t, solver_type=solver_type)
        raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))

    def dpm_solver_adaptive(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5,
                            solver_type="dpmsolver"):
        """Adaptive step size solver (:982-1043).  The error norm is a reduction over the image, done with
        torch reductions on the device (control flow needs it on the host every step anyway)."""
        ns = self.noise_schedule
        s = torch.tensor([t_T], dtype=torch.float32)
        lam_s, lam_0 = ns.marginal_lambda(s), ns.marginal_lambda(torch.tensor([t_0], dtype=torch.float32))
        h = torch.tensor([h_init], dtype=torch.float32)
        x_prev, nfe = x, 0
        if order == 2:
            lower = lambda x, s, t: self.dpm_solver_first_update(x, s, t, return_intermediate=True)
            higher = lambda x, s, t, **kw: self.singlestep_dpm_solver_second_update(x, s, t, r1=0.5,
                                                                                   solver_type=solver_type, **kw)
        elif order == 3:
            lower = lambda x, s, t: self.singlestep_dpm_solver_second_update(x, s, t, r1=1.0 / 3.0,
                                                                            return_intermediate=True,
                                                                            solver_type=solver_type)
            higher = lambda x, s, t, **kw: self.singlestep_dpm_solver_third_update(x, s, t, r1=1.0 / 3.0, r2=2.0 / 3.0,
                                                                                  solver_type=solver_type, **kw)
        else:
            raise ValueError("For adaptive step size solver, order must be 2 or 3, got {}".format(order))
        while torch.abs(s - t_0).mean() > t_err:
            t = ns.inverse_lambda(lam_s + h)
            x_lower, kw = lower(x, s, t)
            x_higher = higher(x, s, t, **kw)
            delta = torch.max(torch.ones_like(x) * atol, rtol * torch.max(torch.abs(x_lower), torch.abs(x_prev)))
            err = torch.sqrt(torch.square(((x_higher - x_lower) / delta).reshape(x.shape[0], -1)).mean(dim=-1)).max().cpu()
            if torch.all(err <= 1.0):
                x, s, x_prev = x_higher, t, x_lower
                lam_s = ns.marginal_lambda(s)
            h = torch.min(theta * h * torch.float_power(err, -1.0 / order).float(), lam_0 - lam_s)
            nfe += order
        return x

    def add_noise(self, x, t, noise=None):
        ns = self.noise_schedule
        t = NoiseScheduleVP._cpu(t)
        alpha, sigma = ns.marginal_alpha(t), ns.marginal_std(t)
        if noise is None:
            noise = torch.randn((t.shape[0], *x.shape), device=x.device)
        outs = [ops.axpby([x.contiguous(), noise[k].contiguous()], [float(alpha[k]), float(sigma[k])])
                for k in range(t.shape[0])]
        return outs[0] if t.shape[0] == 1 else torch.stack(outs)

    def inverse(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
                lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
                return_intermediate=False):
        t_0 = 1.0 / self.noise_schedule.total_N if t_start is None else t_start
        t_T = self.noise_schedule.T if t_end is None else t_end
        assert t_0 > 0 and t_T > 0
        return self.sample(x, steps=steps, t_start=t_0, t_end=t_T, order=order, skip_type=skip_type, method=method,
                           lower_order_final=lower_order_final, denoise_to_zero=denoise_to_zero,
                           solver_type=solver_type, atol=atol, rtol=rtol, return_intermediate=return_intermediate)

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
               return_intermediate=False):
        ns = self.noise_schedule
        t_0 = 1.0 / ns.total_N if t_end is None else t_end
        t_T = ns.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0, "Time range needs to be greater than 0. For discrete-time DPMs, it needs to be " \
                                    "in [1 / N, 1], where N is the length of betas array"
        fixed = method in ["multistep", "singlestep", "singlestep_fixed"]
        if return_intermediate:
            assert fixed, "Cannot use adaptive solver when saving intermediate values"
        if self.correcting_xt_fn is not None:
            assert fixed, "Cannot use adaptive solver when correcting_xt_fn is not None"
        fix = self.correcting_xt_fn if self.correcting_xt_fn is not None else (lambda x, t, k: x)
        x = x.contiguous()
        inter = []
        step = 0
        with torch.no_grad():
            if method == "adaptive":
                x = self.dpm_solver_adaptive(x, order=order, t_T=t_T, t_0=t_0, atol=atol, rtol=rtol,
                                             solver_type=solver_type)
            elif method == "multistep":
                assert steps >= order
                ts = self.get_time_steps(skip_type, t_T, t_0, steps)
                assert ts.shape[0] - 1 == steps
                t_prev, m_prev = [ts[0:1]], [self.model_fn(x, ts[0:1])]
                x = fix(x, ts[0:1], 0)
                inter.append(x)
                for step in range(1, order):
                    t = ts[step:step + 1]
                    x = fix(self.multistep_dpm_solver_update(x, m_prev, t_prev, t, step, solver_type=solver_type), t, step)
                    inter.append(x)
                    t_prev.append(t)
                    m_prev.append(self.model_fn(x, t))
                for step in range(order, steps + 1):
                    t = ts[step:step + 1]
                    k = min(order, steps + 1 - step) if (lower_order_final and steps < 10) else order
                    x = fix(self.multistep_dpm_solver_update(x, m_prev, t_prev, t, k, solver_type=solver_type), t, step)
                    inter.append(x)
                    t_prev = t_prev[1:] + [t]
                    m_prev = m_prev[1:] + ([self.model_fn(x, t)] if step < steps else [m_prev[-1]])
            elif method in ["singlestep", "singlestep_fixed"]:
                if method == "singlestep":
                    outer, orders = self.get_orders_and_timesteps_for_singlestep_solver(steps, order, skip_type, t_T, t_0)
                else:
                    orders = [order] * (steps // order)
                    outer = self.get_time_steps(skip_type, t_T, t_0, steps // order)
                for step, k in enumerate(orders):
                    s, t = outer[step:step + 1], outer[step + 1:step + 2]
                    lam = ns.marginal_lambda(self.get_time_steps(skip_type, s.item(), t.item(), k))
                    hh = lam[-1] - lam[0]
                    r1 = None if k <= 1 else (lam[1] - lam[0]) / hh
                    r2 = None if k <= 2 else (lam[2] - lam[0]) / hh
                    x = fix(self.singlestep_dpm_solver_update(x, s, t, k, solver_type=solver_type, r1=r1, r2=r2), t, step)
                    inter.append(x)
            else:
                raise ValueError("Got wrong method {}".format(method))
            if denoise_to_zero:
                t = torch.ones(1) * t_0
                x = fix(self.denoise_to_zero_fn(x, t), t, step + 1)
                inter.append(x)
        return (x, inter) if return_intermediate else x
