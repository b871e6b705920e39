"""Trainer wrapper ``DDPM`` -- the caller of the hot-path boundary.

Mirror of the reference's GeneralModel/model.py:14-140 + base_model.py + __init__.py::create_model: same methods
(feed_data, optimize_parameters, test, set_loss, set_new_noise_schedule, get_current_log, get_current_visuals,
save_network, load_network), same option keys, same checkpoint files (``I{iter}_gen.pth`` = GeneralDiffusion
state_dict, ``I{iter}_opt.pth`` = {iter, scheduler, optimizer}).  Differences: the LR schedule
(``transformers.get_scheduler("linear", warmup 100)``, model.py:32) is an equivalent LambdaLR so that transformers is
not required, and with more than one process the gradients are SUM-all-reduced over RCCL before the optimizer
step (``tmdiff_amd.dist.GradReducer``: flat buckets, all-reduces started from backward hooks so the exchange overlaps
the rest of backward) instead of wrapping the module in nn.DataParallel.
"""
import logging
import os
from collections import OrderedDict

import torch

from . import dist as tdist
from . import networks

logger = logging.getLogger("base")


def linear_warmup_decay(optimizer, num_warmup_steps, num_training_steps):
    """HF ``get_scheduler("linear")``: lr * step/warmup, then linear decay to 0 at num_training_steps."""
    def f(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, f)


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt["gpu_ids"] is not None else "cpu")
        self.begin_step = 0
        self.begin_epoch = 0

    def set_device(self, x):
        if isinstance(x, dict):
            for key, item in x.items():
                if item is not None and torch.is_tensor(item):
                    x[key] = item.to(self.device)
        elif isinstance(x, list):
            x = [item.to(self.device) if item is not None else None for item in x]
        else:
            x = x.to(self.device)
        return x

    def get_network_description(self, network):
        return str(network), sum(p.numel() for p in network.parameters())


class DDPM(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.optG = None
        self.scheduler = None
        self.netG = self.set_device(networks.define_General(opt))
        self.schedule_phase = None
        self.grad_reduce = (opt.get("train") or {}).get("grad_reduce", "sum") if isinstance(opt, dict) else "sum"
        self.reducer = None
        self.set_loss()
        if self.opt["phase"] == "train":
            self.netG.train()
            optim_params = [p for n, p in self.netG.named_parameters() if "clip_text" not in n]
            self.optG = torch.optim.AdamW(optim_params, lr=opt["train"]["optimizer"]["lr"], weight_decay=1e-4)
            self.scheduler = linear_warmup_decay(self.optG, 100, opt["train"]["max_iter"])
            self.log_dict = OrderedDict()
        self.load_network()
        self.print_network()

    def feed_data(self, data):
        self.data = self.set_device(data)

    def optimize_parameters(self, prompt=None):
        if self.reducer is None:       # created here: the process group exists by the first step
            self.reducer = tdist.GradReducer(self.netG, op=self.grad_reduce)
        l_pix = self.netG(self.data, prompt).sum()
        l_pix.backward()                # bucket all-reduces start from hooks while backward is still running
        self.reducer.finish()           # no-op in a single process
        self.optG.step()
        self.scheduler.step()
        self.reducer.zero_grad()
        self.log_dict["l_pix"] = l_pix.detach()
        self.log_dict["lr"] = self.optG.state_dict()["param_groups"][0]["lr"]

    def test(self, continous=False, prompt="QB", guidance=3.0):
        self.netG.eval()
        with torch.no_grad():
            self.SR = self.netG.super_resolution(self.data, continous, prompt, guidance)
        self.netG.train()

    def set_loss(self):
        self.netG.set_loss(self.device)

    def set_new_noise_schedule(self, schedule_opt, schedule_phase="train"):
        if self.schedule_phase is None or self.schedule_phase != schedule_phase:
            self.schedule_phase = schedule_phase
            self.netG.set_new_noise_schedule(schedule_opt, self.device)

    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self):
        out = OrderedDict()
        out["SR"] = self.SR.detach().float().cpu()
        for k in ("HR", "MS", "PAN", "LR"):
            if k in self.data:
                out[k] = self.data[k].detach().float().cpu()
        return out

    def print_network(self):
        s, n = self.get_network_description(self.netG)
        logger.info("Network G structure: {}, with parameters: {:,d}".format(self.netG.__class__.__name__, n))
        logger.info(s)

    def save_network(self, iter_step):
        ckpt = self.opt["path"]["checkpoint"]
        os.makedirs(ckpt, exist_ok=True)
        gen_path = os.path.join(ckpt, "I{}_gen.pth".format(iter_step))
        opt_path = os.path.join(ckpt, "I{}_opt.pth".format(iter_step))
        torch.save({k: v.cpu() for k, v in self.netG.state_dict().items()}, gen_path)
        torch.save({"iter": iter_step, "scheduler": self.scheduler.state_dict(), "optimizer": self.optG.state_dict()},
                   opt_path)
        logger.info("Saved model in [{:s}] ...".format(gen_path))

    def load_network(self):
        load_path = (self.opt.get("path") or {}).get("resume") if isinstance(self.opt, dict) else self.opt["path"]["resume"]
        if load_path is not None:
            logger.info("Loading pretrained model for G [{:s}] ...".format(load_path))
            self.netG.load_state_dict(torch.load("{}_gen.pth".format(load_path), map_location="cpu"), strict=False)
            if self.opt["phase"] == "train":
                self.begin_step = torch.load("{}_opt.pth".format(load_path), map_location="cpu")["iter"]


def create_model(opt):
    m = DDPM(opt)
    logger.info("Model [{:s}] is created.".format(m.__class__.__name__))
    return m


class EmaUpdater:
    """EMA of ``netG.denoise_fn`` parameters (reference utils/EmaUpdater.py:4-65), decay 0.9999; the update is one
    fused HIP axpby per tensor when the parameters live on the GPU."""

    def __init__(self, model, ema_model, decay=0.9999, start_iter=0):
        self.model, self.ema_model = model, ema_model
        self.decay, self.start_iter, self.iteration = decay, start_iter, start_iter
        self._mt = None          # ops.MultiAxpby over all parameter pairs (built on the first update)

    @torch.no_grad()
    def update(self, iteration):
        self.iteration = iteration
        pairs = list(zip(self.model.netG.denoise_fn.parameters(), self.ema_model.netG.denoise_fn.parameters()))
        if iteration > self.start_iter:
            gpu = [(p, pe) for p, pe in pairs if p.is_cuda]
            if gpu:       # ONE multi-tensor launch: p_ema = decay * p_ema + (1 - decay) * p for every tensor
                from . import ops
                triples = [(pe.data, pe.data, p.data) for p, pe in gpu]
                if self._mt is None or self._mt.stale(triples):
                    self._mt = ops.MultiAxpby(triples)
                self._mt.run(self.decay, 1.0 - self.decay)
            for p, p_ema in pairs:
                if not p.is_cuda:
                    p_ema.data.mul_(self.decay).add_(p.data, alpha=1.0 - self.decay)
        else:
            for p, p_ema in pairs:
                p_ema.data.copy_(p.data)
        # the writes above go through .data / a raw kernel and do not bump Parameter._version, which is what WavBEST
        # keys its packed convolution weights and projection banks on: drop them explicitly
        net = self.ema_model.netG.denoise_fn
        if hasattr(net, "invalidate_prepared"):
            net.invalidate_prepared()

    def load_ema_params(self):
        self.model.netG.denoise_fn.load_state_dict(self.ema_model.netG.denoise_fn.state_dict())

    def load_model_params(self):
        self.ema_model.netG.denoise_fn.load_state_dict(self.model.netG.denoise_fn.state_dict())

    @property
    def on_fly_model_state_dict(self):
        return self.model.netG.denoise_fn.state_dict()

    @property
    def ema_model_state_dict(self):
        return self.ema_model.netG.denoise_fn.state_dict()
