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
from . import networks, ops

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


class CapturedStep:
    """The finetune step of reference model.py:40-47 -- p_losses_dynamic (diffusion_general.py:349-370) forward, backward,
    AdamW -- for ONE (prompt, batch shape) recorded into a HIP graph: ~700 kernel launches become one graph launch.

    Everything the recorded launches read lives at fixed device addresses: the batch (`Res`, `PAN`, `MS`), the timesteps
    `t` and their sqrt(alpha_bar) values `a`, the noise, the parameters, the gradients (allocated inside the capture, from
    the graph's private pool), the AdamW state (tmdiff_amd.optim.FusedAdamW: step counter and learning rate are device scalars)
    and the per-step dropout word (ops.DROP_WORD, bumped by the graph itself).  Per step the host only (1) copies the new
    batch into the static tensors, (2) draws the timesteps from NumPy's RNG exactly as the reference does and copies them
    and their table values in (two small pinned, non-blocking copies), (3) fills the noise tensor, (4) launches the graph.
    The recorded step is the eager one launch for launch (same kernels, same order, same arithmetic)."""

    def __init__(self, trainer, data, prompt):
        self.trainer, self.prompt = trainer, prompt
        dev = trainer.device
        self.static = {k: torch.empty_like(data[k], dtype=torch.float32, memory_format=torch.contiguous_format)
                       for k in ("Res", "PAN", "MS")}
        b = data["Res"].shape[0]
        self.t = torch.ones(b, 1, dtype=torch.int64, device=dev)
        self.a = torch.ones(b, dtype=torch.float32, device=dev)
        self.noise = torch.empty_like(self.static["Res"])
        self.graph = torch.cuda.CUDAGraph()
        self.loss = None
        self.replays = 0

    def load(self, data):
        """Host side of a step: the batch, the host-drawn timesteps and the noise into the static tensors."""
        diff = self.trainer.netG
        for k, dst in self.static.items():
            dst.copy_(data[k], non_blocking=True)
        t_host, a_host = diff.draw_training_inputs(self.t.shape[0])
        self.t.copy_(t_host.pin_memory(), non_blocking=True)
        self.a.copy_(a_host.pin_memory(), non_blocking=True)
        if diff.noise_fn is not None:
            self.noise.copy_(diff.noise_fn(self.noise).to(self.noise.device))
        else:
            self.noise.normal_()

    def _step(self):
        tr = self.trainer
        ops.DROP_WORD.add_(1)                  # a fresh dropout mask per replay: the kernels add this word to their seeds
        loss = tr.netG.p_losses_with(self.static, self.prompt, self.t, self.a, self.noise).sum()
        loss.backward()
        return loss

    def capture(self):
        tr = self.trainer
        tr.netG.zero_grad(set_to_none=True)          # gradients are (re)allocated inside the capture: static from here on
        # (one stream: a graph with the two branches of forward_train as parallel paths replays slower than the single-stream
        #  one on this runtime -- 27.7 against 27.1 ms, and its launch costs the host 13 ms instead of 1)
        with ops.config.override(train_two_streams=False), torch.cuda.graph(self.graph):
            loss = self._step()
            if not tr.split_graph:
                tr.optG.step()
        self.loss = loss.detach()
        self.opt_graph = None
        if tr.split_graph:       # > 1 rank: the gradient exchange runs between two graphs (see DDPM.optimize_parameters)
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph):
                tr.optG.step()
        return self

    def replay(self):
        self.graph.replay()
        if self.opt_graph is not None:
            tdist.allreduce_gradients(self.trainer.netG, op=self.trainer.grad_reduce)
            self.opt_graph.replay()
        self.replays += 1


class DDPM(BaseModel):
    GRAPH_WARMUP = 2       # eager steps per (prompt, batch shape) before its step is captured (optimizer state, packed-weight
                           # sets and workspaces exist by then)

    def __init__(self, opt):
        super().__init__(opt)
        self.optG = None
        self.scheduler = None
        self.netG = self.set_device(networks.define_General(opt))
        self.schedule_phase = None
        train_opt = (opt.get("train") or {}) if isinstance(opt, dict) else {}
        self.grad_reduce = train_opt.get("grad_reduce", "sum")
        # train.hip_graph (default: ops.config.train_graph / TMDIFF_TRAIN_GRAPH): record the finetune step into a HIP graph
        self.use_graph = bool(train_opt.get("hip_graph", ops.config.train_graph)) and self.device.type == "cuda"
        self.split_graph = False
        self._captured, self._eager_steps, self._drop_word = {}, {}, None
        self.reducer = None
        self.set_loss()
        if self.opt["phase"] == "train":
            self.netG.train()
            optim_params = [p for n, p in self.netG.named_parameters() if "clip_text" not in n]
            lr = opt["train"]["optimizer"]["lr"]
            if self.device.type == "cuda":
                # torch.optim.AdamW as ONE multi-tensor HIP launch (tmdiff_amd.optim; same state layout and checkpoint
                # files); the learning rate is a device scalar the scheduler fills, so the step can live in a HIP graph
                from .optim import FusedAdamW
                self.optG = FusedAdamW(optim_params, lr=torch.tensor(float(lr), device=self.device), weight_decay=1e-4)
            else:
                self.optG = torch.optim.AdamW(optim_params, lr=lr, weight_decay=1e-4)
            self.scheduler = linear_warmup_decay(self.optG, 100, opt["train"]["max_iter"])
            self.log_dict = OrderedDict()
        self.load_network()
        self.print_network()

    def feed_data(self, data):
        self.data = self.set_device(data)

    def _lr(self):
        lr = self.optG.param_groups[0]["lr"]
        return lr            # (a device tensor in graph mode: formatting it is the only host read, done by whoever logs)

    def optimize_parameters(self, prompt=None):
        if self.reducer is None:       # created here: the process group exists by the first step
            self.reducer = tdist.GradReducer(self.netG, op=self.grad_reduce)
            self.split_graph = self.use_graph and self.reducer.active
        if self.use_graph:
            return self._optimize_captured(prompt)
        l_pix = self.netG(self.data, prompt).sum()
        l_pix.backward()                # bucket all-reduces start from hooks while backward is still running
        self.reducer.finish()           # no-op in a single process
        self.optG.step()
        self.scheduler.step()
        self.reducer.zero_grad()
        self.log_dict["l_pix"] = l_pix.detach()
        self.log_dict["lr"] = self._lr()

    def _optimize_captured(self, prompt):
        """The same step from a HIP graph (CapturedStep).  The first GRAPH_WARMUP steps of every (prompt, batch shape) run
        eagerly -- without the reducer's hooks: with more than one rank the exchange of a graph step is a flat all-reduce
        between the backward graph and the optimizer graph, and the warm-up steps do the same."""
        key = (prompt if not isinstance(prompt, (list, tuple)) else tuple(prompt),
               tuple((k, tuple(self.data[k].shape)) for k in ("Res", "PAN", "MS")))
        if self._drop_word is None:      # ONE word per trainer, alive as long as its graphs (they hold its address)
            self._drop_word = torch.randint(0, 2 ** 40, (1,), dtype=torch.int64).to(self.device)
        ops.DROP_WORD = self._drop_word
        step = self._captured.get(key)
        if step is None:
            done = self._eager_steps.get(key, 0)
            if done < self.GRAPH_WARMUP:
                self._eager_steps[key] = done + 1
                keep, self.reducer.active = self.reducer.active, False
                try:
                    self.netG.zero_grad(set_to_none=True)
                    l_pix = self.netG(self.data, prompt).sum()
                    l_pix.backward()
                finally:
                    self.reducer.active = keep
                if self.split_graph:
                    tdist.allreduce_gradients(self.netG, op=self.grad_reduce)
                self.optG.step()
                self.scheduler.step()
                self.log_dict["l_pix"], self.log_dict["lr"] = l_pix.detach(), self._lr()
                return
            step = CapturedStep(self, self.data, prompt)
            step.load(self.data)
            keep, self.reducer.active = self.reducer.active, False
            try:
                step.capture()            # (recording runs nothing: the replay below is this step)
            finally:
                self.reducer.active = keep
            self._captured[key] = step
        else:
            step.load(self.data)
        step.replay()
        self.scheduler.step()             # (fills the device learning-rate tensor for the next step)
        self.log_dict["l_pix"], self.log_dict["lr"] = step.loss, self._lr()

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
