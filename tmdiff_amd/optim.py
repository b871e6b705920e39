"""AdamW of the finetune step as ONE HIP launch (tmdiff_multi_adamw).

``FusedAdamW`` IS a ``torch.optim.AdamW`` (reference GeneralModel/model.py:30-31: ``AdamW(optim_params, lr, weight_decay=1e-4)``):
same constructor arguments, same ``param_groups`` / ``state`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so
``state_dict()`` / ``load_state_dict()`` and the ``I{iter}_opt.pth`` files interchange with the reference's optimizer and LR
schedulers drive it unchanged.  Only ``step()`` differs: every parameter that has a gradient is updated by one multi-tensor kernel
that reads the learning rate and the step count from device scalars -- no host read anywhere, so the step can be recorded into a
HIP graph (torch's capturable AdamW does the same update in ~25 multi-tensor / elementwise launches: 3 ms per step of WavBEST's
272 tensors against 0.15 ms).  amsgrad / maximize / fp16 parameters are not supported (the reference uses none of them)."""
import ctypes as C
import struct

import torch

from . import ops
from ._lib import check, lib


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        for bad in ("amsgrad", "maximize"):
            if kw.get(bad):
                raise ValueError(f"FusedAdamW: {bad} is not supported")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._tables = {}        # group index -> dict(key, entries, chunk_tensor, chunk_index, n_chunks, pinned host copies)
        self._lr_dev = {}        # group index -> (device scalar, last host value) when the group's lr is a Python float

    # -- state --------------------------------------------------------------------------------------------------------------
    def _group_step(self, group):
        """ONE device step counter per group, shared by the ``state[p]['step']`` entries of its parameters (torch keeps one
        scalar per parameter: 272 increments per step; shared, the counter is bumped once and every entry sees it)."""
        shared = None
        for p in group["params"]:
            st = self.state.get(p)
            if st and "step" in st and torch.is_tensor(st["step"]) and st["step"].is_cuda:
                shared = st["step"] if shared is None else shared
        return shared

    def _init_state(self, group, params):
        shared = self._group_step(group)
        if shared is None:
            # (after load_state_dict the entries are host / per-parameter scalars: continue from their value)
            t0 = max([float(self.state[p]["step"]) for p in params if self.state.get(p) and "step" in self.state[p]] or [0.0])
            shared = torch.full((), t0, dtype=torch.float32, device=params[0].device)
        for p in group["params"]:
            st = self.state[p] if p in self.state or p.grad is not None else None
            if st is None:
                continue
            st["step"] = shared
            if p.grad is not None and "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return shared

    def _table(self, gi, params):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                     p.numel()) for p in params)
        tb = self._tables.get(gi)
        if tb is not None and tb["key"] == key:
            return tb
        chunk = lib.tmdiff_multi_axpby_chunk()
        ent, ct, ci = bytearray(), [], []
        for k, (pp, gp, mp, vp, n) in enumerate(key):
            ent += struct.pack("<QQQQq", pp, gp, mp, vp, n)
            nck = (n + chunk - 1) // chunk
            ct += [k] * nck
            ci += list(range(nck))
        dev = params[0].device
        # pinned host copies + non-blocking uploads: legal inside a HIP-graph capture (the gradients of a captured step are
        # allocated inside the capture, so their addresses -- hence this table -- are first known there); the pinned tensors
        # stay alive with the table, a replay copies the same bytes again
        host = [torch.frombuffer(bytearray(ent), dtype=torch.uint8).clone().pin_memory(),
                torch.tensor(ct, dtype=torch.int32).pin_memory(), torch.tensor(ci, dtype=torch.int32).pin_memory()]
        devt = [h.to(dev, non_blocking=True) for h in host]
        tb = self._tables[gi] = {"key": key, "host": host, "entries": devt[0], "chunk_tensor": devt[1], "chunk_index": devt[2],
                                 "n_chunks": len(ct)}
        return tb

    def _lr_scalar(self, gi, group, dev):
        lr = group["lr"]
        if torch.is_tensor(lr):
            if not lr.is_cuda or lr.dtype != torch.float32:
                raise ValueError("FusedAdamW: a tensor learning rate must be a float32 scalar on the GPU")
            return lr
        cur = self._lr_dev.get(gi)
        if cur is None:
            cur = self._lr_dev[gi] = [torch.full((), float(lr), dtype=torch.float32, device=dev), float(lr)]
        elif cur[1] != float(lr):
            cur[0].fill_(float(lr))
            cur[1] = float(lr)
        return cur[0]

    def state_dict(self):
        """torch's layout, with a step scalar OF ITS OWN per parameter (here they share one device counter; torch.optim.AdamW,
        which a reference-side run may load this checkpoint into, increments every entry and must not find them aliased)."""
        sd = super().state_dict()
        sd["state"] = {k: dict(v) for k, v in sd["state"].items()}       # (the packed state holds the LIVE per-parameter dicts)
        steps = {}
        for st in sd["state"].values():
            if "step" in st and torch.is_tensor(st["step"]):
                key = st["step"].data_ptr()
                if key not in steps:
                    steps[key] = float(st["step"])          # (one host read per group, at checkpoint time only)
                st["step"] = torch.tensor(steps[key], dtype=torch.float32)
        return sd

    # -- the step -------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and
                        p.grad.dtype == torch.float32 and not p.grad.is_sparse):
                    raise RuntimeError("FusedAdamW: parameters and gradients must be dense contiguous float32 tensors on the GPU")
            step = self._init_state(group, params)
            step.add_(1.0)
            tb = self._table(gi, params)
            b1, b2 = group["betas"]
            check(lib.tmdiff_multi_adamw(tb["entries"].data_ptr(), tb["chunk_tensor"].data_ptr(), tb["chunk_index"].data_ptr(),
                                         tb["n_chunks"], self._lr_scalar(gi, group, params[0].device).data_ptr(), step.data_ptr(),
                                         float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), ops.stream_ptr()),
                  "multi_adamw")
        return loss
