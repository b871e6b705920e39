"""One-process-per-GPU data parallelism for the hot path.

The reference only has single-process ``nn.DataParallel`` (GeneralModel/networks.py:88-91): it scatters the
batch, sums the replica losses (``model.py:41`` ``.sum()``) and so ends up with the SUM of the per-replica mean-loss
gradients.  The MI355X-native equivalent: each rank holds a full replica and its share of the batch;

  * inference / sampling: ``shard_batch`` splits the image dict; no collective touches the data path;
  * finetune: ONE exchange per step -- gradients live in a few large flat fp32 buckets (fully connected xGMI
    favours few large collectives) that are all-reduced with SUM (default, the reference's DataParallel arithmetic)
    or averaged.  ``GradReducer`` makes ``p.grad`` a view of its bucket and starts a bucket's all-reduce from a
    post-accumulate hook as soon as backward has produced its last gradient (buckets follow the order in which
    backward produces gradients), so the exchange overlaps the rest of backward and there is no flatten / copy-back
    pass.  ``allreduce_gradients(module)`` is the simple after-backward form (used for the first step, and by tests).

Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` on CPU tensors (used by the CPU tests).  Parameters whose
gradient is None on this rank (56 WavBEST tensors never get one) are left untouched on every rank, so the bucket
layout is identical everywhere without communication.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return world


def shard_batch(x_in, rank=None, world=None):
    """Contiguous slice of every batched tensor in the dict for this rank (images are independent)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    out = {}
    for k, v in x_in.items():
        if torch.is_tensor(v) and v.dim() > 0:
            n = v.shape[0]
            lo, hi = n * rank // world, n * (rank + 1) // world
            out[k] = v[lo:hi].contiguous()
        else:
            out[k] = v
    return out


def broadcast_module(module, src=0, group=None):
    """Make every rank's parameters and buffers those of rank ``src`` (a few flat broadcasts).  The data-parallel
    replicas must start identical: gradients are only ever SUM-reduced, never the weights (reference: DataParallel
    replicates device 0's module every step, networks.py:88-91).  Returns the number of tensors synchronised."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
    by_kind = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    for ts in by_kind.values():
        cur, cur_bytes = [], 0
        chunks = []
        for t in ts:
            nb = t.numel() * t.element_size()
            if cur and cur_bytes + nb > (64 << 20):
                chunks.append(cur)
                cur, cur_bytes = [], 0
            cur.append(t)
            cur_bytes += nb
        if cur:
            chunks.append(cur)
        for ch in chunks:
            flat = torch.cat([t.reshape(-1) for t in ch])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ch:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
    inval = getattr(getattr(module, "denoise_fn", module), "invalidate_prepared", None)
    if inval is not None:            # packed weights / projection banks were built from the old values
        inval()
    return len(tensors)


def allreduce_gradients(module, op="sum", bucket_bytes=64 << 20, group=None):
    """Sum (or average) ``p.grad`` over ranks with a few flat-bucket all-reduces.  Returns #buckets."""
    if op not in ("sum", "mean"):
        raise ValueError("op must be 'sum' (DataParallel semantics of the reference) or 'mean'")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    return _allreduce_tensors([p.grad for p in module.parameters() if p.grad is not None], op, bucket_bytes, group)


def _allreduce_tensors(grads, op, bucket_bytes, group):
    world = dist.get_world_size(group)
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nb = g.numel() * g.element_size()
        if cur and cur_bytes + nb > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if op == "mean":
            flat /= world
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(buckets)


class GradReducer:
    """Bucketed gradient all-reduce overlapped with backward ("bucketed and overlapped with backward").

    Step 1 runs backward normally, records the order in which parameters received their gradients and reduces with
    ``allreduce_gradients``.  ``finish()`` of that step then lays the gradients out in flat buckets in that order
    (parameters that got no gradient -- 56 WavBEST tensors never do -- stay outside, identically on every rank) and
    re-points ``p.grad`` at views of them.  From step 2 on, autograd accumulates into the views in place, the
    post-accumulate hook of the last parameter of a bucket launches ``all_reduce(async_op=True)`` on the flat bucket
    (RCCL runs it on its own stream after the producing kernels), and ``finish()`` waits for the handles before
    the optimizer step.  Use ``zero_grad()`` of this object (zeroes the buckets in place) instead of
    ``optimizer.zero_grad()`` (which would drop the views).
    """

    def __init__(self, module, op="sum", bucket_bytes=32 << 20, group=None, min_world=2):
        """min_world: the smallest group the buckets / hooks are set up for (2: a single process does no exchange;
        1 lets a one-rank group run the whole machinery, which is how the RCCL path is exercised on a one-GPU box)."""
        if op not in ("sum", "mean"):
            raise ValueError("op must be 'sum' (DataParallel semantics of the reference) or 'mean'")
        self.module, self.op, self.bucket_bytes, self.group = module, op, bucket_bytes, group
        self.active = dist.is_initialized() and dist.get_world_size(group) >= min_world
        self.world = dist.get_world_size(group) if self.active else 1
        self.buckets = None          # list of dicts {flat, params, pending}
        self._order, self._bucket_of, self._handles = [], {}, []
        self.launched = 0            # all-reduces started from hooks during the current backward
        self.launched_last = 0       # ... during the last finished step (kept across zero_grad, for reports)
        self._hooks = []
        if self.active:
            for p in module.parameters():
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        if not self.active:
            return
        if self.buckets is None:
            self._order.append(p)
            return
        bk = self._bucket_of.get(id(p))
        if bk is None:               # a parameter outside the recorded layout got a gradient: rebuild after this step
            self._order.append(p)
            return
        view = bk["views"][id(p)]
        if p.grad.data_ptr() != view.data_ptr():
            # something replaced p.grad (optimizer.zero_grad(set_to_none=True), module.zero_grad(), an assignment): the
            # bucket slot would be reduced stale and the real gradient not at all -- copy it in and re-point p.grad
            view.copy_(p.grad)
            p.grad = view
        if p.grad.is_cuda:           # (forward_train may run its two branches on two streams: autograd runs each backward
            s = torch.cuda.current_stream(p.grad.device)      # node -- and this hook -- on its forward's stream)
            bk["streams"][s.cuda_stream] = s
        bk["pending"] -= 1
        if bk["pending"] == 0:
            self._launch(bk)
            self.launched += 1

    def _launch(self, bk):
        """All-reduce of a bucket whose gradients are all enqueued: the collective is ordered after the CURRENT stream only, so
        that stream first waits for every other stream that produced one of the bucket's gradients."""
        if bk["streams"]:
            cur = torch.cuda.current_stream(bk["flat"].device)
            for sid, s in bk["streams"].items():
                if sid != cur.cuda_stream:
                    cur.wait_stream(s)
        self._handles.append(dist.all_reduce(bk["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _build(self):
        params = [p for p in self._order if p.grad is not None]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.buckets, self._bucket_of = [], {}
        cur, cur_bytes = [], 0
        groups = []
        for p in uniq:
            nb = p.numel() * 4
            if cur and cur_bytes + nb > self.bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            groups.append(cur)
        for ps in groups:
            flat = torch.zeros(sum(p.numel() for p in ps), device=ps[0].device, dtype=torch.float32)
            off = 0
            views = {}
            for p in ps:
                view = flat[off:off + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view
                views[id(p)] = view
                off += p.numel()
            bk = {"flat": flat, "params": ps, "pending": len(ps), "views": views, "streams": {}}
            self.buckets.append(bk)
            for p in ps:
                self._bucket_of[id(p)] = bk
        self._order = []

    def finish(self):
        """Call after backward, before the optimizer step.  Returns the number of all-reduces of this step."""
        if not self.active:
            return 0
        if self.buckets is None:     # first step: plain path, then lay the buckets out in the recorded order
            n = _allreduce_tensors([p.grad for p in self.module.parameters() if p.grad is not None], self.op,
                                   self.bucket_bytes, self.group)
            self._build()
            return n
        # buckets whose hooks did not all fire (a parameter got no gradient this step: its slot is still zero)
        for bk in self.buckets:
            if bk["pending"] > 0:
                self._launch(bk)
                bk["pending"] = 0
        for h in self._handles:
            h.wait()
        n, self._handles = len(self._handles), []
        if self.op == "mean":
            for bk in self.buckets:
                bk["flat"] /= self.world
        if self._order:              # gradients outside the layout (graph changed): reduce them now, extend the layout
            stray = [p for p in self._order if p.grad is not None]
            n += _allreduce_tensors([p.grad for p in stray], self.op, self.bucket_bytes, self.group)
            self._order = [p for bk in self.buckets for p in bk["params"]] + stray
            self._build()
        return n

    def zero_grad(self):
        self.launched_last, self.launched = self.launched, 0
        if self.buckets is None:
            self.module.zero_grad(set_to_none=True)
            self._order = []
            return
        for bk in self.buckets:
            bk["flat"].zero_()
            bk["pending"] = len(bk["params"])
            bk["streams"] = {}
        for p in self.module.parameters():           # gradient-free parameters stay None
            if p.grad is not None and id(p) not in self._bucket_of:
                p.grad = None

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def gather_images(local, group=None):
    """Concatenate per-rank result tensors (equal shapes) on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    outs = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, local.contiguous(), group=group)
    return torch.cat(outs, dim=0)
