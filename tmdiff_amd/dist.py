"""One-process-per-GPU data parallelism for the hot path.

The reference only has single-process ``nn.DataParallel`` (GeneralModel/networks.py:88-91): it scatters the
batch, sums the replica losses (``model.py:41`` ``.sum()``) and so ends up with the SUM of the per-replica mean-loss
gradients.  The MI355X-native equivalent: each rank holds a full replica and its share of the batch;

  * inference / sampling: ``shard_batch`` splits the image dict; no collective touches the data path;
  * finetune: after ``loss.backward()`` call ``allreduce_gradients(module)`` -- ONE exchange per step: gradients
    are flattened into a few large fp32 buckets (fully connected xGMI favours few large collectives) and
    all-reduced with SUM (default, the reference's DataParallel arithmetic) or averaged.

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


def allreduce_gradients(module, op="sum", bucket_bytes=64 << 20, group=None):
    """Sum (or average) ``p.grad`` over ranks with a few flat-bucket all-reduces.  Returns #buckets."""
    if op not in ("sum", "mean"):
        raise ValueError("op must be 'sum' (DataParallel semantics of the reference) or 'mean'")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    grads = [p.grad for p in module.parameters() if p.grad is not None]
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


def gather_images(local, group=None):
    """Concatenate per-rank result tensors (equal shapes) on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    outs = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, local.contiguous(), group=group)
    return torch.cat(outs, dim=0)
