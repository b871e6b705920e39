"""world_size-2 gloo tests (CPU) of the multi-process path: batch sharding, gradient all-reduce with the
reference's SUM semantics, bucket layout with gradient-free parameters, result gathering."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    from tmdiff_amd import dist as D
    assert D.init_from_env("gloo") == world
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    for p in net[2].parameters():            # a gradient-free tail, like the 56 unused WavBEST tensors
        p.requires_grad_(True)
    data = {"x": torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10, "tag": "keep"}
    shard = D.shard_batch(data)
    assert shard["tag"] == "keep" and shard["x"].shape[0] == 4
    loss = net[1](net[0](shard["x"])).abs().mean()        # net[2] unused -> grad None
    loss.backward()
    local = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    nb = D.allreduce_gradients(net, op="sum", bucket_bytes=64)      # tiny buckets: several all-reduces
    summed = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    for p, l in zip(net.parameters(), local):
        if l is not None:
            p.grad.copy_(l)
    D.allreduce_gradients(net, op="mean")
    mean = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    gathered = D.gather_images(shard["x"] * (rank + 1))
    npy = lambda ts: [None if t is None else t.numpy().copy() for t in ts]      # by value: no shared-memory handles
    q.put((rank, nb, npy(local), npy(summed), npy(mean), gathered.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ten = lambda ts: [None if t is None else torch.from_numpy(t) for t in ts]
    res = [(r, nb, ten(a), ten(b), ten(c), torch.from_numpy(g)) for r, nb, a, b, c, g in res]
    (_, nb0, loc0, sum0, mean0, gat0), (_, nb1, loc1, sum1, mean1, gat1) = res
    assert nb0 == nb1 and nb0 > 1
    for a, b, s0, s1, m0 in zip(loc0, loc1, sum0, sum1, mean0):
        if a is None:
            assert b is None and s0 is None and s1 is None          # untouched on every rank
            continue
        assert torch.allclose(s0, a + b) and torch.equal(s0, s1)      # SUM, identical on both ranks
        assert torch.allclose(m0, (a + b) / 2)
    assert torch.equal(gat0, gat1) and gat0.shape[0] == 8
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10
    assert torch.allclose(gat0, torch.cat([x[:4] * 1, x[4:] * 2]))
    # the all-reduced SUM equals the gradient of the sum of the two replica mean-losses (reference DataParallel)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    total = net[1](net[0](x[:4])).abs().mean() + net[1](net[0](x[4:])).abs().mean()
    total.backward()
    for p, s in zip(net.parameters(), sum0):
        if s is not None:
            assert torch.allclose(p.grad, s, atol=1e-6)


def test_single_process_is_a_noop():
    from tmdiff_amd import dist as D
    net = torch.nn.Linear(3, 2)
    net(torch.ones(1, 3)).sum().backward()
    g = net.weight.grad.clone()
    assert D.allreduce_gradients(net) == 0 and torch.equal(net.weight.grad, g)
    t = torch.ones(2, 3)
    assert D.gather_images(t) is t
    assert D.shard_batch({"x": torch.arange(6.0).reshape(6, 1)}, rank=1, world=3)["x"].flatten().tolist() == [2.0, 3.0]


# ---- overlapped bucketed all-reduce (GradReducer) ---------------------------------------------------------------
def _reducer_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    from tmdiff_amd import dist as D
    assert D.init_from_env("gloo") == world
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2), torch.nn.Linear(3, 2))
    red = D.GradReducer(net, op="sum", bucket_bytes=80)          # tiny buckets: several per step
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10
    shard = D.shard_batch({"x": x})["x"]
    log = []
    for step in range(4):
        h = net[1](net[0](shard))
        # net[2] never gets a gradient; net[3] only from step 2 on (a parameter joining the layout late)
        loss = h.abs().mean() + (net[3](h).square().mean() if step >= 2 else 0.0)
        loss.backward()
        n = red.finish()
        log.append((n, red.launched, [None if p.grad is None else p.grad.clone().numpy() for p in net.parameters()]))
        opt.step()
        red.zero_grad()
    w = [p.detach().clone().numpy() for p in net.parameters()]
    q.put((rank, log, w))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_world2_matches_single_process_sum():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, log0, w0), (_, log1, w1) = res
    # single-process reference: gradient of the SUM of the two replica losses, same SGD
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2), torch.nn.Linear(3, 2))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10
    for step in range(4):
        total = 0.0
        for part in (x[:4], x[4:]):
            h = net[1](net[0](part))
            total = total + h.abs().mean() + (net[3](h).square().mean() if step >= 2 else 0.0)
        opt.zero_grad()
        total.backward()
        n0, launched0, g0 = log0[step]
        n1, launched1, g1 = log1[step]
        assert n0 == n1 and n0 >= 1
        for p, a, b in zip(net.parameters(), g0, g1):
            if p.grad is None:
                assert a is None and b is None
                continue
            assert a is not None and torch.allclose(torch.from_numpy(a), p.grad, atol=1e-6), step
            assert (a == b).all()                                   # identical on both ranks
        if step == 1:
            assert launched0 == n0 and launched0 > 1                 # every bucket went out from a backward hook
        opt.step()
    for p, a, b in zip(net.parameters(), w0, w1):
        assert torch.allclose(p.detach(), torch.from_numpy(a), atol=1e-6) and (a == b).all()


def test_ranks_draw_different_randomness():
    """every rank used to seed NumPy / torch identically, so all replicas drew the same timesteps, noise and
    dropout masks.  seed_all(rank=r) keeps `random` shared (the per-iteration dataset choice) and offsets the rest."""
    import random
    import numpy as np
    from tmdiff_amd.train import per_rank_batch, seed_all
    draws = []
    for r in (0, 1):
        seed_all(3407, rank=r)
        draws.append((random.random(), np.random.randint(1, 1001, size=8).tolist(), torch.randn(4).tolist()))
    assert draws[0][0] == draws[1][0]
    assert draws[0][1] != draws[1][1] and draws[0][2] != draws[1][2]
    assert per_rank_batch(64, 8) == 8 and per_rank_batch(32, 1) == 32
    import pytest
    for bad in ((4, 8), (30, 8), (0, 2)):          # a global batch the ranks cannot share evenly is refused, not rounded
        with pytest.raises(ValueError):
            per_rank_batch(*bad)


def _replica_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    from tmdiff_amd import dist as D
    from tmdiff_amd import model as Model
    from tmdiff_amd.train import build_replica, seed_all
    assert D.init_from_env("gloo") == world
    opt = {"phase": "train", "gpu_ids": None, "distributed": world > 1, "path": {"resume": None},
           "model": {"unet": {"channel_multiplier": [4, 8, 16, 32]}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
           "train": {"optimizer": {"lr": 1e-4}, "max_iter": 100}}
    seed_all(rank=rank)                                    # what train.main does before the loaders are built
    m = build_replica(lambda: Model.create_model(opt), rank, world)
    sd = {k: v.numpy().copy() for k, v in m.netG.state_dict().items()}
    draw = torch.randn(4).tolist()                         # after the build the ranks' generators differ again
    # broadcast_module alone repairs replicas that were built under different seeds (seed + rank before create_model)
    torch.manual_seed(100 + rank)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7))
    before = lin[0].weight.detach().numpy().copy()
    n = D.broadcast_module(lin, src=0)
    q.put((rank, sd, draw, before, lin[0].weight.detach().numpy().copy(), n))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_start_identical_world2():
    """train.main seeded torch with seed + rank BEFORE create_model, so every rank trained its own
    random initialisation with a shared gradient.  build_replica: model under the shared seed + broadcast from rank 0,
    rank offset only afterwards."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0, draw0, before0, after0, n0), (_, sd1, draw1, before1, after1, n1) = res
    assert sd0.keys() == sd1.keys() and len(sd0) >= 272
    for k in sd0:
        assert (sd0[k] == sd1[k]).all(), k
    assert any(abs(v).max() > 0 for k, v in sd0.items() if k.endswith("conv20.weight"))
    assert draw0 != draw1
    assert (before0 != before1).any() and (after0 == after1).all() and (after0 == before0).all()
    assert n0 == n1 and n0 >= 6          # weight, bias, BN affine + running stats


def test_grad_reducer_survives_replaced_grads():
    """optimizer.zero_grad(set_to_none=True) / an assigned .grad used to leave the hooks reducing a stale
    bucket.  The hook now notices that p.grad is no longer its bucket view, copies the gradient in and re-points it."""
    from tmdiff_amd import dist as D
    port = _free_port()
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
        red = D.GradReducer(net, op="sum", bucket_bytes=64, min_world=1)
        x = torch.randn(4, 6)
        for step in range(3):
            if step == 2:
                net.zero_grad(set_to_none=True)            # drops the bucket views
            net(x).abs().mean().backward()
            red.finish()
            want = torch.autograd.grad(net(x).abs().mean(), list(net.parameters()))
            for p, w in zip(net.parameters(), want):
                assert torch.allclose(p.grad, w, atol=1e-7)
                bk = red._bucket_of[id(p)]
                assert p.grad.data_ptr() == bk["views"][id(p)].data_ptr()      # a view of its bucket again
            red.zero_grad()
    finally:
        dist.destroy_process_group()
