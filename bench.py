#!/usr/bin/env python3
"""Headline benchmark: UNet denoise-steps/sec on batch-32 8-channel 64x64 tiles (BASELINE config 2).

A *step* is one ``GeneralDiffusion.p_sample`` call on the batch: the full WavBEST forward (both
branches, 172.39 conv-GFLOP per sample -- exactly the work the reference does per step) plus the
fused DDPM update.  Inputs are resident in HBM before the timed region.  fp32 throughout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sample|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank denoises its own batch of 32 tiles (inference is batch-parallel: no collective on
the data path, "scaling": "weak"); value = N*K / max-over-ranks time.  Started WITHOUT a launcher (no WORLD_SIZE in
the environment) and with --gpus N > 1, this script is its own launcher: the parent starts N worker processes (one
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, backend nccl = RCCL) before it has touched the GPU, waits
for them and exits with their status -- it never initialises HIP and never re-execs itself.

--mode train measures the finetune step of BASELINE configs[3] instead (local batch 8 per GPU, ch 32-256, dropout on,
forward + backward + SUM all-reduce of the gradients over RCCL overlapped with backward + AdamW + EMA); the default
mode also reports a short run of it as `train_step`, beside `value`.

Extra objects on the JSON line (tier contract):
  roofline      conv3d 3x3x3 MFMA kernel: algorithmic FLOPs / summed HIP-event kernel durations measured
                inside the timed region, against the 157.3 TFLOP/s fp32 matrix peak.
  cpu_baseline  the CPU oracle (parity-locked restatement of the reference) timed on this box's host
                cores on a bounded sample, rank 0, N == 1 only.
  cond_cached   the same loop with the step-invariant condition branch hoisted out (evaluated once per
                sampling run): the rate a real 1000-step run sees.  Reported beside, never as, `value`.
  bf16_compute  the same full-forward loop with bf16 conv operands / fp32 accumulation (the config-3 mode);
                reduced precision, so also only beside `value`.
  parity        PSNR(build, oracle) of a short DDPM chain with shared noise (rank 0, N == 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL = [32, 64, 128, 256]
BATCH, BANDS, SIZE, T = 32, 8, 64, 1000
GFLOP_PER_SAMPLE = 172.39          # algorithmic conv FLOPs per sample per forward
PEAK_FP32_MFMA = 157.3             # TFLOP/s, MI355X_MICROARCH.md
COPY_RATE_TB_S = 6.29              # plain device-to-device copy rate of the box (tools/bench_hbm_kernels.py): the HBM roof in practice


def visible_gpus():
    """Number of GPUs this process tree may use, WITHOUT loading any GPU runtime (the launcher parent must stay GPU-free
    by construction: a process that has initialised HIP must not be the one that forks the ranks): the *_VISIBLE_DEVICES
    lists if set, else the KFD topology's nodes with SIMDs (CPU nodes have simd_count 0).  None = cannot tell."""
    counts = []
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            counts.append(len([t for t in v.split(",") if t.strip() != ""]))
    if counts:
        return min(counts)
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as fh:
                    props = dict(l.split()[:2] for l in fh if len(l.split()) >= 2)
            except OSError:
                continue
            n += int(props.get("simd_count", "0")) > 0
        return n
    except OSError:
        return None


def launch_ranks(args, timeout_s=3000.0):
    """Parent of a launcher-less multi-GPU run: start one worker per GPU, poll them, and on the first failure (or the
    overall timeout) stop the others -- a rank that died leaves its peers blocked in a collective, so waiting for them in
    order would never return.  Nothing here imports torch or touches the GPU.  Returns the exit status for the parent:
    0 only if every rank exited 0."""
    n = args.gpus
    if not args.rehearse and os.environ.get("TMDIFF_BENCH_BACKEND", "nccl") == "nccl":   # (gloo: ranks may share a GPU)
        ndev = visible_gpus()
        if ndev is not None and ndev < n:
            raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    deadline = time.monotonic() + float(os.environ.get("TMDIFF_BENCH_LAUNCH_TIMEOUT", timeout_s))
    status, reason = 0, None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            status, reason = max(abs(rc) for _, rc in bad) or 1, f"rank {bad[0][0]} exited with status {bad[0][1]}"
            break
        if all(rc == 0 for rc in rcs):
            return 0
        if time.monotonic() > deadline:
            status, reason = 124, "ranks still running at the launcher's timeout"
            break
        time.sleep(0.2)
    print(f"[bench] launcher: {reason}; stopping the remaining ranks", file=sys.stderr, flush=True)
    for p in procs:                       # exactly the PIDs this function started
        if p.poll() is None:
            p.terminate()
    t_end = time.monotonic() + 10.0
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return status


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """CPU share of this process: affinity mask, capped by the cgroup quota (the GPU box gives a
    one-GPU job 16 of the host's cores; os.cpu_count() would report the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline():
    """The oracle on the host cores: p_sample steps on a bounded sample (~10-30 s), scaled to batch 32."""
    import torch
    from oracle import unet_ref as U
    from oracle.diffusion_ref import GeneralDiffusionRef
    from tmdiff_amd.util import synthetic_tile_batch
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads")
    net = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    diff = GeneralDiffusionRef(net, "l1")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": T}, "cpu")
    b, budget, max_steps = 4, 12.0, 200
    d = synthetic_tile_batch(3407, b, BANDS, SIZE)
    x = d["x_t"]
    with torch.no_grad():
        x = diff.p_sample(x, T - 1, condition_x=d, prompt="WV3")         # warm-up (page-in, thread pool)
        log("cpu_baseline: warm-up step done")
        t0 = time.perf_counter()
        steps = 0
        while steps < max_steps and time.perf_counter() - t0 < budget:   # bounded sample: ~12 s of CPU work
            x = diff.p_sample(x, T - 2 - steps, condition_x=d, prompt="WV3")
            steps += 1
            if steps % 8 == 0:
                log(f"cpu_baseline: step {steps} at {time.perf_counter() - t0:.1f} s")
        dt = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), model)
    except OSError:
        pass
    return {"value": round(b * steps / dt / BATCH, 6), "unit": "batch32-steps/s", "cores": cores, "cpu_model": model,
            "kind": "port",
            "sample": f"{steps} p_sample steps on {b} tiles of 8x64x64 ({dt:.1f} s), scaled to batch {BATCH}; "
                      f"torch {torch.__version__} CPU, {cores} threads",
            "sample_steps_per_s": round(b * steps / dt, 4)}


def parity_check(dev):
    """PSNR(build, oracle) of a short DDPM chain with shared noise (the "PSNR vs ref" half of the metric) ON THE KERNEL FAMILY
    THAT PRODUCED `value`: full-width network, two 8x64x64 tiles, T = 10, with the grid-size threshold that keeps small
    launches on the direct kernels switched off (ops.config.wino_min_blocks = 1), so that the chain runs on the Winograd kernels
    (in-kernel transform where the plane has >= 16 columns, transform pass + kernel below) and the composed Conv_0 + LL
    kernel, as the batch-32 workload does.  The launch counts per C entry point are reported.  The oracle is the checker
    here, as in tests/ and smoke()."""
    import collections
    import torch
    from oracle import unet_ref as U
    from oracle.diffusion_ref import GeneralDiffusionRef
    from tmdiff_amd import ops
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.util import psnr, synthetic_tile_batch
    ref_net = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    net = WavBEST(channels=FULL)
    net.load_state_dict(ref_net.state_dict())
    net = net.to(dev).eval()
    d = synthetic_tile_batch(77, 2, BANDS, SIZE)
    noise = lambda like: torch.randn(like.shape, dtype=torch.float32)       # CPU generator on both sides
    want = GeneralDiffusionRef(ref_net, "l1")
    want.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, "cpu")
    got = GeneralDiffusion(net, "l1", noise_fn=noise).to(dev)
    got.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, dev)
    torch.set_num_threads(usable_cores())
    torch.manual_seed(5)
    with torch.no_grad():
        y_ref = want.p_sample_loop(d, continous=False, prompt="WV3")
    keep, ops.config.wino_min_blocks, ops.COUNTS = ops.config.wino_min_blocks, 1, collections.Counter()
    try:
        torch.manual_seed(5)
        y = got.p_sample_loop({k: v.to(dev) for k, v in d.items()}, continous=False, prompt="WV3").cpu()
    finally:
        counts, ops.COUNTS, ops.config.wino_min_blocks = dict(ops.COUNTS), None, keep
    k3 = {k: v for k, v in counts.items() if not k.endswith("_k1")}
    return {"psnr_db": round(float(psnr(y, y_ref)), 1), "max_abs_diff": float((y - y_ref).abs().max()),
            "launches_by_entry": k3,
            "what": "10-step DDPM chain (T=10 cosine) on two 8x64x64 tiles, ch 32-256, shared CPU noise: fused image of the "
                    "HIP path vs the CPU oracle, with the production kernel family forced onto this small batch "
                    "(conv3d_wf_fwd = Winograd F(4,3) with in-kernel input transform, conv3d_wino*_fwd = transform pass + "
                    "Winograd kernel, conv3d_ll_fwd = composed Conv_0 + LL, conv3d_wfll_fwd = the same with Winograd on top; conv3d_fwd* = direct kernels, the 8x8 level "
                    "here); budget PSNR >= 60 dB (tests/test_gpu_sampling.py hold full-width 50- and 1000-step chains of "
                    "the reference to the same on these kernels)"}


def collective_evidence(dist, world, rank, backend, device=None, device_index=None):
    """What the process group itself says about the run (`n_gpus` used to be WORLD_SIZE from the environment,
    not something a collective had confirmed): a SUM all-reduce of a ones tensor ON THE DEVICE (ranks_seen must equal world),
    and an all-gather of every rank's device index and host PID.  Called right after init_process_group and again after the
    timed region; rank 0 puts both into the line.  The reference's multi-GPU path is nn.DataParallel over `gpu_ids`
    (GeneralModel/networks.py:88-92): one replica per listed device, which `device_ids` here evidences."""
    import torch
    if dist is None:
        return {"world": 1, "ranks_seen": 1, "backend": None, "device_ids": [device_index], "pids": [os.getpid()]}
    one = torch.ones(1, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    gdev = device if backend == "nccl" else None          # (gloo gathers host tensors only; its all-reduce takes device ones)
    mine = torch.tensor([rank, -1 if device_index is None else device_index, os.getpid()], device=gdev, dtype=torch.int64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    got = sorted((int(g[0]), int(g[1]), int(g[2])) for g in got)
    return {"world": dist.get_world_size(), "ranks_seen": int(round(float(one[0]))), "backend": dist.get_backend(),
            "ranks": [g[0] for g in got], "device_ids": [g[1] for g in got], "pids": [g[2] for g in got]}


def per_rank_ms(dist, world, dt, steps, device=None):
    """ms per step of every rank (all-gather), beside the MAX the contract asks for."""
    import torch
    if dist is None:
        return [round(dt / steps * 1e3, 3)]
    mine = torch.tensor([dt / steps * 1e3], device=device if dist.get_backend() == "nccl" else None, dtype=torch.float64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return [round(float(g[0]), 3) for g in got]


def rehearse(world, rank, args):
    """Launcher rehearsal without a GPU (--rehearse; used by tests/test_bench_launcher.py): the same rendezvous,
    barrier-bracketed timed region and MAX-over-ranks reduction as the real run, over gloo, with a token CPU step."""
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    x = torch.ones(64, 64)
    if os.environ.get("TMDIFF_BENCH_REHEARSE_FAIL_RANK") == str(rank):     # (tests: a rank that dies before the collective)
        os._exit(7)
    barrier = (lambda: dist.barrier()) if dist is not None else (lambda: None)
    ev0 = collective_evidence(dist, world, rank, "gloo")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = (x @ x) / 64.0
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    ranks = torch.tensor([float(rank + 1)])
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)          # every rank took part: 1 + 2 + ... + world
    ev1 = collective_evidence(dist, world, rank, "gloo")
    ms = per_rank_ms(dist, world, dt, max(args.steps, 1))
    if rank == 0:
        print(json.dumps({"metric": "launcher rehearsal (no GPU work)", "rehearsal": True, "n_gpus": ev1["world"],
                          "steps": args.steps, "warmup": args.warmup, "rank_sum": float(ranks[0]),
                          "rccl": dict(ev0, after_timed_region=ev1),
                          "ms_per_step_by_rank": {"min": min(ms), "max": max(ms), "all": ms},
                          "value": round(world * args.steps / float(tt[0]), 3), "unit": "token-steps/s"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


PEAK_BF16_MFMA = 2500.0            # TFLOP/s dense, MI355X_MICROARCH.md


def bf16_by_level(timer, steps):
    """Per layer class of the bf16 compute mode (3x3x3 launches by plane size; 1x1x1 apart): time per step, FLOP rate against the
    2.5 PFLOP/s bf16 matrix peak, algorithmic bytes (packed bf16 input 2 B, fp32 outputs / residual 4 B, packed second output
    2 B per element) against the copy rate -- and which of the two fractions is the larger, i.e. the roof that binds."""
    import re
    import torch
    torch.cuda.synchronize()
    acc = {}
    for rec in timer.records:
        e0, e1, fl, k, what, tag = rec[:6]
        nbytes = rec[6] if len(rec) > 6 else 0.0
        m = re.search(r"(\d+)x(\d+)x(\d+) b", tag)
        key = ("k1 (all levels)" if k == 1 else f"k3 {m.group(2)}x{m.group(3)}") + ("" if "bf16" in what else " [fp32 kernel]")
        a = acc.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl; a[3] += nbytes
    out = {}
    for key, (n, ms, fl, nb) in sorted(acc.items()):
        tf, tb = fl / (ms * 1e-3) / 1e12, nb / (ms * 1e-3) / 1e12
        fm, fh = tf / PEAK_BF16_MFMA, tb / COPY_RATE_TB_S
        out[key] = {"launches_per_step": n // steps, "ms_per_step": round(ms / steps, 3), "tflops": round(tf, 1),
                    "frac_of_bf16_mfma_peak": round(fm, 3), "tb_per_s": round(tb, 2), "frac_of_copy_rate": round(fh, 3),
                    "nearer_roof": "hbm" if fh > fm else "mfma"}
    return out


def source_sha16():
    """Digest of the sources that decide which kernels the benchmark launches and what they do (csrc, ops.py, the UNet): the
    PMC traffic file records it at collection time, so a figure collected from other code is reported as stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "tmdiff_amd", "csrc", "*"))) + [
        os.path.join(ROOT, "tmdiff_amd", "ops.py"), os.path.join(ROOT, "tmdiff_amd", "Hyper_unet_general.py")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


TRAFFIC_FILE = "profiles/r04_bench_traffic.json"


def load_traffic():
    """HBM bytes per 3x3x3 conv launch from the committed PMC passes over this same command (tools/bench_traffic.py:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of bench.py, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside the process, so the line carries
    the figure of the last committed collection, says where it came from, and says whether the kernel sources have
    changed since (`stale`)."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    try:
        with open(path) as fh:
            t = json.load(fh)
        src = {"file": TRAFFIC_FILE, **{k: t[k] for k in t if k != "bytes_per_k3_launch"}}
        now = source_sha16()
        src["stale"] = t.get("source_sha16") != now
        if src["stale"]:
            src["stale_note"] = f"collected from sources {t.get('source_sha16')}, this run is {now}: re-run tools/bench_traffic.sh"
        return t["bytes_per_k3_launch"], src
    except (OSError, KeyError, ValueError):
        return None, TRAFFIC_FILE + " missing: run tools/bench_traffic.sh on the GPU box"


def train_leg(dev, world, rank, dist, steps, warmup, graph=False):
    """BASELINE configs[3] per-GPU share: local batch 8 of 8x64x64 tiles, ch 32-256, dropout on, forward + backward +
    SUM all-reduce (overlapped with backward, tmdiff_amd.dist.GradReducer) + AdamW + EMA (reference model.py:40-47,
    general_finetune.json:64-66, utils/EmaUpdater.py).  Returns seconds for `steps` steps plus all-reduce figures.
    graph: the step recorded into a HIP graph (tmdiff_amd.model.CapturedStep; one rank: forward + backward + AdamW are one
    graph launch, the EMA update one more kernel) -- the first two steps run eagerly, so warmup >= 3 puts only replays
    into the timed region."""
    import copy
    import numpy as np
    from tmdiff_amd.model import DDPM, EmaUpdater
    from tmdiff_amd.util import fill_weights_, synthetic_tile_batch
    opt = {"phase": "train", "gpu_ids": [dev.index], "distributed": world > 1, "path": {"resume": None},
           "model": {"unet": {"channel_multiplier": FULL}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
           "train": {"optimizer": {"lr": 1e-4}, "max_iter": 150000, "hip_graph": bool(graph)}}
    m = DDPM(opt)
    fill_weights_(m.netG.denoise_fn)                     # same weights on every rank (replicas start identical) ...
    if dist is not None:                                 # ... and, as tmdiff_amd.train.build_replica does, made so by a
        from tmdiff_amd import dist as tdist             # broadcast from rank 0 (the first RCCL traffic of the leg)
        tdist.broadcast_module(m.netG, src=0)
    m.netG.denoise_fn.invalidate_prepared()
    m.set_new_noise_schedule({"schedule": "cosine", "n_timestep": T}, "train")
    ema = EmaUpdater(m, copy.deepcopy(m))
    np.random.seed(3407 + rank)
    d = synthetic_tile_batch(4000 + rank, 8, BANDS, SIZE, device=dev)
    d["LR"] = d["MS"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n, it0):
        for i in range(n):
            m.feed_data(d)
            m.optimize_parameters("WV3")
            ema.update(it0 + i)

    run(max(warmup, 4 if graph else 2), 0)   # step 1 lays out the gradient buckets; overlapped all-reduce from step 2 on
                                             # (graph: two eager steps, the capture, one replay before the clock starts)
    barrier()
    t0 = time.perf_counter()
    run(steps, 10)
    t_host = time.perf_counter() - t0          # launches enqueued (the host runs ahead of the GPU when it can)
    barrier()
    dt = time.perf_counter() - t0
    from tmdiff_amd import ops as _ops
    _ops.FLOPS = [0.0]                          # one more step with the executed-FLOP counter on (outside the timed region;
    keep_graph, m.use_graph = m.use_graph, False    # eagerly: a graph replay passes no Python counter)
    run(1, 50)
    m.use_graph = keep_graph
    barrier()
    executed, _ops.FLOPS = _ops.FLOPS[0], None
    out = {"seconds": dt, "host_seconds": t_host, "loss": float(m.get_current_log()["l_pix"]), "allreduce": None,
           "executed_gflop_per_step": executed / 1e9, "captured": bool(graph),
           "graph_replays": sum(c.replays for c in m._captured.values()) if graph else 0}
    if dist is not None:
        red = m.reducer
        hook_launches = red.launched_last      # of the last timed step (the exchange-free steps below reset it)
        nbytes = sum(bk["flat"].numel() * 4 for bk in red.buckets)
        # the same exchange un-overlapped (all buckets back to back, nothing else running), for the exposed share
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            hs = [dist.all_reduce(bk["flat"], async_op=True) for bk in red.buckets]
            for h in hs:
                h.wait()
        barrier()
        alone = (time.perf_counter() - t1) / 5
        # and the step without any exchange
        red.active = False
        run(2, 100)
        barrier()
        t2 = time.perf_counter()
        run(steps, 200)
        barrier()
        dt_noex = time.perf_counter() - t2
        red.active = True
        tt = torch.tensor([dt, alone, dt_noex], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, alone, dt_noex = (float(v) for v in tt)
        out["seconds"] = dt
        out["allreduce"] = {"buckets": len(red.buckets), "launched_from_backward_hooks": hook_launches,
                            "payload_mb": round(nbytes / 1e6, 1), "standalone_ms": round(alone * 1e3, 3),
                            "bus_gb_s": round(2 * (world - 1) / world * nbytes / alone / 1e9, 1),
                            "step_ms_without_exchange": round(dt_noex / steps * 1e3, 3),
                            "exposed_ms_per_step": round((dt - dt_noex) / steps * 1e3, 3)}
    del m, ema
    torch.cuda.empty_cache()
    return out


TRAIN_GFLOP_PER_SAMPLE = 3 * GFLOP_PER_SAMPLE      # training step ~ forward + dgrad + wgrad = 517 GF/sample


def train_object(leg, world, steps, eager=None):
    dt = leg["seconds"]
    extra = {}
    if leg.get("captured"):
        extra = {"captured_in_hip_graph": True, "graph_replays": leg["graph_replays"],
                 "capture": "forward + backward + AdamW of the step as ONE HIP-graph launch (tmdiff_amd.model.CapturedStep); per "
                            "step the host copies the batch, the NumPy-drawn timesteps and their table values into fixed device "
                            "tensors, fills the noise tensor, launches the graph and the EMA kernel"}
        if eager is not None:
            extra["eager_ms_per_step"] = round(eager["seconds"] / steps * 1e3, 3)
            extra["eager_host_enqueue_ms_per_step"] = round(eager["host_seconds"] / steps * 1e3, 3)
    return {"value": round(world * 8 * steps / dt, 2), "unit": "samples/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(leg["host_seconds"] / steps * 1e3, 3), **extra,
            "steps": steps, "global_batch": 8 * world,
            # what the matrix pipe EXECUTES (Winograd forward / data-gradient convolutions and the composed Conv_0 + LL
            # convolution run fewer multiply-adds than the reference's operator order; the weight gradient runs them all)
            "executed_tflops_per_gpu": round(leg["executed_gflop_per_step"] * 1e-3 * steps / dt, 2),
            "frac_of_fp32_mfma_peak_executed": round(leg["executed_gflop_per_step"] * 1e-3 * steps / dt / PEAK_FP32_MFMA, 4),
            # ... and the same step priced at 3 x the forward's FLOPs in the reference's operator order: a rate
            # for comparisons with the reference, NOT a fraction of the peak
            "reference_order_tflops_per_gpu": round(8 * TRAIN_GFLOP_PER_SAMPLE * 1e-3 * steps / dt, 2),
            "loss": round(leg["loss"], 5), "allreduce": leg["allreduce"],
            "what": "BASELINE configs[3] per-GPU share: local batch 8 of 8x64x64 tiles, ch 32-256, dropout 0.2 on, fwd + bwd + "
                    "SUM all-reduce of 216 gradient tensors (RCCL, flat buckets, started from backward hooks) + AdamW + EMA; "
                    "one rank: the step is replayed from a HIP graph (eager figures beside it), several ranks: eager launches "
                    "with the exchange overlapped"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["sample", "train"], default="sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region and the instrumented pass "
                    "(no cond-cached / bf16 / train / parity / cpu legs): what tools/bench_traffic.sh profiles")
    ap.add_argument("--rehearse", action="store_true", help="CPU-only launcher rehearsal over gloo (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))          # parent: starts the ranks, never touches the GPU

    global torch
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world}: reporting n_gpus={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.rehearse:
        return rehearse(world, rank, args)
    dist = None
    ndev = torch.cuda.device_count()
    backend = os.environ.get("TMDIFF_BENCH_BACKEND", "nccl")      # nccl == RCCL over xGMI
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, {ndev} visible (one device per rank)")
    local_dev = local % max(ndev, 1)      # (gloo rehearsals on a 1-GPU box map every rank to device 0)
    if world > 1:
        import torch.distributed as dist
        kw = {"device_id": torch.device("cuda", local_dev)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # the first collective of the run, on the device: how many ranks the process group really spans, and on which devices
    evidence = collective_evidence(dist, world, rank, backend, dev, torch.cuda.current_device())
    if evidence["ranks_seen"] != world:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but the all-reduce saw {evidence['ranks_seen']} rank(s)")

    from tmdiff_amd import ops
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.util import fill_weights_, synthetic_tile_batch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        log(f"rank {rank}/{world}: finetune step on {torch.cuda.get_device_name(dev)}")
        graph = world == 1 and os.environ.get("TMDIFF_BENCH_TRAIN_GRAPH", "1") != "0"
        eager = train_leg(dev, world, rank, dist, args.steps, args.warmup) if (
            graph and os.environ.get("TMDIFF_BENCH_TRAIN_EAGER", "1") != "0") else None      # (profiling runs: only the graph leg)
        leg = train_leg(dev, world, rank, dist, args.steps, args.warmup, graph=graph)
        if rank == 0:
            obj = train_object(leg, world, args.steps, eager)
            line = {"metric": "finetune train samples/sec (8-ch 64x64 tiles, local batch 8 per GPU)", "value": obj["value"],
                    "unit": "samples/s", "n_gpus": evidence["world"], "rccl": evidence, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": obj["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "BASELINE configs[3]: " + obj["what"], "global_batch": 8 * world,
                               "parallelism": f"data-parallel x{world}, SUM all-reduce"},
                    "train_step": obj}
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    net = fill_weights_(WavBEST(channels=FULL)).to(dev).eval()
    diff = GeneralDiffusion(net, "l1").to(dev)
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": T}, dev)
    d = synthetic_tile_batch(3407 + rank, BATCH, BANDS, SIZE, device=dev)      # resident in HBM from here on
    torch.manual_seed(1234 + rank)
    torch.cuda.manual_seed(1234 + rank)

    def run(steps, first_t, x):
        for i in range(steps):
            x = diff.p_sample(x, first_t - i, condition_x=d, prompt="WV3")
        return x

    log(f"rank {rank}/{world}: model and inputs on {torch.cuda.get_device_name(dev)}")
    x = torch.randn_like(d["Res"])
    x = run(args.warmup, T - 1, x)
    torch.cuda.synchronize()
    log("warm-up done")
    # ---- timed region: K full steps, nothing but the hot path inside --------------------------------------------
    barrier()
    t0 = time.perf_counter()
    x = run(args.steps, T - 1 - args.warmup, x)
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    rank_ms = per_rank_ms(dist, world, dt, args.steps, dev)
    evidence = dict(evidence, after_timed_region=collective_evidence(dist, world, rank, backend, dev, torch.cuda.current_device()))

    # ---- the same K steps again with a HIP event pair around every conv launch (roofline object).  Kept out of the
    # run that produces `value`; its wall time is reported so the two can be compared. -------------------------------
    ops.TIMER = ops.ConvTimer()
    barrier()
    ti = time.perf_counter()
    xi = run(args.steps, T - 1 - args.warmup, x)
    barrier()
    dt_instr = time.perf_counter() - ti
    timer, ops.TIMER = ops.TIMER, None
    conv = timer.summary()
    conv_by_entry = timer.summary(by_entry=True)
    k1_n, k1_ms, k1_bytes = timer.bytes_summary(1)
    del xi

    dt_cached = dt_bf16 = bf16_levels = None
    train = None
    if not args.no_extras:
        # ---- the same loop with the condition branch hoisted (a real sampling run), outside `value` -------
        barrier()
        t1 = time.perf_counter()
        net.begin_condition_cache(d["PAN"], d["MS"], "WV3")
        x2 = run(args.steps, T - 1, torch.randn_like(d["Res"]))
        net.end_condition_cache()
        barrier()
        dt_cached = time.perf_counter() - t1
        log(f"cond-cached loop: {args.steps} steps in {dt_cached:.3f} s")

        # ---- the full-forward loop again with bf16-operand convolutions (SURVEY 8d config-3 mode), outside `value` ----
        net.set_compute_dtype("bf16")
        x3 = run(2, T - 1, torch.randn_like(d["Res"]))          # packs the bf16 weights
        barrier()
        t2 = time.perf_counter()
        x3 = run(args.steps, T - 1, x3)
        barrier()
        dt_bf16 = time.perf_counter() - t2
        # ... and three more steps with an event pair around every convolution launch: which roof binds each layer class
        ops.TIMER = ops.ConvTimer()
        run(3, T - 1, x3)
        bf16_timer, ops.TIMER = ops.TIMER, None
        bf16_levels = bf16_by_level(bf16_timer, 3)
        net.set_compute_dtype("fp32")
        assert torch.isfinite(x3).all()
        log(f"bf16-compute loop: {args.steps} steps in {dt_bf16:.3f} s")
        del x2, x3

    if dist is not None:
        tt = torch.tensor([dt, dt_cached or 0.0, dt_bf16 or 0.0, dt_instr], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_cached, dt_bf16, dt_instr = (float(v) for v in tt)

    line = None
    if rank == 0:
        n3, ms3, fl3 = conv.get(3, (0, 0.0, 0.0))
        n1, ms1, fl1 = conv.get(1, (0, 0.0, 0.0))
        n0, ms0, by0 = conv.get(0, (0, 0.0, 0.0))        # Winograd input-transform passes (HBM-bound; "flops" slot = bytes)
        achieved = fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else 0.0
        # FLOPs in the reference's operator order: a conv3d_ll_fwd launch (Conv_0 + halved LL band of a main-branch down
        # block as one strided convolution, csrc/conv3d_ll.hip) executes 48 of the 4 x 27 multiply-adds per output
        # ... and a conv3d_wino{4,2}_fwd launch (Winograd F(4,3) / F(2,3) along the band axis, csrc/conv3d_wino.hip; its
        # input-transform pass is timed apart) 54 of the 4 x 27 per four output bands / 36 of the 2 x 27 per pair
        ref_factor = {"conv3d_ll_fwd": 108.0 / 48.0, "conv3d_wfll_fwd": 108.0 / 24.0, "conv3d_wino2_fwd": 1.5, "conv3d_wino4_fwd": 2.0,
                      "conv3d_wf_fwd": 2.0}   # composed LL (+ F(4,3) on top) / F(2,3) / F(4,3)
        fl3_ref = fl3 + sum(fl * (ref_factor[what] - 1.0) for (k, what), (n, ms, fl) in conv_by_entry.items()
                            if k == 3 and what in ref_factor)
        traffic, traffic_src = load_traffic()
        line = {
            "metric": "UNet denoise-steps/sec (8-ch 64x64, batch 32)",
            "value": round(world * args.steps / dt, 4),
            "unit": "batch32-steps/s",
            "n_gpus": evidence["world"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            # what the collectives themselves saw (a device SUM all-reduce of ones + an all-gather of device indices, once
            # after init_process_group and once after the timed region), and every rank's own step time beside the MAX
            "rccl": evidence,
            "ms_per_step_by_rank": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch-32 8-ch 64x64 tiles, DDPM p_sample steps of the "
                                   "T=1000 cosine schedule, channel_multiplier [32,64,128,256], full UNet forward "
                                   "(both branches) + fused DDPM update per step",
                       "batch_per_gpu": BATCH, "tile": [BANDS, SIZE, SIZE], "parallelism": f"batch-parallel x{world}",
                       "weights": "key-hashed random init", "text_embedding": "fixed synthetic 768-d"},
            "sample_steps_per_s": round(world * BATCH * args.steps / dt, 2),
            # 172.39 GFLOP per sample per forward in the REFERENCE's operator order / step time: a rate for comparisons with the
            # reference (the Winograd / composed kernels execute about half of these multiply-adds, so it may exceed the
            # 157.3 TFLOP/s matrix peak); the roofline object counts executed FLOPs
            "unet_reference_order_tflops": round(BATCH * GFLOP_PER_SAMPLE * 1e-3 * args.steps / dt, 2),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA, 4), "traffic": traffic, "traffic_source": traffic_src,
                         # every pass in front of a convolution kernel (Winograd input transform of the 8x8 level, prologue
                         # pass of the three-segment inputs) counted INTO the time: the figure DESIGN.md quotes
                         "frac_with_transforms": round(fl3 / ((ms3 + ms0) * 1e-3) / 1e12 / PEAK_FP32_MFMA, 4) if ms3 > 0 else 0.0,
                         "kernel": "3x3x3 conv launches, fp32 v_mfma_f32_32x32x2_f32.  Dominant: conv3d_wf_kernel (Winograd F(4,3) "
                                   "along the band axis, input transform inside the kernel: 2x fewer multiply-adds, no transform "
                                   "pass; its prologue pass for three-segment inputs and the reduction kernel of its split-K "
                                   "launches are inside its time).  At the benchmark batch it runs EVERY 3x3x3 convolution: "
                                   "conv3d_wf_fwd = the stride-1 ones (8x8 level: two images per tile), conv3d_wfll_fwd = Conv_0 + "
                                   "LL band of the main branch's down blocks as one convolution with Winograd on top (24 instead of "
                                   "48 / 108 multiply-adds per output, on the producer's space-to-depth output).  Fallbacks for other "
                                   "shapes: conv3d_ll_kernel, wino_input_kernel + conv3d_wino_kernel, conv3d_dma / conv3d_mfma (direct)",
                         "flops_counted": "EXECUTED on the matrix pipe (what the roofline bounds); in the reference's operator "
                                          "order the same launches are worth `reference_order_tflops`",
                         "reference_order_tflops": round(fl3_ref / (ms3 * 1e-3) / 1e12, 2) if ms3 > 0 else 0.0,
                         "measured_in": f"a second pass of the same {args.steps} steps with a HIP event pair around every conv "
                                        f"launch ({round(dt_instr / args.steps * 1e3, 3)} ms/step), outside the run that gives `value`",
                         "launches": n3, "avg_launch_us": round(ms3 / max(n3, 1) * 1e3, 2),
                         "executed_gflop_per_launch": round(fl3 / max(n3, 1) / 1e9, 2),
                         "algorithmic_gflop_per_launch": round(fl3_ref / max(n3, 1) / 1e9, 2),
                         # per entry point, to set beside the rocprofv3 kernel averages in profiles/: "conv3d_fwd" =
                         # conv3d_mfma_kernel<3,..>; "conv3d_fwd_staged" = prologue_apply_kernel (when the input has a
                         # prologue) + conv3d_dma_kernel<3,..>; "conv3d_ll_fwd" = conv3d_ll_kernel (executed FLOPs)
                         "by_entry": {what: {"launches": n, "avg_launch_us": round(ms / n * 1e3, 2),
                                             "tflops": round(fl / (ms * 1e-3) / 1e12, 2)}
                                      for (k, what), (n, ms, fl) in sorted(conv_by_entry.items()) if k == 3},
                         "input_transform_passes": {"launches": n0, "avg_launch_us": round(ms0 / max(n0, 1) * 1e3, 2),
                                                    "tb_per_s": round(by0 / (ms0 * 1e-3) / 1e12, 2) if ms0 > 0 else 0.0,
                                                    "note": "wino_input_kernel in front of the conv3d_wino_kernel launches that remain "
                                                            "(the 8x8 level: planes too narrow for conv3d_wf's 8x16 tiles): "
                                                            "HBM-bound (4 B read + 6.6 B (F(4,3)) / 8.8 B (F(2,3)) written per "
                                                            "input element), timed apart and NOT inside `achieved`; with them "
                                                            "the 3x3x3 launches run at "
                                                            f"{round(fl3 / ((ms3 + ms0) * 1e-3) / 1e12, 2) if ms3 > 0 else 0.0} "
                                                            "TFLOP/s executed = `frac_with_transforms`"},
                         # the 1x1x1 convolutions are bandwidth kernels: algorithmic bytes (input + output (+ residual) once) over
                         # their summed durations, against the box's 6.29 TB/s copy rate (tools/bench_hbm_kernels.py)
                         "k1_conv": {"launches": n1, "avg_launch_us": round(ms1 / max(n1, 1) * 1e3, 2), "bound": "hbm",
                                     "ms_per_step": round(ms1 / args.steps, 3),
                                     "tb_per_s": round(k1_bytes / (k1_ms * 1e-3) / 1e12, 2) if k1_ms > 0 else 0.0,
                                     "frac_of_copy_rate": round(k1_bytes / (k1_ms * 1e-3) / 1e12 / COPY_RATE_TB_S, 3) if k1_ms > 0 else 0.0,
                                     "copy_rate_tb_s": COPY_RATE_TB_S,
                                     "note": f"{n1 // max(args.steps, 1)} launches per step (the res_conv / Conv_2 launches that do not ride in a "
                                             "3x3x3 epilogue), most of them small (8x8 / 16x16 levels: 4-60 MB each, 10-30 us = launch + "
                                             "ramp + a short channel loop, not bandwidth); the two level-0 / level-1 ones, which also "
                                             "write conv20's prologue output, run at 4.9-5.1 TB/s"}},
        }
        if not args.no_extras:
            line["cond_cached"] = {"value": round(world * args.steps / dt_cached, 4), "unit": "batch32-steps/s",
                                   "note": "condition branch (62.82 of 172.39 GFLOP/sample, independent of x_t and t) "
                                           "evaluated once inside the timed run instead of every step; outputs are "
                                           "bit-identical (tests/test_gpu_sampling.py)"}
            line["bf16_compute"] = {"value": round(world * args.steps / dt_bf16, 4), "unit": "batch32-steps/s",
                                    "by_layer_class": bf16_levels,
                                    "unet_reference_order_tflops": round(BATCH * GFLOP_PER_SAMPLE * 1e-3 * args.steps / dt_bf16, 2),
                                    "note": "same full-forward steps with bf16 conv operands / fp32 accumulation "
                                            "(set_compute_dtype('bf16'), the config-3 mode; forward rel-L2 7e-3 vs fp32, "
                                            "tests/test_gpu_bf16.py) -- reduced precision, never `value`"}
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            line["parity"] = parity_check(dev)
            log(f"parity: PSNR {line['parity']['psnr_db']} dB")
            line["cpu_baseline"] = cpu_baseline()

    if not args.no_extras:
        # ---- finetune step (BASELINE configs[3] share of this GPU), a short run beside `value`, LAST: at N > 1 it is the
        # first code to push 123 MB gradient buckets through RCCL.  Everything measured so far is already in `line`; if
        # this leg raises, or a collective never completes (watchdog), the line is still printed, with the reason. -------
        del net, diff
        torch.cuda.empty_cache()
        tsteps = max(3, min(args.steps, 8))
        import threading

        def bail():
            # a stuck collective must not read as success: the headline line (measured before this leg) is still printed
            # by rank 0, then EVERY rank exits non-zero
            if rank == 0:
                line["train_step"] = {"error": "finetune leg did not finish within 240 s (collective stuck?); "
                                               "headline fields above were measured before it; exit status 3"}
                print(json.dumps(line), flush=True)
            sys.stderr.write(f"[bench] rank {rank}: finetune leg watchdog fired, exiting with status 3\n")
            sys.stderr.flush()
            os._exit(3)

        dog = threading.Timer(240.0, bail)
        dog.daemon = True
        dog.start()
        try:
            graph = world == 1 and os.environ.get("TMDIFF_BENCH_TRAIN_GRAPH", "1") != "0"
            eager = train_leg(dev, world, rank, dist, tsteps, 2) if graph else None
            train = train_object(train_leg(dev, world, rank, dist, tsteps, 2, graph=graph), world, tsteps, eager)
            log(f"train leg: {train['ms_per_step']} ms/step")
        except Exception as e:       # the headline line must survive a failure of this side measurement
            train = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
        if rank == 0:
            line["train_step"] = train
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
