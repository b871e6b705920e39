#!/usr/bin/env python3
"""Headline benchmark: UNet denoise-steps/sec on batch-32 8-channel 64x64 tiles (BASELINE config 2).

A *step* is one ``GeneralDiffusion.p_sample`` call on the batch: the full WavBEST forward (both
branches, 172.39 conv-GFLOP per sample -- exactly the work the reference does per step) plus the
fused DDPM update.  Inputs are resident in HBM before the timed region.  fp32 throughout.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank denoises its own batch of 32 tiles (inference is batch-parallel: no collective on
the data path, "scaling": "weak"); value = N*K / max-over-ranks time.

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
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FULL = [32, 64, 128, 256]
BATCH, BANDS, SIZE, T = 32, 8, 64, 1000
GFLOP_PER_SAMPLE = 172.39          # algorithmic conv FLOPs per sample per forward
PEAK_FP32_MFMA = 157.3             # TFLOP/s, MI355X_MICROARCH.md


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
    """PSNR(build, oracle) of a short DDPM chain with shared noise (the "PSNR vs ref" half of the metric): full-width
    network, one 8x16x16 tile, T = 10; the oracle is the checker here, as in tests/ and smoke()."""
    from oracle import unet_ref as U
    from oracle.diffusion_ref import GeneralDiffusionRef
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.util import psnr, synthetic_tile_batch
    ref_net = U.fill_weights_(U.WavBESTRef(channels=FULL)).eval()
    net = WavBEST(channels=FULL)
    net.load_state_dict(ref_net.state_dict())
    net = net.to(dev).eval()
    d = synthetic_tile_batch(77, 1, BANDS, 16)
    noise = lambda like: torch.randn(like.shape, dtype=torch.float32)       # CPU generator on both sides
    want = GeneralDiffusionRef(ref_net, "l1")
    want.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, "cpu")
    got = GeneralDiffusion(net, "l1", noise_fn=noise).to(dev)
    got.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 10}, dev)
    torch.manual_seed(5)
    with torch.no_grad():
        y_ref = want.p_sample_loop(d, continous=False, prompt="WV3")
    torch.manual_seed(5)
    y = got.p_sample_loop({k: v.to(dev) for k, v in d.items()}, continous=False, prompt="WV3").cpu()
    return {"psnr_db": round(float(psnr(y, y_ref)), 1), "max_abs_diff": float((y - y_ref).abs().max()),
            "what": "10-step DDPM chain (T=10 cosine) on one 8x16x16 tile, ch 32-256, shared CPU noise: fused image of the "
                    "HIP path vs the CPU oracle; budget PSNR >= 60 dB (tests/test_gpu_sampling.py hold 50-step and "
                    "1000-step chains to the same)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ndev = torch.cuda.device_count()
    local_dev = local % max(ndev, 1)      # (rehearsals on a 1-GPU box map every rank to device 0)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL over xGMI; TMDIFF_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed on one GPU
        backend = os.environ.get("TMDIFF_BENCH_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", local_dev)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    from tmdiff_amd import ops
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.util import fill_weights_, synthetic_tile_batch

    net = fill_weights_(WavBEST(channels=FULL)).to(dev).eval()
    diff = GeneralDiffusion(net, "l1").to(dev)
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": T}, dev)
    d = synthetic_tile_batch(3407 + rank, BATCH, BANDS, SIZE, device=dev)      # resident in HBM from here on
    torch.manual_seed(1234 + rank)
    torch.cuda.manual_seed(1234 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(steps, first_t, x):
        for i in range(steps):
            x = diff.p_sample(x, first_t - i, condition_x=d, prompt="WV3")
        return x

    log(f"rank {rank}/{world}: model and inputs on {torch.cuda.get_device_name(dev)}")
    x = torch.randn_like(d["Res"])
    x = run(args.warmup, T - 1, x)
    torch.cuda.synchronize()
    log("warm-up done")
    # ---- timed region: K full steps ---------------------------------------------------------------
    ops.TIMER = ops.ConvTimer()
    barrier()
    t0 = time.perf_counter()
    x = run(args.steps, T - 1 - args.warmup, x)
    barrier()
    dt = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    conv = timer.summary()
    conv_by_entry = timer.summary(by_entry=True)
    assert torch.isfinite(x).all()
    log(f"timed region: {args.steps} steps in {dt:.3f} s")

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
    net.set_compute_dtype("fp32")
    assert torch.isfinite(x3).all()
    log(f"bf16-compute loop: {args.steps} steps in {dt_bf16:.3f} s")

    if dist is not None:
        tt = torch.tensor([dt, dt_cached, dt_bf16], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_cached, dt_bf16 = float(tt[0]), float(tt[1]), float(tt[2])

    if rank == 0:
        n3, ms3, fl3 = conv.get(3, (0, 0.0, 0.0))
        n1, ms1, fl1 = conv.get(1, (0, 0.0, 0.0))
        achieved = fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else 0.0
        line = {
            "metric": "UNet denoise-steps/sec (8-ch 64x64, batch 32)",
            "value": round(world * args.steps / dt, 4),
            "unit": "batch32-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch-32 8-ch 64x64 tiles, DDPM p_sample steps of the "
                                   "T=1000 cosine schedule, channel_multiplier [32,64,128,256], full UNet forward "
                                   "(both branches) + fused DDPM update per step",
                       "batch_per_gpu": BATCH, "tile": [BANDS, SIZE, SIZE], "parallelism": f"batch-parallel x{world}",
                       "weights": "key-hashed random init", "text_embedding": "fixed synthetic 768-d"},
            "sample_steps_per_s": round(world * BATCH * args.steps / dt, 2),
            "unet_tflops": round(BATCH * GFLOP_PER_SAMPLE * 1e-3 * args.steps / dt, 2),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA, 4), "traffic": None,
                         "traffic_note": "not collected live; PMC passes over the same layer shapes (profiles/"
                                         "r01_conv_pmc_microbench.txt): 14.7 GB per B=32 forward over its 51 3x3x3 launches "
                                         "= 288 MB per launch against 188 MB of input + output bytes",
                         "kernel": "3x3x3 conv launches, fp32 v_mfma_f32_32x32x2_f32: conv3d_mfma_kernel<3,..> (fused prologue) "
                                   "or prologue_apply_kernel + conv3d_dma_kernel<3,..> (staged), chosen per layer",
                         "launches": n3, "avg_launch_us": round(ms3 / max(n3, 1) * 1e3, 2),
                         "algorithmic_gflop_per_launch": round(fl3 / max(n3, 1) / 1e9, 2),
                         # per entry point, to set beside the rocprofv3 kernel averages in profiles/: "conv3d_fwd" =
                         # conv3d_mfma_kernel<3,..>; "conv3d_fwd_staged" = prologue_apply_kernel (when the input has a
                         # prologue) + conv3d_dma_kernel<3,..>
                         "by_entry": {what: {"launches": n, "avg_launch_us": round(ms / n * 1e3, 2),
                                             "tflops": round(fl / (ms * 1e-3) / 1e12, 2)}
                                      for (k, what), (n, ms, fl) in sorted(conv_by_entry.items()) if k == 3},
                         "k1_conv": {"launches": n1, "avg_launch_us": round(ms1 / max(n1, 1) * 1e3, 2),
                                     "tflops": round(fl1 / (ms1 * 1e-3) / 1e12, 2) if ms1 > 0 else 0.0}},
            "cond_cached": {"value": round(world * args.steps / dt_cached, 4), "unit": "batch32-steps/s",
                            "note": "condition branch (62.82 of 172.39 GFLOP/sample, independent of x_t and t) "
                                    "evaluated once inside the timed run instead of every step; outputs are "
                                    "bit-identical (tests/test_gpu_sampling.py)"},
            "bf16_compute": {"value": round(world * args.steps / dt_bf16, 4), "unit": "batch32-steps/s",
                             "unet_tflops": round(BATCH * GFLOP_PER_SAMPLE * 1e-3 * args.steps / dt_bf16, 2),
                             "note": "same full-forward steps with bf16 conv operands / fp32 accumulation "
                                     "(set_compute_dtype('bf16'), the config-3 mode; forward rel-L2 7e-3 vs fp32, "
                                     "tests/test_gpu_bf16.py) -- reduced precision, never `value`"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["parity"] = parity_check(dev)
            log(f"parity: PSNR {line['parity']['psnr_db']} dB")
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
