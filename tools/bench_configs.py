#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (not the headline bench): BASELINE configs 1, 3 (fp32 and bf16 compute), the
finetune step of config 4 and the per-GPU share of config 5 (mixed 4-band GF-2 / 8-band WV-3 tiles) on ONE GPU.
Prints one line per config.  Usage: python tools/bench_configs.py [c1] [c3] [c4] [c5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tmdiff_amd.Hyper_unet_general import WavBEST
from tmdiff_amd.diffusion_general import GeneralDiffusion
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch

def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, out

which = sys.argv[1:] or ["c1", "c3", "c4", "c5"]
if "c1" in which:   # single 8-ch 64x64 tile, 50-step DDPM
    net = fill_weights_(WavBEST(channels=[32, 64, 128, 256])).cuda().eval()
    diff = GeneralDiffusion(net, "l1").cuda(); diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 50}, "cuda")
    d = synthetic_tile_batch(3407, 1, 8, 64, device="cuda")
    diff.super_resolution(d, False, "WV3", 3.0)
    dt, out = sync_time(lambda: diff.super_resolution(d, False, "WV3", 3.0))
    print(f"config1: B=1 8x64x64, 50-step DDPM super_resolution: {dt:.3f} s -> {50 / dt:.1f} denoise-steps/s (eager launches, cond branch cached)", flush=True)
    # the same loop replayed from a HIP graph of one step (launch-bound at B=1)
    x = torch.randn_like(d["Res"]); noise = torch.randn_like(x)
    diff.noise_fn = lambda like: noise
    net.begin_condition_cache(d["PAN"], d["MS"], "QB")
    diff.p_sample(x, 10, condition_x=d, prompt="QB"); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = diff.p_sample(x, 10, condition_x=d, prompt="QB")
    dt, _ = sync_time(lambda: [g.replay() for _ in range(50)])
    net.end_condition_cache()
    print(f"config1: same step as a captured HIP graph: {dt / 50 * 1e3:.2f} ms/step -> {50 / dt:.1f} denoise-steps/s", flush=True)
    del net, diff
if "c3" in which:   # WorldView-3 config: ch 64-512, 8-ch 256x256, DPM-Solver 20 steps (21 NFE), fp32 and bf16 compute
    net = fill_weights_(WavBEST(channels=[64, 128, 256, 512])).cuda().eval()
    diff = GeneralDiffusion(net, "l1").cuda(); diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    d = synthetic_tile_batch(3407, 1, 8, 256, device="cuda")
    gf = 172.39 * 4.0 * 16      # ch x2 -> ~4x, 256x256 -> 16x
    outs = {}
    for mode in ("fp32", "bf16"):
        net.set_compute_dtype(mode)
        torch.manual_seed(1); diff.sample_by_dpmsolver(d, "WV3", steps=20)
        torch.manual_seed(1)
        dt, outs[mode] = sync_time(lambda: diff.sample_by_dpmsolver(d, "WV3", steps=20))
        print(f"config3 ({mode}): B=1 8x256x256 ch 64-512, DPM-Solver++ 20 steps (21 NFE): {dt:.3f} s -> {21 / dt:.2f} NFE/s, "
              f"~{(62.82 + 109.57 * 21) / 172.39 * gf / dt / 1e3:.1f} TFLOP/s executed", flush=True)
    from tmdiff_amd.util import psnr
    print(f"config3: PSNR(bf16, fp32) = {psnr(outs['bf16'], outs['fp32']):.1f} dB", flush=True)
    del net, diff
if "c4" in which:   # finetune step, local batch 8, ch 32-256, AdamW, dropout on
    net = fill_weights_(WavBEST(channels=[32, 64, 128, 256])).cuda().train()
    diff = GeneralDiffusion(net, "l1").cuda(); diff.set_loss("cuda")
    diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-4)
    d = synthetic_tile_batch(3407, 8, 8, 64, device="cuda")
    def step():
        opt.zero_grad(set_to_none=True); loss = diff(d, "WV3").sum(); loss.backward(); opt.step(); return float(loss.detach())
    step(); step()
    dt, loss = sync_time(lambda: [step() for _ in range(5)])
    print(f"config4 (1 GPU share): train step local batch 8, 8x64x64, dropout on: {dt / 5 * 1e3:.1f} ms/step "
          f"({3 * 8 * 172.39 / (dt / 5) / 1e3:.1f} TFLOP/s at 3x forward FLOPs), loss {loss[-1]:.4f}", flush=True)
if "c5" in which:   # multi-satellite config: a 512x512 scene = 64 tiles of 64x64; one GPU's share of a mixed GF-2 (4 bands,
    # prompt "GF2") / WV-3 (8 bands, "WV3") job is a batch of each.  The reference never mixes band counts within a batch
    # (general_sharpening...py:45-53): two sub-batches of 32 tiles, 10 DDPM steps each, condition branch cached.
    net = fill_weights_(WavBEST(channels=[32, 64, 128, 256])).cuda().eval()
    diff = GeneralDiffusion(net, "l1").cuda(); diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cuda")
    for bands, prompt in ((4, "GF2"), (8, "WV3")):
        d = synthetic_tile_batch(3407 + bands, 32, bands, 64, device="cuda")
        x = torch.randn_like(d["Res"])
        net.begin_condition_cache(d["PAN"], d["MS"], prompt)
        for i in range(3):
            x = diff.p_sample(x, 999 - i, condition_x=d, prompt=prompt)
        def run(x=x):
            for i in range(10):
                x = diff.p_sample(x, 990 - i, condition_x=d, prompt=prompt)
            return x
        dt, _ = sync_time(run)
        net.end_condition_cache()
        gf = 109.57 * bands / 8          # per sample per step, condition branch cached
        print(f"config5 ({prompt}, {bands} bands): 32 tiles of {bands}x64x64, condition branch cached: {dt / 10 * 1e3:.2f} ms per step "
              f"-> {32 * 10 / dt:.0f} tile-steps/s ({32 * gf / (dt / 10) / 1e3:.1f} TFLOP/s in the reference's operator order)", flush=True)
