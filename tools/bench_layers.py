#!/usr/bin/env python3
"""Per-layer time of the convolution launches inside the benchmark step (batch-32 8x64x64, ch 32-256): HIP-event pairs
around every launch (ops.ConvTimer), grouped by entry point and shape.
Usage: python tools/bench_layers.py [steps] [c3] [bf16]     (c3: BASELINE configs[2], one 8x256x256 tile, ch 64-512)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops
from tmdiff_amd.Hyper_unet_general import WavBEST
from tmdiff_amd.diffusion_general import GeneralDiffusion
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
C3, BF16 = "c3" in sys.argv, "bf16" in sys.argv
dev = torch.device("cuda", 0)
net = fill_weights_(WavBEST(channels=[64, 128, 256, 512] if C3 else [32, 64, 128, 256])).to(dev).eval()
if BF16:
    net.set_compute_dtype("bf16")
diff = GeneralDiffusion(net, "l1").to(dev)
diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, dev)
d = synthetic_tile_batch(3407, 1 if C3 else 32, 8, 256 if C3 else 64, device=dev)
x = torch.randn_like(d["Res"])
for i in range(3):
    x = diff.p_sample(x, 999 - i, condition_x=d, prompt="WV3")
torch.cuda.synchronize()
ops.TIMER = ops.ConvTimer()
for i in range(steps):
    x = diff.p_sample(x, 990 - i, condition_x=d, prompt="WV3")
t, ops.TIMER = ops.TIMER, None
rows = sorted(t.summary(by_entry="layer").items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"conv launches: {tot / steps:.3f} ms per step")
for (k, what, tag), (n, ms, fl) in rows:
    rate = (f"{fl / (ms * 1e-3) / 1e12:6.1f} TFLOP/s executed" if k else f"{fl / (ms * 1e-3) / 1e12:6.2f} TB/s")
    if what.endswith("_k1"):
        what = what[:-3]
    print(f"{ms / steps:7.3f} ms/step  {n // steps:3d} x {ms / n * 1e3:7.1f} us  {rate}  {what:18s} {tag}")
