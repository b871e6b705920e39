#!/usr/bin/env python3
"""The bench workload (BASELINE configs[1]: batch 32, 8-band 64x64, ch 32-256) in the bf16 compute mode, for
`rocprofv3 --kernel-trace --stats -- python3 tools/prof_bf16_step.py [steps]`; prints steps/s of the un-profiled loop
when run bare.  TMDIFF_PROF_DTYPE=fp32 profiles the default mode instead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd.Hyper_unet_general import WavBEST
from tmdiff_amd.diffusion_general import GeneralDiffusion
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
net = fill_weights_(WavBEST(channels=(32, 64, 128, 256))).to(dev).eval()
diff = GeneralDiffusion(net, "l1").to(dev)
diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, dev)
d = synthetic_tile_batch(3407, B, 8, 64, device=dev)
net.set_compute_dtype(os.environ.get("TMDIFF_PROF_DTYPE", "bf16"))
x = torch.randn_like(d["Res"])
for i in range(3):
    x = diff.p_sample(x, 999 - i, condition_x=d, prompt="WV3")
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    x = diff.p_sample(x, 990 - i, condition_x=d, prompt="WV3")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{STEPS} steps in {dt * 1e3:.1f} ms = {dt / STEPS * 1e3:.3f} ms/step = {STEPS / dt:.1f} steps/s", flush=True)
