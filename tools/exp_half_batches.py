#!/usr/bin/env python3
"""Experiment: the benchmark step on 32 tiles as ONE batch against the same 32 tiles as k sub-batches run back to back (does the
256 MB Infinity Cache reward smaller producer -> consumer tensors more than the extra launches and shorter grids cost?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd.Hyper_unet_general import WavBEST
from tmdiff_amd.diffusion_general import GeneralDiffusion
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch

dev = torch.device("cuda")
net = fill_weights_(WavBEST(channels=[32, 64, 128, 256])).to(dev).eval()
diff = GeneralDiffusion(net, "l1").to(dev)
diff.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, dev)
d = synthetic_tile_batch(3407, 32, 8, 64, device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for parts in (1, 2, 4, 1, 2):
    n = 32 // parts
    subs = [{k: v[i * n:(i + 1) * n].contiguous() for k, v in d.items()} for i in range(parts)]
    xs = [torch.randn_like(s["Res"]) for s in subs]
    def run(k):
        for i in range(k):
            for j, s in enumerate(subs):
                xs[j] = diff.p_sample(xs[j], 999 - i, condition_x=s, prompt="WV3")
    run(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
    print(f"{parts} x batch {n}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per 32-tile step", flush=True)
