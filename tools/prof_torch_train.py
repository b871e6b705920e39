#!/usr/bin/env python3
"""Which torch (ATen) kernels one finetune step launches beside the library's own, and from where: torch.profiler with stacks over
one eager step of the config-4 share (local batch 8); prints per ATen op the launch count, device time and the Python frames."""
import copy, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from tmdiff_amd.model import DDPM, EmaUpdater
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch
opt = {"phase": "train", "gpu_ids": [0], "distributed": False, "path": {"resume": None},
       "model": {"unet": {"channel_multiplier": [32, 64, 128, 256]}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
       "train": {"optimizer": {"lr": 1e-4}, "max_iter": 150000, "hip_graph": False}}
m = DDPM(opt); fill_weights_(m.netG.denoise_fn)
m.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "train")
ema = EmaUpdater(m, copy.deepcopy(m))
d = synthetic_tile_batch(1, 8, 8, 64, device="cuda"); d["LR"] = d["MS"]
def run(n):
    for i in range(n):
        m.feed_data(d); m.optimize_parameters("WV3"); ema.update(i + 1)
run(3); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    run(1); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    dt = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if "tmdiff_amd" in f or "torch/autograd" in f][:2]
    key = ev.name
    agg[key][0] += len(ev.kernels); agg[key][1] += dt
    agg[key][2][(" <- ".join(f.split("/")[-1] for f in frames) or "?", str(ev.input_shapes)[:60])] += 1
for k, (n, dt, where) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} {n:4d} kernels {dt / 1e3:8.3f} ms")
    for (w, shp), c in where.most_common(8):
        print(f"      {c:3d} x {w}  {shp}")
