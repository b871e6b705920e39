#!/usr/bin/env python3
"""Host-side (Python) cost of one finetune step: cProfile over a few steps of the config-4 share (local batch 8)."""
import cProfile, copy, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tmdiff_amd.model import DDPM, EmaUpdater
from tmdiff_amd.util import fill_weights_, synthetic_tile_batch
opt = {"phase": "train", "gpu_ids": [0], "distributed": False, "path": {"resume": None},
       "model": {"unet": {"channel_multiplier": [32, 64, 128, 256]}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
       "train": {"optimizer": {"lr": 1e-4}, "max_iter": 150000}}
m = DDPM(opt); fill_weights_(m.netG.denoise_fn)
m.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "train")
ema = EmaUpdater(m, copy.deepcopy(m))
d = synthetic_tile_batch(1, 8, 8, 64, device="cuda"); d["LR"] = d["MS"]
def run(n):
    for i in range(n):
        m.feed_data(d); m.optimize_parameters("WV3"); ema.update(i + 1)
run(3); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(5); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
