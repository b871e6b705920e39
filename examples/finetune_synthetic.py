#!/usr/bin/env python3
"""BASELINE config 4 shaped finetune loop on synthetic PAN/MS tiles: global batch 64 = 8 GPUs x 8, ch 32-256,
AdamW lr 1e-4, dropout on, SUM gradient all-reduce over RCCL, EMA.  Single GPU:  python examples/finetune_synthetic.py
8 GPUs: python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/finetune_synthetic.py"""
import argparse, copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import dist as tdist
from tmdiff_amd.model import DDPM, EmaUpdater
from tmdiff_amd.util import synthetic_tile_batch

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--local-batch", type=int, default=8)
ap.add_argument("--channels", type=int, nargs=4, default=[32, 64, 128, 256])
args = ap.parse_args()
world = tdist.init_from_env()
rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
opt = {"phase": "train", "gpu_ids": [0], "distributed": world > 1, "path": {"resume": None, "checkpoint": "/tmp/tmdiff_ckpt"},
       "model": {"unet": {"channel_multiplier": args.channels}, "diffusion": {"loss_type": "l1"}, "init_type": "orthogonal"},
       "train": {"optimizer": {"lr": 1e-4}, "max_iter": 150000}}
torch.manual_seed(3407)                     # identical initial weights on every rank
model = DDPM(opt)
ema = EmaUpdater(model, copy.deepcopy(model))
model.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "train")
for it in range(args.iters):
    model.feed_data(synthetic_tile_batch(1000 * rank + it, args.local_batch, 8, 64))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.optimize_parameters("WV3")
    ema.update(it)
    torch.cuda.synchronize()
    if rank == 0:
        log = model.get_current_log()
        print(f"iter {it}: l_pix {float(log['l_pix']):.4f} lr {log['lr']:.2e} {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
