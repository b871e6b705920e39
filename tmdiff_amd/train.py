"""Finetune / validation driver (reference general_sharpening_joint_random_batch_finetune.py:56-180) on tmdiff_amd.

    python -m tmdiff_amd.train -c config/general_finetune.json -p train            # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m tmdiff_amd.train -c ... -p train

Same option file, dataset names (``train_qb / train_gf2 / train_wv3``, ``val_QB / val_GF2 / val_WV3``), per-iteration
random choice of the training set with the reference's weights (4, 4, 8 per batch of QB, GF2, WV3: the band counts),
print / validation / checkpoint cadence and ``.mat`` outputs.  Multi-GPU is one process per GPU: every rank draws the
same dataset each iteration (same seeded ``random``) but its own batches (``batch_size // world`` items: the option
file's batch size is the global one, as under DataParallel), timesteps, noise and dropout masks, and ``DDPM.optimize_parameters`` SUM-
all-reduces the gradients over RCCL (the reference's DataParallel arithmetic, model.py:41).
"""
import argparse
import logging
import os
import random

import numpy as np
import torch

from . import config as Config
from . import data as Data
from . import dist as tdist
from . import evaluate
from . import model as Model

TRAIN_SETS = (("train_qb", "QB", 4), ("train_gf2", "GF2", 4), ("train_wv3", "WV3", 8))
VAL_SETS = (("val_QB", "QB"), ("val_GF2", "GF2"), ("val_WV3", "WV3"))


def seed_all(seed=3407, rank=0, reseed_python=True):
    """reference seed_torch (:24-33).  ``random`` (the per-iteration choice of the training set, shared by all ranks)
    is seeded identically everywhere; NumPy (the timesteps of p_losses_dynamic), torch and the device generator
    (noise, dropout masks) are offset by the rank, so that the ranks draw different (t, eps, mask) like the
    reference's DataParallel replicas do.  The MODEL must be built under rank 0's seeds (``build_replica``): the
    replicas of a data-parallel run start from identical weights."""
    if reseed_python:
        random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed + rank)
    torch.manual_seed(seed + rank)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed + rank)


def build_replica(create, rank, world, seed=3407):
    """Build this rank's model replica so that every rank holds the SAME weights (the reference's nn.DataParallel
    re-broadcasts device 0's parameters every step, networks.py:88-91; one process per GPU must start identical and
    stay identical through identical SUM-reduced gradients): the constructor and ``init_weights`` run under the
    shared seed (rank offset 0), then -- belt and braces against a ``strict=False`` checkpoint load or any
    rank-dependent initialisation -- every parameter and buffer is broadcast from rank 0, and only then are NumPy /
    torch / the device generator re-seeded with the rank offset (timesteps, noise, dropout masks differ per replica).
    ``create`` is a zero-argument callable returning an object with a ``netG`` module (model.DDPM) or a module."""
    seed_all(seed, rank=0)
    model = create()
    if world > 1:
        tdist.broadcast_module(getattr(model, "netG", model), src=0)
    seed_all(seed, rank=rank, reseed_python=False)
    return model


def per_rank_batch(batch_size, world):
    """The option file's ``batch_size`` is the GLOBAL batch: nn.DataParallel (reference networks.py:88-91) scatters it
    over the GPUs, so one process per GPU takes batch_size // world.  A global batch that the ranks cannot share evenly
    would silently change the effective batch, so it is refused."""
    batch_size, world = int(batch_size), max(1, int(world))
    if batch_size % world or batch_size < world:
        raise ValueError(f"global batch_size {batch_size} is not a positive multiple of the world size {world}: every "
                         f"rank takes batch_size // world items (nn.DataParallel scatter, reference networks.py:88-91)")
    return batch_size // world


def dataset_probabilities(lengths):
    """{name: number of batches} -> {name: probability}, weights 4 / 4 / 8 per batch (ref :158-160)."""
    w = {name: wt * lengths[name] for name, _, wt in TRAIN_SETS if name in lengths}
    total = float(sum(w.values()))
    return {k: v / total for k, v in w.items()}


def sample_dataset(probs, u=None):
    """One draw of the reference's sample_data (:45-53): cumulative thresholds in the order QB, GF2, WV3."""
    u = random.random() if u is None else u
    acc = 0.0
    names = [n for n, _, _ in TRAIN_SETS if n in probs]
    for name in names[:-1]:
        acc += probs[name]
        if u < acc:
            return name
    return names[-1]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, default="config/general_finetune.json")
    ap.add_argument("-p", "--phase", type=str, choices=["train", "val"], default="val")
    ap.add_argument("-gpu", "--gpu_ids", type=str, default=None)
    ap.add_argument("-debug", "-d", action="store_true")
    ap.add_argument("--root", type=str, default="experiments", help="where the run directory is created")
    ap.add_argument("--max-iter", type=int, default=None, help="override opt['train']['max_iter']")
    args = ap.parse_args(argv)

    world = tdist.init_from_env()
    rank = int(os.environ.get("RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    opt = Config.parse(args.config, args.phase, args.gpu_ids, args.debug, root=args.root, make_dirs=rank == 0)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s %(message)s")
    logger = logging.getLogger("base")
    logger.info(Config.dict2str(opt))
    seed_all(rank=rank)

    loaders, gens = {}, {}
    for name, ds_opt in opt["datasets"].items():
        is_train = name.startswith("train")
        if is_train and args.phase == "val":
            continue
        ds = Data.create_dataset(ds_opt, "train" if is_train else "val")
        if is_train and world > 1:
            ds_opt = dict(ds_opt, batch_size=per_rank_batch(ds_opt["batch_size"], world))
        g = torch.Generator().manual_seed(3407 + 1000 * rank) if is_train else None   # each rank its own batches
        loaders[name] = Data.create_dataloader(ds, ds_opt, name, generator=g)
        if is_train:
            gens[name] = Data.get_data_generator(loaders[name])
    logger.info("Initial Dataset Finished")

    # (seed_all(rank=rank) above only matters for the loaders; the model is built under the shared seed and synchronised)
    diffusion = build_replica(lambda: Model.create_model(opt), rank, world)
    logger.info("Initial Model Finished")
    step = diffusion.begin_step
    diffusion.set_new_noise_schedule(opt["model"]["beta_schedule"][opt["phase"]], schedule_phase=opt["phase"])
    results = opt["path"]["results"]

    def validate():
        scores = {}
        if rank == 0:                       # (validation items are few; the other ranks wait at the barrier)
            for name, prompt in VAL_SETS:
                if name in loaders:
                    scores.update(evaluate.val_dataset(diffusion, prompt, loaders[name], results, log=logger.info))
        if world > 1:
            torch.distributed.barrier()
        return scores

    if opt["phase"] == "train":
        probs = dataset_probabilities({n: len(l) for n, l in loaders.items() if n in gens})
        prompt_of = {n: p for n, p, _ in TRAIN_SETS}
        max_iter = args.max_iter if args.max_iter is not None else opt["train"]["max_iter"]
        while step < max_iter:
            name = sample_dataset(probs)
            step += 1
            diffusion.feed_data(next(gens[name]))
            diffusion.optimize_parameters(prompt_of[name])
            if step % opt["train"]["print_freq"] == 0:
                logs = diffusion.get_current_log()
                logger.info("iter %d [%s] %s", step, name, " ".join(f"{k}: {float(v):.4e}" for k, v in logs.items()))
            if step % opt["train"]["val_freq"] == 0:
                validate()
                if rank == 0:
                    diffusion.save_network(step)
        return step
    logger.info("Begin Model Evaluation.")
    return validate()


if __name__ == "__main__":
    main()
