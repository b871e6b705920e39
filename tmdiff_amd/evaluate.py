"""Validation pass around the sampler: what ``val_dataset`` in the reference driver does
(general_sharpening_joint_random_batch_finetune.py:126-152) -- sample every validation item, rescale to the sensor's
digital numbers, write ``output_mulExm_{idx}.mat`` with key ``sr`` (H x W x C), and average SSIM / SAM against HR.
"""
import os
import time

import numpy as np
import scipy.io as scio

from . import metrics

IMG_SCALE = {"GF2": 1023.0}          # every other sensor: 2047 (ref :134)


def to_hwc01(img, min_max=(0.0, 1.0)):
    """[1,C,H,W] or [C,H,W] tensor -> clamped, rescaled H x W x C float32 array (ref ``normlization`` :39-42)."""
    lo, hi = min_max
    data = img.detach().squeeze().float().cpu().clamp(lo, hi).numpy()
    return np.transpose((data - lo) / (hi - lo), (1, 2, 0))


def val_dataset(trainer, dataset, val_loader, result_root, continous=False, log=print):
    """``trainer`` is a ``tmdiff_amd.model.DDPM`` (or the reference's); ``dataset`` is the prompt name.
    Returns ``{"ssim_<dataset>": ..., "sam_<dataset>": ..., "sec_per_item": ...}``."""
    result_path = os.path.join(result_root, dataset)
    os.makedirs(result_path, exist_ok=True)
    scale = IMG_SCALE.get(dataset, 2047.0)
    ssim_sum = sam_sum = 0.0
    n = 0
    t0 = time.time()
    for idx, val_data in enumerate(val_loader):
        trainer.feed_data(val_data)
        trainer.test(continous=continous, prompt=dataset)
        vis = trainer.get_current_visuals()
        sr = to_hwc01(vis["SR"][-1])                       # last image of the returned stack (ref :136)
        scio.savemat(os.path.join(result_path, f"output_mulExm_{idx}.mat"), {"sr": sr * scale})
        if "HR" in vis:
            hr = to_hwc01(vis["HR"])
            ssim_sum += metrics.ssim(hr, sr, 1)
            sam_sum += metrics.sam(hr, sr)
        n += 1
    score = {f"ssim_{dataset}": ssim_sum / max(n, 1), f"sam_{dataset}": sam_sum / max(n, 1),
             "sec_per_item": (time.time() - t0) / max(n, 1)}
    log(dataset, score)
    return score
