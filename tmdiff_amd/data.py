"""PanCollection data either side of the hot path (reference data/__init__.py, data/LRHR_dataset.py:87-133).

``LRHRDataset`` reads the four arrays ``gt`` (absent in full-resolution test files: ``lms`` stands in), ``ms``, ``lms``,
``pan`` ([N, C, h, w] digital numbers) from an ``.h5`` file (needs h5py, which this image does not ship), an ``.npz`` file
or any mapping of arrays, divides by the sensor range (1023 for GaoFen-2 files -- "gf2" in the path -- else 2047) and
serves the dictionaries the trainer consumes: ``LR`` (ms), ``PAN``, ``MS`` (the upsampled lms), ``HR`` (gt),
``Res`` = HR - MS.
"""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .util import img2res


def _open(dataroot):
    if isinstance(dataroot, dict):
        return dataroot, "<mapping>"
    name = str(dataroot)
    if name.endswith(".npz"):
        return np.load(name), name
    try:
        import h5py
    except ImportError as e:        # pragma: no cover - h5py is absent in the build image
        raise ImportError(f"{name}: reading .h5 files needs h5py (not installed); convert to .npz or pass arrays") from e
    return h5py.File(name, "r"), name


class LRHRDataset(Dataset):
    def __init__(self, dataroot, data_len=-1, phase="train", img_scale=None):
        data, name = _open(dataroot)
        if img_scale is None:
            img_scale = 1023.0 if "gf2" in name else 2047.0
        self.img_scale, self.phase = img_scale, phase
        keys = set(data.keys())
        arr = lambda k: torch.from_numpy(np.array(data[k][...], dtype=np.float32) / img_scale)
        self.gt = arr("gt") if "gt" in keys else arr("lms")
        self.ms, self.lms, self.pan = arr("ms"), arr("lms"), arr("pan")
        n = self.ms.shape[0]
        self.data_len = n if data_len is None or data_len <= 0 else min(data_len, n)

    def __len__(self):
        return self.data_len

    def __getitem__(self, index):
        hr, ms = self.gt[index].float(), self.lms[index].float()
        return {"LR": self.ms[index].float(), "PAN": self.pan[index].float(), "MS": ms, "HR": hr, "Res": img2res(hr, ms)}


def create_dataset(dataset_opt, phase):
    return LRHRDataset(dataroot=dataset_opt["dataroot"], data_len=dataset_opt["data_len"], phase=phase)


create_dataset2 = create_dataset       # the reference keeps two identical factories (data/__init__.py:22-38)


def create_dataloader(dataset, dataset_opt, phase, generator=None):
    """Training loaders use the option file's batch size / shuffle / workers; validation is one item at a time."""
    if "train" in phase:
        return DataLoader(dataset, batch_size=dataset_opt["batch_size"], shuffle=bool(dataset_opt["use_shuffle"]),
                          num_workers=dataset_opt["num_workers"] or 0, pin_memory=True, generator=generator)
    return DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0, pin_memory=True)


def get_data_generator(loader):
    """Endless iterator over a loader (reference utils/util.py get_data_generator)."""
    while True:
        for batch in loader:
            yield batch
