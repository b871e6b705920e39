"""Option files of the reference driver (core/logger.py:20-125): JSON with ``//`` comments -> nested ``NoneDict``.

Same keys and path handling (every ``opt['path']`` entry except resume / experiments ones becomes
``<root>/<name>_<timestamp>/<entry>`` and is created).  Differences, both forced by one-process-per-GPU launches:
``CUDA_VISIBLE_DEVICES`` is not rewritten here (the launcher owns it) and ``distributed`` also turns on when
``WORLD_SIZE`` > 1.
"""
import json
import os
from collections import OrderedDict
from datetime import datetime


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def dict2str(opt, indent_l=1):
    msg = ""
    for k, v in opt.items():
        pad = " " * (indent_l * 2)
        if isinstance(v, dict):
            msg += f"{pad}{k}:[\n{dict2str(v, indent_l + 1)}{pad}]\n"
        else:
            msg += f"{pad}{k}: {v}\n"
    return msg


def load_json_with_comments(path):
    with open(path, "r", encoding="UTF-8") as f:
        text = "\n".join(line.split("//")[0] for line in f)
    return json.loads(text, object_pairs_hook=OrderedDict)


def parse(config, phase, gpu_ids=None, debug=False, root="experiments", make_dirs=True):
    """``config``: path of the option file; ``phase``: 'train' | 'val'; ``gpu_ids``: '0,1' style override."""
    opt = load_json_with_comments(config)
    if debug:
        opt["name"] = "debug_{}".format(opt["name"])
    experiments_root = os.path.join(root, "{}_{}".format(opt["name"], datetime.now().strftime("%y%m%d_%H%M%S")))
    opt["path"]["experiments_root"] = experiments_root
    for key, path in list(opt["path"].items()):
        if "resume" not in key and "experiments" not in key and "pan2ms" not in key and "ms2pan" not in key:
            opt["path"][key] = os.path.join(experiments_root, path)
            if make_dirs:
                os.makedirs(opt["path"][key], exist_ok=True)
    opt["phase"] = phase
    if gpu_ids is not None:
        opt["gpu_ids"] = [int(i) for i in str(gpu_ids).split(",")]
    opt["distributed"] = len(opt.get("gpu_ids") or []) > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1
    if "debug" in opt["name"]:           # (the reference's debug overrides, for the dataset names this driver uses)
        opt["train"].update(val_freq=2, print_freq=2, save_checkpoint_freq=3)
        for sched in opt["model"]["beta_schedule"].values():
            sched["n_timestep"] = 10
        for name, ds in opt["datasets"].items():
            ds["data_len"] = 6 if name.startswith("train") else 3
            if name.startswith("train"):
                ds["batch_size"] = 2
    return dict_to_nonedict(opt)
