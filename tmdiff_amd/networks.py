"""Factory boundary: ``define_General(opt)`` (reference GeneralModel/networks.py:80-92).

Reads the same option keys as the reference -- ``opt['model']['unet']['channel_multiplier']``,
``opt['model']['diffusion']['loss_type']``, ``opt['model']['init_type']``, ``opt['phase']``,
``opt['gpu_ids']``, ``opt['distributed']`` -- plus two optional keys of ours,
``opt['model']['text_embeddings']`` (dict or path; default: fixed synthetic vectors, because the CLIP
weights are not shipped) and ``opt['model']['compute_dtype']`` ("fp32" default | "bf16": bf16-operand convs for
inference).  Missing keys behave like the reference's ``NoneDict`` (-> None).

Multi-GPU: the reference wraps the module in single-process ``nn.DataParallel`` (:88-91).  Here the
unit of parallelism is one process per GPU (``tmdiff_amd.dist``), so with ``distributed`` set the
module is returned unwrapped and the caller shards the batch / all-reduces gradients over RCCL.
"""
import functools
import logging

import torch
import torch.nn as nn
from torch.nn import init

from . import Hyper_unet_general as unet
from . import diffusion_general as diffusion

logger = logging.getLogger("base")


def _get(d, *keys):
    for k in keys:
        if d is None:
            return None
        try:
            d = d[k]
        except (KeyError, TypeError):
            return None
    return d


def weights_init_normal(m, std=0.02):
    name = m.__class__.__name__
    if name.find("Conv") != -1 or name.find("Linear") != -1:
        init.normal_(m.weight.data, 0.0, std)
        if m.bias is not None:
            m.bias.data.zero_()
    elif name.find("BatchNorm2d") != -1:
        init.normal_(m.weight.data, 1.0, std)
        init.constant_(m.bias.data, 0.0)


def weights_init_kaiming(m, scale=1):
    name = m.__class__.__name__
    # the reference tests for 'Conv2d' (:33), so Conv3d layers keep their default init under 'kaiming'
    if name.find("Conv2d") != -1 or name.find("Linear") != -1:
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
        m.weight.data *= scale
        if m.bias is not None:
            m.bias.data.zero_()
    elif name.find("BatchNorm2d") != -1:
        init.constant_(m.weight.data, 1.0)
        init.constant_(m.bias.data, 0.0)


def weights_init_orthogonal(m):
    name = m.__class__.__name__
    if name.find("Conv") != -1 or name.find("Linear") != -1:
        init.orthogonal_(m.weight.data, gain=1)
        if m.bias is not None:
            m.bias.data.zero_()
    elif name.find("BatchNorm2d") != -1:
        init.constant_(m.weight.data, 1.0)
        init.constant_(m.bias.data, 0.0)


def init_weights(net, init_type="kaiming", scale=1, std=0.02):
    logger.info("Initialization method [{:s}]".format(init_type))
    if init_type == "normal":
        net.apply(functools.partial(weights_init_normal, std=std))
    elif init_type == "kaiming":
        net.apply(functools.partial(weights_init_kaiming, scale=scale))
    elif init_type == "orthogonal":
        net.apply(weights_init_orthogonal)
    else:
        raise NotImplementedError("initialization method [{:s}] not implemented".format(init_type))


def define_General(opt):
    model_opt = opt["model"]
    model = unet.WavBEST(channels=_get(model_opt, "unet", "channel_multiplier"),
                         text_embeddings=_get(model_opt, "text_embeddings"))
    if _get(model_opt, "compute_dtype") is not None:      # ours, optional: "fp32" (default) | "bf16" (inference convs)
        model.set_compute_dtype(_get(model_opt, "compute_dtype"))
    netG = diffusion.GeneralDiffusion(denoise_fn=model, loss_type=_get(model_opt, "diffusion", "loss_type"))
    if _get(opt, "phase") == "train":
        init_weights(netG, init_type=_get(model_opt, "init_type"))
    if _get(opt, "gpu_ids") and _get(opt, "distributed"):
        assert torch.cuda.is_available()
        logger.info("distributed_training: one process per GPU over RCCL (tmdiff_amd.dist), no DataParallel wrap")
    return netG
