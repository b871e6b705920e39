"""Boundary helpers of the hot path (reference utils/util.py:135-142) and the PSNR used to compare
the build with the oracle (core/metrics.py:72-85 semantics: per-band PSNR, data_range 1, mean)."""
import torch


def res2img(img_, img_lr_up):
    """residual + up-sampled MS = fused image (utils/util.py:135-137)."""
    if img_.is_cuda:
        from . import ops
        return ops.add(img_.contiguous(), img_lr_up.contiguous())
    return img_ + img_lr_up


def img2res(x, img_lr_up):
    """fused image - up-sampled MS = residual (utils/util.py:140-142)."""
    if x.is_cuda:
        from . import ops
        return ops.add(x.contiguous(), img_lr_up.contiguous(), sign_b=-1.0)
    return x - img_lr_up


def psnr(a, b, data_range=1.0):
    """Mean over bands of 10*log10(range^2 / MSE_band); a, b: [..., C, H, W] (or [C,H,W])."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    mse = ((a - b) ** 2).flatten(-2).mean(-1)            # per band (and per batch element)
    vals = 10.0 * torch.log10(data_range ** 2 / mse.clamp_min(1e-300))
    return float(vals.mean())


def fill_weights_(module, seed=0):
    """Deterministic key-hashed weight filler for benchmarks and tests (no weight files are shipped):
    every fp32 matrix/kernel in ``state_dict`` gets N(0, 1/fan_in), biases N(0, 0.05^2) -- except the
    ``Dense`` projections (modulation scales / shifts), mean 1 -- from a CPU generator seeded by
    crc32(key) ^ seed.  Same recipe as the oracle's filler, so both sides hold identical weights."""
    import math
    import zlib
    with torch.no_grad():
        for key, ten in module.state_dict().items():
            if not ten.is_floating_point() or "clip_text_model" in key:
                continue
            g = torch.Generator(device="cpu").manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
            if ten.dim() >= 2:
                ten.copy_(torch.randn(ten.shape, generator=g) * math.sqrt(1.0 / ten[0].numel()))
            elif key.endswith("bias"):
                mean = 1.0 if key.endswith(".dense.bias") else 0.0
                ten.copy_(mean + torch.randn(ten.shape, generator=g) * 0.05)
            elif key.endswith(".b"):                     # NIN bias of AttnBlockpp
                ten.copy_(torch.randn(ten.shape, generator=g) * 0.05)
            elif key.endswith("weight"):
                ten.copy_(1.0 + torch.randn(ten.shape, generator=g) * 0.1)
    return module


def synthetic_tile_batch(seed, b, c, h, w=None, device="cpu"):
    """SURVEY 8(d) synthetic inputs: MS, PAN, HR ~ U[0,1) from a seeded CPU generator, Res = HR - MS."""
    w = h if w is None else w
    g = torch.Generator(device="cpu").manual_seed(seed)
    ms = torch.rand(b, c, h, w, generator=g)
    pan = torch.rand(b, 1, h, w, generator=g)
    hr = torch.rand(b, c, h, w, generator=g)
    x_t = torch.randn(b, c, h, w, generator=g)
    out = {"MS": ms, "PAN": pan, "HR": hr, "Res": hr - ms, "x_t": x_t}
    return {k: v.to(device).contiguous() for k, v in out.items()}
