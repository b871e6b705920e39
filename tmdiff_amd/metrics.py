"""Evaluation metrics of the validation loop: MPSNR, SAM, SSIM with the semantics of the reference's
core/metrics.py:56-112 (which delegates to skimage: PSNR per band with data_range, SAM = mean spectral angle in
degrees with NaNs -> 0, SSIM = skimage.structural_similarity defaults: 7x7 uniform window, K1 = 0.01, K2 = 0.03,
sample covariance, border of 3 pixels cropped, mean over bands).  Inputs are [H, W, C] (reference layout) or
[..., C, H, W] tensors on any device; evaluation only -- not on the hot path."""
import math

import torch
import torch.nn.functional as F


def _chw(x, hwc):
    x = torch.as_tensor(x).double()
    return x.permute(2, 0, 1) if hwc else x


def mpsnr(x_true, x_pred, data_range=1.0, hwc=True):
    a, b = _chw(x_true, hwc), _chw(x_pred, hwc)
    mse = ((a - b) ** 2).flatten(-2).mean(-1)
    return float((10.0 * torch.log10(data_range ** 2 / mse)).mean())


def sam(x_true, x_pred, hwc=True):
    a, b = _chw(x_true, hwc), _chw(x_pred, hwc)
    dot = (a * b).sum(-3)
    ang = torch.arccos(dot / b.norm(dim=-3) / a.norm(dim=-3))
    ang = torch.where(torch.isnan(ang), torch.zeros_like(ang), ang)
    return float(ang.mean() * 180.0 / math.pi)


def ssim(x_true, x_pred, data_range=1.0, hwc=True, win=7):
    a, b = _chw(x_true, hwc), _chw(x_pred, hwc)
    a, b = a.reshape(-1, 1, *a.shape[-2:]), b.reshape(-1, 1, *b.shape[-2:])
    k = torch.ones(1, 1, win, win, dtype=a.dtype, device=a.device) / (win * win)
    f = lambda t: F.conv2d(t, k)                       # 'valid' uniform filter == skimage's crop of the border
    npx = win * win
    cov_norm = npx / (npx - 1.0)                        # sample covariance
    ux, uy = f(a), f(b)
    vx = cov_norm * (f(a * a) - ux * ux)
    vy = cov_norm * (f(b * b) - uy * uy)
    vxy = cov_norm * (f(a * b) - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.flatten(1).mean(1).mean())
