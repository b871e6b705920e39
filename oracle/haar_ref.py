"""Oracle (test infrastructure): 2-D Haar DWT / IDWT in closed form, CPU fp32.

Restates DWT_IDWT/DWT_IDWT_layer.py:256-334 (DWT_2D), :337-430 (IDWT_2D) and
DWT_IDWT/DWT_IDWT_Functions.py:47-69, :89-112 for the only wavelet the hot path
uses ("haar", Hyper_unet_general.py:363-364).

The reference builds banded matrices L0/H0 [H/2,H] and L1/H1 [W,W/2] whose only
non-zero taps are +-s, s = float32(1/sqrt(2)), and evaluates

    L = L0 @ X ; Hh = H0 @ X ; LL = L @ L1 ; LH = L @ H1 ; HL = Hh @ L1 ; HH = Hh @ H1

With a = x[2i,2j], b = x[2i,2j+1], c = x[2i+1,2j], d = x[2i+1,2j+1] that is the
two-stage butterfly below (rows first, then columns, each stage scaled by s), which
keeps the reference's rounding order.  The IDWT is the transposed chain.
"""
import math

import torch

S = float(torch.tensor(1.0 / math.sqrt(2.0), dtype=torch.float32))  # the fp32 tap the reference uses


def haar_dwt2d(x: torch.Tensor):
    """x [..., H, W] (H, W even) -> (LL, LH, HL, HH), each [..., H/2, W/2]."""
    assert x.shape[-1] % 2 == 0 and x.shape[-2] % 2 == 0
    top, bot = x[..., 0::2, :], x[..., 1::2, :]
    lo = S * top + S * bot          # L  = L0 @ X
    hi = S * top - S * bot          # Hh = H0 @ X
    ll = S * lo[..., 0::2] + S * lo[..., 1::2]
    lh = S * lo[..., 0::2] - S * lo[..., 1::2]
    hl = S * hi[..., 0::2] + S * hi[..., 1::2]
    hh = S * hi[..., 0::2] - S * hi[..., 1::2]
    return ll, lh, hl, hh


def haar_idwt2d(ll, lh, hl, hh):
    """Inverse of :func:`haar_dwt2d`: four [..., h, w] bands -> [..., 2h, 2w]."""
    h, w = ll.shape[-2:]
    lo = ll.new_empty(*ll.shape[:-1], 2 * w)
    hi = ll.new_empty(*ll.shape[:-1], 2 * w)
    lo[..., 0::2] = S * ll + S * lh     # L  = LL @ L1^T + LH @ H1^T
    lo[..., 1::2] = S * ll - S * lh
    hi[..., 0::2] = S * hl + S * hh     # Hh = HL @ L1^T + HH @ H1^T
    hi[..., 1::2] = S * hl - S * hh
    out = ll.new_empty(*ll.shape[:-2], 2 * h, 2 * w)
    out[..., 0::2, :] = S * lo + S * hi  # L0^T @ L + H0^T @ Hh
    out[..., 1::2, :] = S * lo - S * hi
    return out
