#!/usr/bin/env python3
"""Winograd-along-n convolution (csrc/conv3d_wino.hip) beside the direct staged kernel, per layer shape of the bench
workload, plain input.  Usage: python tools/bench_conv_wino.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import fallback, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
EPI = "epi" in sys.argv[3:]      # with residual + second output (the in-network conv21 epilogue)
NB = 4 if "n4" in sys.argv[3:] else 8     # band count of the tensors (n4: GF-2 / QuickBird)
LAYERS = [("L0 32->32", 32, 32, 64), ("L0 32->64", 32, 64, 64), ("L0 64->64", 64, 64, 64), ("L1 64->128", 64, 128, 32),
          ("L1 128->128", 128, 128, 32), ("L2 256->256", 256, 256, 16), ("L3 256->256", 256, 256, 8)]


def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


for name, ci, co, h in LAYERS:
    x = torch.randn(B, ci, NB, h, h, device="cuda")
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") / (ci * 27) ** 0.5
    y = torch.empty(B, co, NB, h, h, device="cuda")
    ww, wd = ops.pack_conv_weight_wino(w, planes=fallback.wino_planes(NB)), ops.pack_conv_weight(w)
    wf = ops.pack_conv_weight_wino(w, planes=6, mode=2)
    kw = {}
    if EPI:
        res = torch.randn(B, co, NB, h, h, device="cuda")
        sc = torch.rand(B, co, device="cuda") + 0.5
        kw = dict(residual=res, emit=dict(act=True, scale=sc))
    f = t(lambda: ops.conv3d_wf([x], wf, co, **kw))
    a = t(lambda: fallback.conv3d_wino([x], ww, co, **kw))
    b = t(lambda: ops.conv3d([x], wd, co, 3, out=y))
    fl = 2.0 * B * co * ci * 27 * NB * h * h
    red = 27.0 / (9.0 * fallback.wino_planes(NB) / (fallback.wino_planes(NB) - 2))       # 2 for F(4,3), 1.5 for F(2,3)
    print(f"{name:12s} wf {f:6.3f} ms ({fl / 2 / f / 1e9:6.1f} executed) | winograd+pass {a:6.3f} ms ({fl / a / 1e9:6.1f} TFLOP/s in the direct count, {fl / red / a / 1e9:6.1f} executed) | "
          f"direct {b:6.3f} ms ({fl / b / 1e9:6.1f} TFLOP/s)", flush=True)
