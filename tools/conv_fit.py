#!/usr/bin/env python3
"""Decompose conv3d time into per-chunk slope and per-round fixed cost: Cin sweep x batch sweep, Cout=32 (4x1 tile) and 64 (2x2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

MATH = sys.argv[1] if len(sys.argv) > 1 else "fp32"


def run(B, ci, co, h, reps=10):
    x = torch.randn(B, ci, 8, h, h, device="cuda")
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") / (ci * 27) ** 0.5
    wp = (ops.pack_conv_weight_bf16 if MATH == 'bf16' else ops.pack_conv_weight)(w); y = torch.empty(B, co, 8, h, h, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5
    f = lambda: ops.conv3d([x], wp, co, 3, in_scale=sc, in_act=True, out=y, math=MATH)
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for co in (32, 64):
    for B in (4, 8, 16, 32):
        wgs = B * 2 * 8 * (4 if co == 32 else 8)
        row = []
        for ci in (32, 64, 128, 256):
            us = run(B, ci, co, 64)
            row.append(us)
        slope = (row[3] - row[0]) / ((256 - 32) / 8)
        icpt = row[0] - slope * 4
        print(f"co={co} B={B:2d} wgs={wgs:5d} rounds={wgs/512:5.2f}  t(ci=32,64,128,256)=" + " ".join(f"{r:8.1f}" for r in row) +
              f" us   slope {slope:6.2f} us/8ch  intercept {icpt:6.1f} us  (per round: slope {slope/max(1,wgs/512):5.2f}, icpt {icpt/max(1,wgs/512):5.1f})", flush=True)
