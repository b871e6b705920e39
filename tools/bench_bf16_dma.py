#!/usr/bin/env python3
"""The packed-input bf16 convolution kernel alone (input already in bf16 units, as its producer leaves it), per layer
shape of the bench workload: y only / y + residual + packed second output / packed output only.  Per case: time, TFLOP/s
(dense bf16 MFMA peak 2 500) and the ALGORITHMIC HBM bytes (packed bf16 input 2 B per element, fp32 output / residual 4 B,
packed second output 2 B; weights and halo re-reads not counted) over the time, as a fraction of the 6.29 TB/s copy rate --
so that which roof a layer sits under is a number.
Usage: python tools/bench_bf16_dma.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
LAYERS = [("L0 32->32", 32, 32, 64, 10), ("L0 32->64", 32, 64, 64, 2), ("L0 64->64", 64, 64, 64, 4), ("L0 96->32", 96, 32, 64, 1),
          ("L1 64->64", 64, 64, 32, 2), ("L1 64->128", 64, 128, 32, 2), ("L1 128->128", 128, 128, 32, 4), ("L1 192->32", 192, 32, 32, 1),
          ("L2 128->128", 128, 128, 16, 2), ("L2 128->256", 128, 256, 16, 2), ("L2 256->256", 256, 256, 16, 4),
          ("L3 256->256", 256, 256, 8, 4), ("L3 768->128", 768, 128, 8, 1)]

def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3

tot = [0.0, 0.0, 0.0]; totf = 0.0
for name, ci, co, h, cnt in LAYERS:
    plane = 8 * h * h
    xp = torch.randn(B, ci // 8, plane, 8, device="cuda").to(torch.bfloat16).view(torch.int16)
    w = ops.pack_conv_weight_bf16(torch.randn(co, ci, 3, 3, 3, device="cuda") / (ci * 27) ** 0.5)
    y = torch.empty(B, co, 8, h, h, device="cuda")
    res = torch.randn(B, co, 8, h, h, device="cuda")
    sc = torch.rand(B, co, device="cuda") + 0.5
    shp = (8, h, h)
    a = t(lambda: ops.conv3d([xp], w, co, 3, math="bf16", out=y, x_bf16_shape=shp))
    b = t(lambda: ops.conv3d([xp], w, co, 3, math="bf16", out=y, x_bf16_shape=shp, residual=res, emit=dict(act=True, scale=sc)))
    c = t(lambda: ops.conv3d([xp], w, co, 3, math="bf16", keep_y=False, x_bf16_shape=shp, emit=dict(act=True, scale=sc)))
    fl = 2.0 * B * co * ci * 27 * plane
    by = [B * plane * (2 * ci + k * co) for k in (4, 4 + 4 + 2, 2)]        # algorithmic bytes of the three cases
    hb = lambda nb, us: f"{nb / us / 1e6:4.2f} TB/s = {nb / us / 1e6 / 6.29:4.2f} of copy rate, {fl / us / 1e6 / 2500:4.2f} of MFMA peak"
    print(f"{name:12s} y {a:7.1f} us {fl / a / 1e6:7.1f} TF ({hb(by[0], a)}) | y+res+y2 {b:7.1f} us {fl / b / 1e6:7.1f} TF ({hb(by[1], b)}) | "
          f"y2 only {c:7.1f} us {fl / c / 1e6:7.1f} TF ({hb(by[2], c)})  x{cnt}", flush=True)
    for i, v in enumerate((a, b, c)): tot[i] += v * cnt
    totf += fl * cnt
print("weighted: " + " | ".join(f"{v / 1e3:.2f} ms {totf / v / 1e6:.0f} TF" for v in tot))
