#!/usr/bin/env python3
"""Per-layer conv3d micro-benchmark at the config-2 shapes (B x 8-ch 64x64, ch 32-256).
Times each distinct conv of one UNet forward with HIP events and prints TFLOP/s vs the
157.3 TFLOP/s fp32 MFMA peak.  Usage: python tools/bench_conv.py [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
MATH = sys.argv[3] if len(sys.argv) > 3 else "fp32"     # "bf16": bf16 operands / fp32 accumulate where supported
H0 = int(sys.argv[4]) if len(sys.argv) > 4 else 64       # plane size of level 0 (config 3: 256)
CM = int(sys.argv[5]) if len(sys.argv) > 5 else 1        # channel multiplier (config 3: 2)
LAYERS = [  # (name, Cin, Cout, H, k, groups, count per forward)
    ("L0 32->32 k3", 32, 32, 64, 3, 1, 10), ("L0 32->64 k3", 32, 64, 64, 3, 1, 2), ("L0 64->64 k3", 64, 64, 64, 3, 1, 4),
    ("L0 96->32 k3", 96, 32, 64, 3, 1, 1), ("L0 32->64 k1", 32, 64, 64, 1, 1, 2), ("L0 64->64 k1", 64, 64, 64, 1, 1, 2),
    ("L1 64->64 k3", 64, 64, 32, 3, 1, 2), ("L1 64->128 k3", 64, 128, 32, 3, 1, 2), ("L1 128->128 k3", 128, 128, 32, 3, 1, 4),
    ("L1 192->32 k3", 192, 32, 32, 3, 1, 1), ("L1 32->32 k3", 32, 32, 32, 3, 1, 2),
    ("L2 128->128 k3", 128, 128, 16, 3, 1, 2), ("L2 128->256 k3", 128, 256, 16, 3, 1, 2), ("L2 256->256 k3", 256, 256, 16, 3, 1, 4),
    ("L2 384->64 k3", 384, 64, 16, 3, 1, 1), ("L2 64->64 k3", 64, 64, 16, 3, 1, 2),
    ("L3 256->256 k3", 256, 256, 8, 3, 1, 4), ("L3 768->128 k3", 768, 128, 8, 3, 1, 1), ("L3 128->128 k3", 128, 128, 8, 3, 1, 2),
    ("L3 768->384 g3", 768, 384, 8, 3, 3, 1), ("L3 768->128 k1", 768, 128, 8, 1, 1, 1),
]
tot_t = tot_f = 0.0
for name, ci, co, h, k, g, cnt in LAYERS:
    ci, co, h = ci * CM, co * CM, h * H0 // 64
    x = torch.randn(B, ci, 8, h, h, device="cuda")
    w = torch.randn(co, ci // g, k, k, k, device="cuda") / (ci // g * k ** 3) ** 0.5
    math = "bf16" if MATH == "bf16" and ops.bf16_conv_supported(co, ci, k, g) else "fp32"
    wp = ops.pack_conv_weight_bf16(w, groups=g) if math == "bf16" else ops.pack_conv_weight(w, groups=g)
    y = torch.empty(B, co, 8, h, h, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5
    ops.conv3d([x], wp, co, k, groups=g, in_scale=sc, in_act=True, out=y, math=math)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        ops.conv3d([x], wp, co, k, groups=g, in_scale=sc, in_act=True, out=y, math=math)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    fl = 2.0 * B * co * (ci // g) * k ** 3 * 8 * h * h
    print(f"{name:18s} {math} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3 * 100:5.1f}% of fp32 MFMA peak)  x{cnt}", flush=True)
    tot_t += ms * cnt; tot_f += fl * cnt
print(f"weighted total: {tot_t:.2f} ms for {tot_f / 1e12:.2f} TFLOP -> {tot_f / tot_t / 1e9:.1f} TFLOP/s")
