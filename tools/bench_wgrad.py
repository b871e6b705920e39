#!/usr/bin/env python3
"""Per-layer weight-gradient micro-benchmark (finetune shapes: local batch 8, 8x64x64, ch 32-256).
Usage: python tools/bench_wgrad.py [B] [reps] [plain]     (plain: x' is a kept tensor, no prologue pass -- the finetune path;
TMDIFF_WGRAD_WINO=0: the direct kernel everywhere).  TFLOP/s in the direct kernel's operation count (27 taps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
PLAIN = "plain" in sys.argv[3:]
LAYERS = [("L0 32->32", 32, 32, 64, 3, 1, 10), ("L0 32->64", 32, 64, 64, 3, 1, 2), ("L0 64->64", 64, 64, 64, 3, 1, 4),
          ("L0 96->32", 96, 32, 64, 3, 1, 1), ("L1 64->128", 64, 128, 32, 3, 1, 2), ("L1 128->128", 128, 128, 32, 3, 1, 4),
          ("L1 192->32", 192, 32, 32, 3, 1, 1), ("L2 128->256", 128, 256, 16, 3, 1, 2), ("L2 256->256", 256, 256, 16, 3, 1, 4),
          ("L3 256->256", 256, 256, 8, 3, 1, 4), ("L3 768->128", 768, 128, 8, 3, 1, 1), ("L3 768->384 g3", 768, 384, 8, 3, 3, 1),
          ("L0 64->64 k1", 64, 64, 64, 1, 1, 2)]
tot_t = tot_f = 0.0
for name, ci, co, h, k, g, cnt in LAYERS:
    x = torch.randn(B, ci, 8, h, h, device="cuda")
    gy = torch.randn(B, co, 8, h, h, device="cuda")
    sc = torch.rand(B, ci, device="cuda") + 0.5
    y = torch.empty(B, co, 8, h, h, device="cuda")
    wp = torch.empty(co * (ci // g) * k ** 3, device="cuda")
    d = ops.make_conv_desc([x], wp, co, k, y, groups=g) if PLAIN else ops.make_conv_desc([x], wp, co, k, y, groups=g, in_scale=sc, in_act=True)
    f = lambda: ops.conv3d_wgrad(d, gy, (co, ci // g, k, k, k))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    fl = 2.0 * B * co * (ci // g) * k ** 3 * 8 * h * h
    wino = ops.wgrad_wino_takes(d)            # F(3,4) along the bands: half the multiply-adds are executed
    ex = fl / (2.0 if wino else 1.0)
    print(f"{name:16s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s direct-order, {ex / ms / 1e9:6.1f} executed ({ex / ms / 1e9 / 157.3 * 100:5.1f}% of fp32 MFMA peak, "
          f"{'winograd' if wino else 'direct'}, passes and reduction included)  x{cnt}", flush=True)
    tot_t += ms * cnt; tot_f += fl * cnt; tot_e = (tot_e if "tot_e" in dir() else 0.0) + ex * cnt
print(f"weighted total: {tot_t:.2f} ms for {tot_f / 1e12:.2f} TFLOP direct-order -> {tot_f / tot_t / 1e9:.1f} TFLOP/s direct-order, {tot_e / tot_t / 1e9:.1f} executed")
