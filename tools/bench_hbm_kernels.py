#!/usr/bin/env python3
"""Achieved HBM bandwidth of the bandwidth-bound kernels at the config-2 shapes (B=32), algorithmic bytes / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

B = 32
def report(name, ms, nbytes):
    print(f"{name:44s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.1f} GB/s  ({nbytes / ms / 1e6 / 8000 * 100:4.1f}% of 8 TB/s spec, {nbytes / ms / 1e6 / 6290 * 100:4.1f}% of 6.29 TB/s copy)", flush=True)

for c, h in ((64, 64), (128, 32), (256, 16)):
    x = torch.randn(B, c, 8, h, h, device="cuda"); n = x.numel() * 4
    report(f"dwt2d 4 bands   [{B},{c},8,{h},{h}]", timeit(lambda: ops.haar_dwt2d(x, True, 0.5)), 2 * n)
    report(f"dwt2d LL only   [{B},{c},8,{h},{h}]", timeit(lambda: ops.haar_dwt2d(x, False, 0.5)), n + n // 4)
for c, h in ((128, 8), (64, 16), (32, 32)):
    a, b2 = torch.randn(B, c, 8, h, h, device="cuda"), torch.randn(B, c, 8, h, h, device="cuda")
    bands = torch.randn(B, 3 * c, 8, h, h, device="cuda"); n = a.numel() * 4
    report(f"idwt2d pair     [{B},{c},8,{h},{h}] -> x2", timeit(lambda: ops.haar_idwt2d([a, b2], None, None, None, 2.0, stacked_bands=bands)), 5 * n + 8 * n)
x = torch.randn(B, 8, 64, 64, device="cuda"); e, nz, ms_ = (torch.randn_like(x) for _ in range(3)); img = torch.empty_like(x)
report("ddpm_step (+img) [32,8,64,64]", timeit(lambda: ops.ddpm_step(x, e, nz, 1.1, 0.4, 0.3, 0.7, 0.1, ms=ms_, img_out=img)), 6 * x.numel() * 4)
w, bias = torch.randn(32, device="cuda"), torch.randn(32, device="cuda")
pan = torch.randn(B, 1, 64, 64, device="cuda")
report("stem (pan-ms -> 32 ch) [32,8,64,64]", timeit(lambda: ops.stem(w, bias, 32, pan=pan, ms=x)), (32 + 1) * x.numel() * 4)
x5 = torch.randn(B, 32, 8, 64, 64, device="cuda"); sc = torch.rand(B, 32, device="cuda")
report("head (32 ch -> 1) [32,32,8,64,64]", timeit(lambda: ops.head(x5, w, sc)), (32 + 1) * x.numel() * 4)
