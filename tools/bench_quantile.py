#!/usr/bin/env python3
"""abs_quantile_clamp timing at the DPM-Solver shapes (per-sample n = C*H*W)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops
for B, n in ((1, 8 * 256 * 256), (1, 8 * 64 * 64), (32, 8 * 64 * 64)):
    x = torch.randn(B, n, device="cuda") * 0.7
    y = x.clone(); s = ops.abs_quantile_clamp_(y, 0.995, 1.0)
    want = torch.quantile(x.abs(), 0.995, dim=1).clamp_min(1.0)
    assert torch.equal(s.cpu(), want.cpu()), (s, want)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.abs_quantile_clamp_(y, 0.995, 1.0)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} n={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
