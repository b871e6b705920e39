#!/usr/bin/env python3
"""Micro-benchmark of the standalone fp32 attention kernel (tmdiff_attn_fwd) and gemm_nt against the fp32 MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for b, h, nq, nk, d in ((32, 8, 1024, 1024, 64), (32, 1, 1024, 1024, 128), (32, 8, 4096, 77, 64), (32, 8, 4096, 4096, 32)):
    q, k, v = (torch.randn(b, n, h * d, device="cuda") for n in (nq, nk, nk))
    ms = timeit(lambda: ops.attention(q, k, v, d ** -0.5, heads=h))
    fl = 4.0 * b * h * nq * nk * d
    print(f"attention B={b} H={h} Nq={nq} Nk={nk} D={d}: {ms:7.3f} ms {fl / ms / 1e9:6.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:4.1f}% of fp32 MFMA peak)", flush=True)
for m, n, k in ((32768, 512, 512), (32768, 1024, 128), (32 * 77, 512, 768)):
    a, w = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda")
    ms = timeit(lambda: ops.gemm_nt(a, w))
    fl = 2.0 * m * n * k
    print(f"gemm_nt M={m} N={n} K={k}: {ms:7.3f} ms {fl / ms / 1e9:6.1f} TFLOP/s ({fl / ms / 1e9 / 157.3 * 100:4.1f}%)", flush=True)
