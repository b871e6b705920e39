#!/usr/bin/env python3
"""The composed `Conv_0 + halved LL band` convolution (csrc/conv3d_ll.hip) at the three main-branch shapes of the bench
workload, with Winograd along the bands on top (conv3d_wf's composed-LL mode on the space-to-depth input) and beside the pair
they replace (3x3x3 convolution + LL-only DWT).  Usage: python tools/bench_conv_ll.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


for name, c, h in (("down1.down.Conv_0  64->64  @64x64", 64, 64), ("down2.down.Conv_0 128->128 @32x32", 128, 32),
                   ("down3.down.Conv_0 256->256 @16x16", 256, 16)):
    x = torch.randn(B, c, 8, h, h, device="cuda")
    w = torch.randn(c, c, 3, 3, 3, device="cuda") / (c * 27) ** 0.5
    bias = torch.randn(c, device="cuda")
    sc = torch.rand(B, c, device="cuda") + 0.5
    wl, wp = ops.pack_conv_weight_ll(w, 0.5), ops.pack_conv_weight(w)
    a = t(lambda: ops.conv3d_ll(x, wl, c, 0.5, bias=bias, emit=dict(act=True, scale=sc), keep_y=False))
    b = t(lambda: ops.haar_dwt2d(ops.conv3d([x], wp, c, 3, bias=bias), want_high=False, ll_scale=0.5,
                                 ll_prologue=dict(act=True, scale=sc)))
    # the same with Winograd F(4,3) along the bands on top (conv3d_wf's composed-LL mode), on the space-to-depth form of x
    xs = x.view(B, c, 8, h // 2, 2, h // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(B, 4 * c, 8, h // 2, h // 2).contiguous()
    wq = ops.pack_conv_weight_wfll(w, 0.5)
    e = t(lambda: ops.conv3d_wf_ll(xs, wq, c, 0.5, bias=bias, emit=dict(act=True, scale=sc), keep_y=False))
    ex = 2.0 * B * c * c * 48 * 8 * (h // 2) ** 2
    print(f"{name}: composed + Winograd {e:6.3f} ms ({ex / 2 / e / 1e9:6.1f} TFLOP/s executed) | composed {a:6.3f} ms ({ex / a / 1e9:6.1f} "
          f"TFLOP/s executed, {ex * 2.25 / a / 1e9:6.1f} in the reference's order) | convolution + DWT {b:6.3f} ms ({ex * 2.25 / b / 1e9:6.1f} TFLOP/s)",
          flush=True)
