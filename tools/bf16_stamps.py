#!/usr/bin/env python3
"""Where the cycles of the packed-input bf16 convolution go, per wave (s_memtime sums, a diagnostic build:
tools/build_variant.sh stamps "-DTMDIFF_BF16_STAMPS=1" conv3d_bf16; TMDIFF_HIP_LIB=tools/lib_stamps.so).
Prints, per layer shape and epilogue kind, the median over waves of: prologue (first pieces landed), MFMA phases,
wait for the own pieces to land, barrier, epilogue -- in cycles per tile and as a share of the wave's lifetime."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops
from tmdiff_amd._lib import lib, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [("L0 32->32", 32, 32, 64), ("L0 64->64", 64, 64, 64), ("L1 128->128", 128, 128, 32), ("L2 256->256", 256, 256, 16)]
for name, ci, co, h in LAYERS:
    plane = 8 * h * h
    xp = torch.randn(B, ci // 8, plane, 8, device="cuda").to(torch.bfloat16).view(torch.int16)
    w = ops.pack_conv_weight_bf16(torch.randn(co, ci, 3, 3, 3, device="cuda") / (ci * 27) ** 0.5)
    y = torch.empty(B, co, 8, h, h, device="cuda")
    res = torch.randn(B, co, 8, h, h, device="cuda")
    sc = torch.rand(B, co, device="cuda") + 0.5
    y2 = torch.empty(B, co // 8, plane, 8, device="cuda", dtype=torch.int16)
    for kind in ("y", "y+res+y2"):
        kw = dict(x_bf16_shape=(8, h, h))
        if kind != "y":
            kw.update(residual=res, y2=y2, y2_act=True, y2_scale=sc)
        d = ops.make_conv_desc([xp], w, co, 3, y, **kw)
        stamps = torch.zeros(12 << 17, device="cuda", dtype=torch.int64)
        d.splitk_ws = stamps.data_ptr()
        for _ in range(3):
            check(lib.tmdiff_conv3d_fwd_bf16(C.byref(d), None, ops.stream_ptr()), "conv3d_fwd_bf16")
        torch.cuda.synchronize()
        raw = stamps.view(-1, 12).cpu()
        raw = raw[raw[:, 5] > 0]
        t = raw.double()
        med = t.median(dim=0).values
        names = ["prologue", "mfma", "landing", "barrier", "epilogue"]
        print(f"{name:12s} {kind:9s} waves {len(t):6d}  lifetime {med[5]:8.0f} cyc: " +
              "  ".join(f"{n} {med[i]:7.0f} ({med[i] / med[5] * 100:4.1f}%)" for i, n in enumerate(names)), flush=True)
        # timeline (100 MHz ticks): kernel span, summed wave residency per wave slot, setup before the first stamp
        span = (raw[:, 9].max() - raw[:, 8].min()).item() / 100.0
        resid = (raw[:, 9] - raw[:, 8]).double().sum().item() / 100.0 / 2048
        hw, la = raw[:, 6], raw[:, 7]
        print(f"    span {span:7.1f} us, residency per wave slot {resid:7.1f} us, setup {med[10]:6.0f} cyc, wave life (realtime) "
              f"{(raw[:, 9] - raw[:, 8]).double().median().item() / 100.0:6.2f} us; HW_ID.wave_id counts "
              f"{torch.bincount(hw & 15).tolist()}, simd {torch.bincount((hw >> 4) & 3).tolist()}, LDS_ALLOC values "
              f"{[hex(v) for v in torch.unique(la).tolist()][:8]}", flush=True)
