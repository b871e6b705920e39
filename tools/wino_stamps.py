#!/usr/bin/env python3
"""Timeline of the Winograd convolution kernel's workgroups (diagnostic build:
    tools/build_variant.sh wstamps "-DTMDIFF_WINO_STAMPS=1" conv3d_wino;  TMDIFF_HIP_LIB=tools/lib_wstamps.so python tools/wino_stamps.py).
Every wave stamps s_memrealtime (100 MHz) at: kernel entry, after the stagger, first chunk landed, end of the chunk loop,
epilogue issued, stores acknowledged.  Printed per layer: medians of the phases, the gap between a workgroup's end and the
start of its successor on the same CU slot, and how much of the time BOTH workgroups of a CU were outside their MFMA phase."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tmdiff_amd import ops
from tmdiff_amd._lib import lib, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
WF = len(sys.argv) > 2 and sys.argv[2] == "wf"      # the in-kernel-transform kernel (build with -DTMDIFF_WF_STAMPS=1 conv3d_wf)
LAYERS = [("L0 32->32", 32, 32, 64), ("L0 32->64", 32, 64, 64), ("L0 64->64", 64, 64, 64), ("L1 128->128", 128, 128, 32),
          ("L2 256->256", 256, 256, 16)]
for name, ci, co, h in LAYERS:
    x = torch.randn(B, ci, 8, h, h, device="cuda")
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") / (ci * 27) ** 0.5
    wp = ops.pack_conv_weight_wino(w, planes=6, mode=2 if WF else 0)
    y = torch.empty(B, co, 8, h, h, device="cuda")
    res = torch.randn(B, co, 8, h, h, device="cuda")
    sc = torch.rand(B, co, device="cuda") + 0.5
    y2 = torch.empty_like(y)
    for kind in ("y", "y+res+y2"):
        kw = {} if kind == "y" else dict(residual=res, y2=y2, y2_act=True, y2_scale=sc)
        d = ops.make_conv_desc([x], wp, co, 3, y, **kw)
        nblk = lib.tmdiff_conv3d_wf_blocks(C.byref(d)) if WF else lib.tmdiff_conv3d_wino_blocks(C.byref(d))
        stamps = torch.zeros(nblk * 4 * 8, device="cuda", dtype=torch.int64)
        d.splitk_ws = stamps.data_ptr()
        if WF:
            for _ in range(3):
                check(lib.tmdiff_conv3d_wf_fwd(C.byref(d), None, ops.stream_ptr()), "conv")
        else:
            ws = ops._workspace(x.device, lib.tmdiff_conv3d_wino_workspace_bytes(C.byref(d)), "wino").data_ptr()
            check(lib.tmdiff_conv3d_wino_fwd_planes(C.byref(d), ws, 1, None, 6, ops.stream_ptr()), "transform")
            for _ in range(3):
                check(lib.tmdiff_conv3d_wino_fwd_planes(C.byref(d), ws, 2, None, 6, ops.stream_ptr()), "conv")
        torch.cuda.synchronize()
        r = stamps.view(nblk, 4, 8).cpu().numpy().astype(np.int64)
        t = r[:, :, :6].astype(np.float64) / 100.0                   # us
        t0 = t[:, :, 0].min()
        wg_start, wg_end = t[:, :, 0].min(1) - t0, t[:, :, 5].max(1) - t0
        mf0, mf1 = t[:, :, 2].max(1) - t0, t[:, :, 3].min(1) - t0    # MFMA phase of the workgroup (all waves inside)
        ids = r[:, 0, 6]
        hw, lds, xcc = (ids >> 32) & 0xffffffff, ids & 0xfff, (ids >> 28) & 0xf
        cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
        slot = (cu << 1) | (lds != 0)
        med = lambda a: float(np.median(a))
        span = wg_end.max()
        ph = {"setup": med(t[:, :, 2] - t[:, :, 1]), "mfma": med(t[:, :, 3] - t[:, :, 2]), "epilogue": med(t[:, :, 4] - t[:, :, 3]),
              "drain": med(t[:, :, 5] - t[:, :, 4]), "life": med(wg_end - wg_start)}
        clk = med(r[:, :, 7] / ((t[:, :, 5] - t[:, :, 0]) * 1e-6 + 1e-12)) / 1e9
        gaps, idle_both, tot = [], 0.0, 0.0
        for c in np.unique(cu):
            m = cu == c
            for s in (0, 1):
                mm = m & ((lds != 0) == bool(s))
                o = np.argsort(wg_start[mm])
                st, en = wg_start[mm][o], wg_end[mm][o]
                gaps += list(st[1:] - en[:-1])
            # time in [0, span] during which NO workgroup of this CU is inside its MFMA phase
            ev = sorted([(a, 1) for a in mf0[m]] + [(b_, -1) for b_ in mf1[m]])
            depth, last, busy = 0, 0.0, 0.0
            for tt, dlt in ev:
                if depth > 0:
                    busy += tt - last
                depth += dlt
                last = tt
            idle_both += span - busy
            tot += span
        print(f"{name:12s} {kind:9s} blocks {nblk:5d} span {span:7.1f} us  clk {clk:4.2f} GHz | per workgroup (median, us): " +
              "  ".join(f"{k} {v:6.1f}" for k, v in ph.items()) +
              f" | slot hand-over gap median {med(gaps) if gaps else 0:5.1f} us (p90 {np.percentile(gaps, 90) if gaps else 0:5.1f}) | "
              f"no workgroup of the CU in its MFMA phase: {idle_both / tot * 100:4.1f}% of the span", flush=True)
