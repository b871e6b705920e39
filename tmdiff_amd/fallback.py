"""Kernels kept for shapes NO BASELINE configuration reaches (tests/test_host_logic.py::test_routing_table_of_the_baseline_configs
asserts that): the round-2 Winograd path -- an input-transform PASS (wino_input_kernel: 4 B read + 6.6 / 8.8 B written per input
element) followed by conv3d_wino_kernel, F(4,3) or F(2,3) along the band axis (csrc/conv3d_wino.hip).  Since round 3 every
8- and 4-band tensor takes conv3d_wf (transform inside the kernel) or, on small grids, the direct kernels; what is left for this
path are EVEN BAND COUNTS OTHER THAN 4 AND 8 (6, 12, 16 ... bands) on grids of at least ops.config.wino_min_blocks workgroups.
The reference's sensors have 4 or 8 bands (config/general*.json), so this module is generality, not the product path; its
tests carry the `fallback` marker (and `gpu`: they run in the same GPU session).

routing.conv3_family returns "wino4" / "wino2" for such shapes and ops.conv3d_auto dispatches here."""
import ctypes as C

import torch

from . import ops
from ._lib import check, lib


def wino_planes(n_bands):
    """Planes of the Winograd transform the library uses for a tensor of n_bands bands: 6 (F(4,3)), 4 (F(2,3)), 0 (odd)."""
    return lib.tmdiff_conv3d_wino_planes(int(n_bands))


def conv3d_wino(segs, w_packed, cout, planes=None, emit=None, keep_y=True, groups=1, xp_out=None, **kw):
    """ops.conv3d(segs, ...) (fp32, 3x3x3, groups 1 / 3) through the transform-pass Winograd kernels: same keyword arguments
    and return convention.  w_packed = ops.pack_conv_weight_wino(w, groups, mode, planes) with planes = 6 (F(4,3), N % 4 == 0)
    or 4 (F(2,3)); planes=None: wino_planes(N).  drop = (seed, p): in-kernel dropout of the prologue output; xp_out: a
    [B, Cin, N, H, W] tensor that receives that output (finetune path)."""
    b, _, n, h, w = segs[0].shape
    dev = segs[0].device
    planes = planes or wino_planes(n)
    if planes not in (4, 6) or w % 4:
        raise ValueError(f"conv3d_wino: {n} bands x {w} columns not supported (even band count, W % 4 == 0)")
    y = torch.empty(b, cout, n, h, w, device=dev, dtype=torch.float32) if keep_y else None
    y2 = None
    if emit is not None:
        y2 = torch.empty(b, cout, n, h, w, device=dev, dtype=torch.float32)
        kw = dict(kw, y2_act=emit.get("act", False), y2_shift=emit.get("shift"), y2_scale=emit.get("scale"),
                  y2_shift_stride=emit.get("shift_stride", 0), y2_scale_stride=emit.get("scale_stride", 0))
    elif y is None:
        raise ValueError("conv3d_wino: keep_y=False needs emit=")
    d = ops.make_conv_desc(segs, w_packed, cout, 3, y, y2=y2, groups=groups, **kw)
    if not lib.tmdiff_conv3d_wino_supported(C.byref(d)):
        raise ValueError("conv3d_wino: shape not supported")
    ws = ops._workspace(dev, lib.tmdiff_conv3d_wino_workspace_bytes(C.byref(d)), "wino").data_ptr()
    ret = y if y2 is None else ((y, y2) if y is not None else y2)
    if xp_out is not None and not (xp_out.is_cuda and xp_out.is_contiguous() and xp_out.numel() == b * d.Cin * n * h * w):
        raise ValueError("conv3d_wino: xp_out must be a contiguous fp32 [B, Cin, N, H, W] tensor")
    mo = planes - 2                  # bands per tile
    flops = 2.0 * b * cout * (d.Cin // groups) * (9.0 * planes / mo) * n * h * w      # EXECUTED: 9 * planes per tile of mo bands
    ops._count(f"conv3d_wino{mo}_fwd", flops)
    if ops.TIMER is None or xp_out is not None:
        check(lib.tmdiff_conv3d_wino_fwd_planes(C.byref(d), ws, 0, xp_out.data_ptr() if xp_out is not None else None, planes,
                                                ops.stream_ptr()), "conv3d_wino_fwd")
        return ret
    # timed: the input-transform pass (an HBM pass, recorded under ksize 0 with its bytes) and the convolution kernel apart
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    check(lib.tmdiff_conv3d_wino_fwd_planes(C.byref(d), ws, 1, None, planes, ops.stream_ptr()), "conv3d_wino_fwd (input transform)")
    ev[1].record()
    check(lib.tmdiff_conv3d_wino_fwd_planes(C.byref(d), ws, 2, None, planes, ops.stream_ptr()), "conv3d_wino_fwd")
    ev[2].record()
    ops.TIMER.records.append((ev[0], ev[1], (4.0 + 4.0 * planes / mo) * b * d.Cin * n * h * w, 0, "wino_input", ops._tag(d)))
    ops.TIMER.records.append((ev[1], ev[2], flops, 3, f"conv3d_wino{mo}_fwd", ops._tag(d)))
    return ret
