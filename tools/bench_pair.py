import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B = 32
for math in ("bf16", "fp32"):
    for cin, cmid, h in ((32, 64, 64), (64, 64, 64), (32, 32, 64), (128, 128, 32), (256, 256, 16)):
        x = torch.randn(B, cin, 8, h, h, device="cuda")
        pk = ops.pack_conv_weight_bf16 if math == "bf16" else ops.pack_conv_weight
        w20 = pk(torch.randn(cmid, cin, 3, 3, 3, device="cuda") / (cin * 27) ** 0.5)
        w21 = pk(torch.randn(cmid, cmid, 3, 3, 3, device="cuda") / (cmid * 27) ** 0.5)
        sc2 = torch.rand(B, cmid, device="cuda") + 0.5
        res = torch.randn(B, cmid, 8, h, h, device="cuda")
        def sep():
            t1 = ops.conv3d([x], w20, cmid, 3, math=math, in_act=True)
            return ops.conv3d([t1], w21, cmid, 3, math=math, in_scale=sc2, in_act=True, residual=res)
        def fused():
            p = ops.conv3d([x], w20, cmid, 3, math=math, in_act=True, keep_y=False, emit=dict(act=True, scale=sc2))
            return ops.conv3d([p], w21, cmid, 3, math=math, residual=res, x_bf16_shape=(8, h, h) if math == "bf16" else None)
        a, b = t(sep), t(fused)
        c20 = t(lambda: ops.conv3d([x], w20, cmid, 3, math=math, in_act=True))
        c20e = t(lambda: ops.conv3d([x], w20, cmid, 3, math=math, in_act=True, keep_y=False, emit=dict(act=True, scale=sc2)))
        print(f"{math} {cin}->{cmid}->{cmid} @{h}: separate {a:7.1f} us  epilogue-fused {b:7.1f} us  ({(a / b - 1) * 100:+.1f} %)   conv20 alone {c20:7.1f} / with emit {c20e:7.1f}", flush=True)
