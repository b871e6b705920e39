import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tmdiff_amd import ops
torch.manual_seed(77 + 32 + 8)
B, cin, cout, N, H, W = 1, 6, 32, 8, 12, 20
x = torch.randn(B, cin, N, H, W).cuda()
w = (torch.randn(cout, cin, 3, 3, 3) / (cin * 27) ** 0.5).cuda(); bias = torch.randn(cout).cuda()
sh, sc = (torch.randn(B, cin) * 0.3).cuda(), (torch.rand(B, cin) + 0.5).cuda()
res = torch.randn(B, cout, N, H, W).cuda()
sh2, sc2 = (torch.randn(B, cout) * 0.3).cuda(), (torch.rand(B, cout) + 0.5).cuda()
wp = ops.pack_conv_weight_wino(w, groups=1, mode=2, planes=6)
kw = dict(bias=bias, in_shift=sh, in_scale=sc, in_act=True, residual=res, out_scale=0.7071)
for it in range(3):
    y, y2 = ops.conv3d_wf([x], wp, cout, emit=dict(act=True, shift=sh2, scale=sc2), **kw)
    only = ops.conv3d_wf([x], wp, cout, emit=dict(act=True, shift=sh2, scale=sc2), keep_y=False, **kw)
    torch.cuda.synchronize()
    d = (only != y2)
    print(it, "equal", torch.equal(only, y2), "ndiff", int(d.sum()), "nan", int(torch.isnan(only).sum()), int(torch.isnan(y2).sum()))
    if d.any():
        idx = d.nonzero()[:10]
        print(idx.tolist())
        print([ (float(only[tuple(i)]), float(y2[tuple(i)])) for i in idx[:5]])
