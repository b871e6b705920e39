import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tmdiff_amd import ops, _lib
torch.manual_seed(0)
B, cin, cout, N, H, W = 2, 32, 64, 8, 16, 16
x = torch.randn(B, cin, N, H, W); g = torch.randn(B, cout, N, H, W)
wd = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
F.conv3d(x.double(), wd, padding=1).backward(g.double())
want = wd.grad
xd, gd = x.cuda(), g.cuda()
d = ops.make_conv_desc([xd], 0, cout, 3, gd)
T = N // 4; Q = B * T; Hb = (H + 7) // 8 * 8; Wb = (W + 15) // 16 * 16; CiP = (cin + 31) // 32 * 32; CoP = (cout + 31) // 32 * 32
BT = torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=torch.float64)
G4 = torch.tensor([[1/4,0,0,0],[-1/6,-1/6,-1/6,-1/6],[-1/6,1/6,-1/6,1/6],[1/24,1/12,1/6,1/3],[1/24,-1/12,1/6,-1/3],[0,0,0,1]], dtype=torch.float64)
xp = F.pad(x.double(), (0, 0, 0, 0, 1, 1))       # bands
xh_want = torch.zeros(6, Q, Hb + 2, Wb + 2, CiP, dtype=torch.float64)
gh_want = torch.zeros(6, Q, Hb, Wb, CoP, dtype=torch.float64)
for b in range(B):
    for t in range(T):
        xt = xp[b, :, 4 * t:4 * t + 6]                       # [c][6][H][W]
        v = torch.einsum("kj,cjhw->khwc", BT, xt)
        xh_want[:, b * T + t, 1:H + 1, 1:W + 1, :cin] = v
        gt = g.double()[b, :, 4 * t:4 * t + 4]
        gh_want[:, b * T + t, :H, :W, :cout] = torch.einsum("kj,cjhw->khwc", G4, gt)
for call in range(3):
    dw = ops.conv3d_wgrad(d, gd, tuple(wd.shape))
    torch.cuda.synchronize()
    ws = ops._WS[(0, ops.stream_ptr(), "wgrad")]
    fl = ws.view(torch.float32)
    nx, ng = xh_want.numel(), gh_want.numel()
    xh = fl[:nx].view_as(xh_want).cpu().double(); gh = fl[nx:nx + ng].view_as(gh_want).cpu().double()
    print("call", call, "x^ err", float((xh - xh_want).abs().max()), "g^ err", float((gh - gh_want).abs().max()))
    e = (dw.cpu().double() - want).abs()
    print("  dw err max", float(e.max()), "scale", float(want.abs().max()))
    for co_t in range(cout // 32):
        for ci_t in range(cin // 32):
            et = e[co_t * 32:(co_t + 1) * 32, ci_t * 32:(ci_t + 1) * 32]
            print("   tile", co_t, ci_t, "max", float(et.max()), "per dn", [float(et[:, :, i].max()) for i in range(3)], "per tap", [round(float(et[:, :, :, i // 3, i % 3].max()), 3) for i in range(9)])
    # partials: m_k from ws vs expected
    npart = fl[nx + ng:]
