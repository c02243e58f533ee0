"""Where does the fc weight gradient of the shallow Bottleneck fixture lose 3.7e-4?  wgrad alone, then the module path of one layer."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "b-cosification_amd"))
import torch, torch.nn.functional as F
from bcos_hip import ops
DEV = "cuda"
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
g = torch.Generator().manual_seed(1)
for (N, Cin, H, Cout) in [(4, 2048, 2, 1000), (4, 2048, 2, 1024), (4, 512, 2, 1000), (64, 2048, 2, 1000), (4, 2048, 8, 1000)]:
    x = torch.randn(N, Cin, H, H, generator=g)
    gl = torch.randn(N, Cout, H, H, generator=g) * 1e-3
    w = torch.zeros(Cout, Cin, 1, 1, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), w), w, gl.double())
    glp = gl.permute(0, 2, 3, 1).contiguous()
    pad = (-Cout) % 4
    if pad: glp = F.pad(glp, (0, pad))
    gw = ops.conv2d_wgrad(glp.to(DEV), x.permute(0, 2, 3, 1).contiguous().to(DEV), Cin, Cout, (1, 1), (1, 1), (0, 0), (1, 1))
    print("wgrad", (N, Cin, H, Cout), rel(gw.permute(0, 3, 1, 2), ref), float(gw.norm()) / float(ref.norm()) - 1)
# the module path of one B-cos 1x1 layer + GAP in train mode against fp64 autograd of the same formula
from bcos.modules.bcosifyconv2d import BcosifyConv2d
torch.manual_seed(0)
for (N, Cin, H, Cout) in [(4, 2048, 2, 1000), (4, 256, 4, 1000), (4, 2048, 2, 1024)]:
    m = BcosifyConv2d(Cin, Cout, 1, 1, 0, b=2).to(DEV)
    m.train()
    x = (torch.rand(N, Cin, H, H, generator=g) * 0.5).to(DEV).requires_grad_(True)
    t = torch.randn(N, Cout, generator=g).to(DEV)
    y = m(x).mean((2, 3))
    loss = (y * t).sum()
    gx, gw = torch.autograd.grad(loss, [x, m.linear.weight])
    wd = m.linear.weight.detach().double().cpu().requires_grad_(True)
    xd = x.detach().double().cpu().requires_grad_(True)
    lin = F.conv2d(xd, wd)
    norm = (xd.pow(2).sum(1, keepdim=True) + 1e-6).sqrt()
    out = lin * (lin.abs() / norm)
    l2 = (out.mean((2, 3)) * t.double().cpu()).sum()
    gxr, gwr = torch.autograd.grad(l2, [xd, wd])
    print("layer", (N, Cin, H, Cout), "gx", rel(gx, gxr), "gw", rel(gw, gwr), float(gw.norm()) / float(gwr.norm()) - 1)
