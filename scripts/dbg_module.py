import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn as nn
from bcos_hip import synth
from oracle import bcos_oracle as O
from bcos.modules.bcosifyconv2d import BcosifyConv2d
from bcos.modules.norms.uncentered_norms import BatchNormUncentered2d
from bcos.common import explanation_mode
dev = "cuda"
def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
torch.manual_seed(0)
cfg = synth.resnet_model_config("resnet18")
conv = BcosifyConv2d(8, 16, 3, 1, 1, b=2).to(dev)
bn = nn.BatchNorm2d(16); bn.running_var.uniform_(0.5, 1.5); bn.weight.data.uniform_(0.5, 1.5)
bnu = BatchNormUncentered2d.from_standard_module(bn, cfg); bnu.bias = None; bnu = bnu.to(dev).eval()
x = torch.randn(2, 8, 10, 10)
def ref(xx, upto):
    y = O.bcos_conv2d(xx, conv.linear.weight.detach().cpu(), stride=1, padding=1, detach=True)
    if upto >= 1: y = O.bn_uncentered_eval(y, bnu.running_var.cpu(), bnu.weight.detach().cpu(), None, bnu.eps)
    if upto >= 2: y = torch.relu(y)
    if upto >= 3: y = torch.nn.functional.avg_pool2d(y, 3, 2, 1)
    return y
relu = nn.ReLU(inplace=True); pool = nn.AvgPool2d(3, 2, 1)
for upto in range(4):
    xg = x.clone().to(dev).requires_grad_(True)
    conv.set_explanation_mode(True)
    y = conv(xg)
    if upto >= 1: y = bnu(y)
    if upto >= 2: y = relu(y)
    if upto >= 3: y = pool(y)
    gy = torch.randn(y.shape)
    (gx,) = torch.autograd.grad(y, xg, gy.to(dev))
    xr = x.clone().requires_grad_(True)
    yr = ref(xr, upto)
    (gxr,) = torch.autograd.grad(yr, xr, gy)
    print(upto, "y", rel(y, yr), "gx", rel(gx, gxr), y.stride(), gx.stride())
print("---- stagewise")
xg = x.clone().to(dev).requires_grad_(True)
y0 = conv(xg); y1 = bnu(y0); y2 = torch.relu(y1); y3 = pool(y2)
gy = torch.randn(y3.shape)
g2, g1, g0, gx = torch.autograd.grad(y3, [y2, y1, y0, xg], gy.to(dev))
xr = x.clone().requires_grad_(True)
r0 = O.bcos_conv2d(xr, conv.linear.weight.detach().cpu(), stride=1, padding=1, detach=True)
r1 = O.bn_uncentered_eval(r0, bnu.running_var.cpu(), bnu.weight.detach().cpu(), None, bnu.eps)
r2 = torch.relu(r1); r3 = torch.nn.functional.avg_pool2d(r2, 3, 2, 1)
h2, h1, h0, hx = torch.autograd.grad(r3, [r2, r1, r0, xr], gy)
for n, a, b in (("g2", g2, h2), ("g1", g1, h1), ("g0", g0, h0), ("gx", gx, hx)):
    print(n, rel(a, b), a.stride(), a.is_contiguous())
