"""Time of the weight-gradient launch on the ResNet-50 training shapes (batch 64), for knock-out builds (BCOS_HIP_LIB=...).
usage: [BCOS_HIP_LIB=lib/variants/wkoN.so] python scripts/probe/wgrad_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd"))
import torch
from bcos_hip import ops, lib
lib.load()
N = 64
if os.environ.get("WGRAD_VIT"):       # the token linears of a ViT-Ti training step (batch 64: 12 608 rows): as 1 x 1 layers over [1, rows]
    N = 1
shapes = [  # (H, Cin, Cout, k, stride)
    (56, 64, 64, 1, 1), (56, 64, 64, 3, 1), (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (56, 256, 128, 1, 1), (28, 128, 128, 3, 1),
    (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (14, 256, 256, 3, 1), (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (7, 512, 512, 3, 1),
    (7, 512, 2048, 1, 1), (7, 2048, 512, 1, 1)]
if os.environ.get("WGRAD_VIT"):
    shapes = [(-12608, 192, 576, 1, 1), (-12608, 192, 192, 1, 1), (-12608, 192, 768, 1, 1), (-12608, 768, 192, 1, 1), (-12608, 192, 1000, 1, 1)]
tot = 0.0
for (H, Cin, Cout, k, s) in shapes:
    if H < 0:
        x = torch.randn(1, 1, -H, Cin, device="cuda"); g = torch.randn(1, 1, -H, Cout, device="cuda"); P = 1
    else:
        x = torch.randn(N, H, H, Cin, device="cuda")
        P = H // s
        g = torch.randn(N, P, P, Cout, device="cuda")
    pad = k // 2
    f = lambda: ops.conv2d_wgrad(g, x, Cin, Cout, (k, k), (s, s), (pad, pad), (1, 1))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * (N * P * P if H > 0 else -H) * Cout * Cin * k * k
    tot += us
    print(f"{H:3d}^2 {Cin:5d}->{Cout:5d} k{k}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
print(f"sum {tot:.1f} us")
