"""Time the ResNet stem input gradient (7x7 stride-2 conv 8 <- 64, batch 256): fused parity-class launch vs one launch
per class (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
B = int(os.environ.get("B", "256"))
w = torch.randn(64, 8, 7, 7, device="cuda") / 20
plan = ops.DgradPlan(w, (2, 2), (3, 3), (1, 1))
g = torch.randn(B, 112, 112, 64, device="cuda")
out = torch.empty(B, 224, 224, 8, device="cuda")
for mode in ("group", "separate"):
    ops._NO_GROUP = mode == "separate"
    for _ in range(3): plan.run(g, 224, 224, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.run(g, 224, 224, out=out)
    e1.record(); torch.cuda.synchronize()
    print(mode, f"{e0.elapsed_time(e1) / 10:.3f} ms", float(out.abs().sum()))
