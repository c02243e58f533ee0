"""Time the ResNet stem input gradient (7x7 stride-2 conv 8 <- 64, batch 256; depth-to-space launch): the input-patch loop in
2-D tiles against the per-tap loop (option patch = 0).  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
from bcos_hip import lib as blib
B = int(os.environ.get("B", "256"))
w = torch.randn(64, 8, 7, 7, device="cuda") / 20
plan = ops.DgradPlan(w, (2, 2), (3, 3), (1, 1))
g = ops.ensure_absmax(torch.randn(B, 112, 112, 64, device="cuda"))
out = torch.empty(B, 224, 224, 8, device="cuda")
res = {}
for rnd in range(3):
    for mode in ("patch", "taps"):
        blib.set_option("patch", 0 if mode == "taps" else 1)
        for _ in range(3): plan.run(g, 224, 224, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.run(g, 224, 224, out=out)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append((e0.elapsed_time(e1) / 10, float(out.abs().sum())))
for mode, v in res.items():
    print(mode, f"{min(t for t, _ in v):.3f} ms", v[0][1])
