"""Does starting the second resident workgroup of a CU LATE (first round of tiles only) put the two workgroups of a CU in antiphase -- one in
its K loop while the other drains its epilogue -- for launches of a few rounds of tiles?  Per-shape timing of the 1 x 1 contraction with
BCOS_STAGGER = n (development build: workgroups 256..511 of a launch sleep n x 8128 clocks before they start).
usage: python scripts/probe/stagger_probe.py   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "b-cosification_amd"))
import torch
from bcos_hip import lib, ops
lib.load()
dev = torch.device("cuda", 0)
staggers = [0, 1, 2, 3, 4, 6, 8, 0]
shapes = [(100864, 192, 768), (100864, 192, 576), (100864, 768, 192), (100864, 192, 192), (33621, 192, 768), (33621, 192, 576),
          (200704, 128, 512), (200704, 512, 128), (50176, 256, 1024), (50176, 1024, 256), (12544, 512, 2048), (12544, 2048, 512),
          (802816, 64, 256)]
torch.manual_seed(0)
print("       M      K      N   " + "  ".join(f"s={t:<5d}" for t in staggers))
for (M, K, N) in shapes:
    a = torch.randn(1, 1, M, K, device=dev)
    ops.ensure_absmax(a)
    w = ops.mark_static(torch.randn(N, 1, 1, K, device=dev) * 0.05)
    out, sc = torch.empty(1, 1, M, N, device=dev), torch.empty(1, 1, M, N, device=dev)
    g = ops.fwd_geom(1, 1, M, K, N, 1, 1, 1, 1, 0, 0)
    res = []
    for t in staggers:
        os.environ["BCOS_STAGGER"] = str(t)
        run = lambda: ops.tapconv(a, w, g, out=out, bcos_mode=lib.BCOS_CONV_EPS, b=2.0, relu=True, scale_out=sc)
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 50.0)
    os.environ["BCOS_STAGGER"] = "0"
    print(f"{M:8d} {K:6d} {N:6d}   " + "  ".join(f"{r:7.1f}" for r in res))
