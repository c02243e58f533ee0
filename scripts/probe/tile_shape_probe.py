"""Per-shape timing of the split-f16 1 x 1 contraction under bcos_option h2_tile (0 = the dispatch's choice, 1 = 128 x 128, 2 = 128 x 256,
3 = 256 x 128: round-6 development build only), forward epilogue (B-cos scale + ReLU + stored multiplier), launches timed back to back on one stream.
usage: python scripts/probe/tile_shape_probe.py [tiles, e.g. 0,3]   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "b-cosification_amd"))
import torch
from bcos_hip import lib, ops
lib.load()
dev = torch.device("cuda", 0)
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]
shapes = [(802816, 64, 256), (802816, 256, 128), (200704, 128, 512), (200704, 512, 128), (200704, 256, 512), (200704, 512, 256),
          (50176, 256, 1024), (50176, 1024, 256), (50176, 512, 1024), (50176, 1024, 512), (12544, 512, 2048), (12544, 2048, 512),
          (12544, 1024, 2048), (12544, 2048, 1024)]
torch.manual_seed(0)
print("       M      K      N   " + "   ".join(f"tile{t:d} us" for t in tiles))
for (M, K, N) in shapes:
    a = torch.randn(1, 1, M, K, device=dev)
    ops.ensure_absmax(a)
    w = ops.mark_static(torch.randn(N, 1, 1, K, device=dev) * 0.05)
    out, sc = torch.empty(1, 1, M, N, device=dev), torch.empty(1, 1, M, N, device=dev)
    g = ops.fwd_geom(1, 1, M, K, N, 1, 1, 1, 1, 0, 0)
    res = []
    for t in tiles:
        lib.set_option("h2_tile", t)
        run = lambda: ops.tapconv(a, w, g, out=out, bcos_mode=lib.BCOS_CONV_EPS, b=2.0, relu=True, scale_out=sc)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100.0)
    lib.set_option("h2_tile", 0)
    print(f"{M:8d} {K:6d} {N:6d}   " + "   ".join(f"{r:8.1f}" for r in res))
