"""A/B of the split-f16 loops on the ResNet-50 batch-256 layer shapes (development aid): LDS-DMA staging (default) against the
register-staged loop (option h2_loop = 1), same process, interleaved rounds.  Columns: us per launch, TFLOP/s (algorithmic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
from bcos_hip import lib as blib
dev = "cuda"
B = int(os.environ.get("B", "256"))
# (H, Cin, Cout, k, stride)
SHAPES = [(14, 256, 256, 3, 1), (28, 128, 128, 3, 1), (56, 64, 64, 3, 1), (7, 512, 512, 3, 1), (14, 1024, 256, 1, 1), (14, 256, 1024, 1, 1),
          (28, 512, 128, 1, 1), (28, 128, 512, 1, 1), (56, 256, 64, 1, 1), (7, 2048, 512, 1, 1), (7, 512, 2048, 1, 1), (224, 8, 64, 7, 2)]
if os.environ.get("QUICK"):
    SHAPES = SHAPES[:5]
if os.environ.get("PSHAPES") == "1":       # the multi-tap shapes
    SHAPES = [(14, 256, 256, 3, 1), (28, 128, 128, 3, 1), (56, 64, 64, 3, 1), (7, 512, 512, 3, 1), (28, 128, 128, 3, 2), (14, 256, 256, 3, 2)]
if os.environ.get("PSHAPES") == "3":
    SHAPES = [(14, 256, 256, 3, 1), (28, 128, 128, 3, 1), (56, 64, 64, 3, 1), (7, 512, 512, 3, 1)]
if os.environ.get("PSHAPES") == "4":
    SHAPES = [(56, 64, 64, 3, 1), (112, 32, 64, 3, 1)]
if os.environ.get("PSHAPES") == "5":
    SHAPES = [(224, 8, 64, 7, 2)]
if os.environ.get("PSHAPES") == "2":
    SHAPES = [(14, 256, 256, 3, 1)]
g = torch.Generator().manual_seed(0)
rows = []
for (H, Cin, Cout, k, st) in SHAPES:
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to(dev))
    w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(dev))
    pd = k // 2
    f = lambda: ops.conv2d_fwd(x, w, stride=(st, st), padding=(pd, pd), relu=True, want_scale=False, want_norm=False, track_absmax=False)
    Ho = ops.conv_out_size(H, k, st, pd)
    fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin
    res = {}
    for rnd in range(3):
        for mode in ("dma", "regs"):
            # second arm: the register-staged loop, or AB_OPT="name=value" (a bcos_set_option switch, e.g. patch=0)
            if os.environ.get("AB_OPT"):
                k_, v_ = os.environ["AB_OPT"].split("=")
                if mode == "regs": blib.set_option(k_, int(v_))
                else: blib.reset_options()
            else:
                blib.set_option("h2_loop", 1 if mode == "regs" else 0)
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    blib.reset_options()
    d, r = min(res["dma"]), min(res["regs"])
    print(f"fwd {B}x{H}x{H} {Cin:5d}->{Cout:5d} k{k} s{st}   dma {d:8.1f} us {fl / d / 1e6:6.1f} TF   regs {r:8.1f} us {fl / r / 1e6:6.1f} TF   dma/regs {d / r:.3f}", flush=True)
    del x, w
