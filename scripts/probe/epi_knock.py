"""What does the epilogue of an HBM-bound 1x1 launch pay for?  Launch time of ResNet-50's 64 -> 256 @56^2 and 256 -> 1024 @14^2 forward layers
(batch 256) with the stored multiplier and / or the per-pixel maxima switched off (development probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
B = int(os.environ.get("B", "256"))
g = torch.Generator().manual_seed(0)
for (H, Cin, Cout) in [(56, 64, 256), (14, 256, 1024), (28, 128, 512)]:
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to("cuda"))
    w = ops.mark_static((torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to("cuda"))
    csc = (torch.rand(Cout, generator=g) + 0.5).to("cuda")
    out = torch.empty(B, H, H, Cout, device="cuda"); sc = torch.empty_like(out)
    res = {}
    for name, kw in (("y + t + maxima", dict(want_scale=True, track_absmax=True)), ("y + t", dict(want_scale=True, track_absmax=False)),
                     ("y + maxima", dict(want_scale=False, track_absmax=True)), ("y", dict(want_scale=False, track_absmax=False))):
        f = lambda: ops.conv2d_fwd(x, w, ch_scale=csc, relu=True, want_norm=False, out=out, scale_out=sc if kw["want_scale"] else None, **kw)
        best = 1e9
        for rnd in range(3):
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        nbytes = x.numel() * 4 + out.numel() * 4 * (2 if kw["want_scale"] else 1)
        res[name] = (best, nbytes / best / 1e6)
    print(f"{Cin}->{Cout} @{H}^2: " + "   ".join(f"{k}: {v[0]:.0f} us {v[1]:.2f} TB/s" for k, v in res.items()), flush=True)
# the same launches with the tile width forced (option h2_tile: 1 = 128 x 128, 2 = 128 x 256)
from bcos_hip import lib as blib
for (H, Cin, Cout) in [(56, 64, 256), (28, 128, 512)]:
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to("cuda"))
    w = ops.mark_static((torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to("cuda"))
    csc = (torch.rand(Cout, generator=g) + 0.5).to("cuda")
    out = torch.empty(B, H, H, Cout, device="cuda"); sc = torch.empty_like(out)
    for tile in (1, 2):
        blib.set_option("h2_tile", tile)
        for name, kw in (("y + t + maxima", dict(want_scale=True, track_absmax=True)), ("y + t", dict(want_scale=True, track_absmax=False))):
            f = lambda: ops.conv2d_fwd(x, w, ch_scale=csc, relu=True, want_norm=False, out=out, scale_out=sc, **kw)
            best = 1e9
            for rnd in range(3):
                f(); f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): f()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
            print(f"{Cin}->{Cout} @{H}^2 tile 128x{128 * tile}: {name}: {best:.0f} us", flush=True)
    blib.reset_options()
