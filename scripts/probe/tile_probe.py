"""Launch time of 1 x 1 forward layers of ResNet-50 (batch 256; BN fold, ReLU, stored multiplier, maxima) under the tile-shape switch
(option h2_tile: 0 cost model, 1 = 128 x 128, 2 = 128 x 256; a 256 x 128 configuration was measured with it in round 4) + bit identity of the outputs (development probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib as blib
B = int(os.environ.get("B", "256"))
g = torch.Generator().manual_seed(0)
for (H, Cin, Cout) in [(28, 512, 128), (14, 1024, 256), (7, 2048, 512), (7, 512, 2048), (14, 256, 1024), (56, 256, 64), (28, 128, 512)]:
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to("cuda"))
    w = ops.mark_static((torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to("cuda"))
    csc = (torch.rand(Cout, generator=g) + 0.5).to("cuda")
    ref = None
    line = f"{Cin}->{Cout} @{H}^2:"
    tiles = [t for t in (0, 1, 2) if not (t == 2 and Cout <= 128)]
    best = {t: 1e9 for t in tiles}
    out = torch.empty(B, H, H, Cout, device="cuda"); sc = torch.empty_like(out)
    f = lambda: ops.conv2d_fwd(x, w, ch_scale=csc, relu=True, want_norm=False, out=out, scale_out=sc, want_scale=True, track_absmax=True)
    same = {}
    for rnd in range(6):                      # the configurations interleaved: clocks drift by 10-15 % over a run of launches
        for tile in tiles:
            blib.set_option("h2_tile", tile)
            f(); f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            best[tile] = min(best[tile], e0.elapsed_time(e1) / 5 * 1e3)
            if ref is None: ref = (out.clone(), sc.clone())
            same[tile] = torch.equal(out, ref[0]) and torch.equal(sc, ref[1])
    blib.reset_options()
    print(line + "".join(f"   tile {t}: {best[t]:.0f} us{'' if same[t] else ' DIFFERENT'}" for t in tiles), flush=True)
