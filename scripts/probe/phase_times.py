"""Where does a tile of the LDS-DMA contraction loop (tile_body_d) spend its time?  Needs a development build with -DBCOS_PHASE_TIMING=1
(scripts/build_d_variants.sh "phase:-DBCOS_PHASE_TIMING=1", BCOS_HIP_LIB=.../lib/variants/phase.so): every workgroup adds the clock of
its prologue (addresses, operand scales, first DMA), K loop and epilogue to device counters; this script runs ResNet-50's 1x1 layer
shapes and prints the split per launch (wall clock: 100 MHz ticks -> us), next to the launch's duration and the tiles per slot."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib as blib
lib = blib.load()
fetch = [getattr(lib, f"bcos_debug_phase_p{k}") for k in range(12) if hasattr(lib, f"bcos_debug_phase_p{k}")]
assert fetch, "not a BCOS_PHASE_TIMING build"
B = int(os.environ.get("B", "256"))
dev = "cuda"
g = torch.Generator().manual_seed(0)


def phases():
    tot = [0] * 8
    tot[4] = 1 << 63
    for f in fetch:
        buf = (C.c_ulonglong * 8)()
        f(buf)
        if buf[3]:
            for i in (0, 1, 2, 3, 6, 7):
                tot[i] += buf[i]
            tot[4] = min(tot[4], buf[4]); tot[5] = max(tot[5], buf[5])
    return tot


shapes = [(56, 64, 256, "fwd 64->256 @56^2"), (56, 256, 64, "fwd 256->64 @56^2"), (28, 128, 512, "fwd 128->512 @28^2"), (28, 512, 128, "fwd 512->128 @28^2"),
          (14, 256, 1024, "fwd 256->1024 @14^2"), (14, 1024, 256, "fwd 1024->256 @14^2"), (7, 512, 2048, "fwd 512->2048 @7^2"), (7, 2048, 512, "fwd 2048->512 @7^2")]
print(f"{'launch':24s} {'tiles':>6s} {'launch us':>10s} | per workgroup: {'prologue':>9s} {'K loop':>8s} {'epilogue':>9s} {'sum':>7s} us | epilogue share")
for (H, Cin, Cout, name) in shapes:
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to(dev))
    w = ops.mark_static((torch.randn(Cout, 1, 1, Cin, generator=g) / Cin ** 0.5).to(dev))
    csc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    f = lambda: ops.conv2d_fwd(x, w, ch_scale=csc, relu=True, want_scale=True, want_norm=False, track_absmax=True)
    f(); f(); phases()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    t = phases()
    n = max(t[3], 1)
    pro, loop, epi = (t[0] / n / 100.0, t[1] / n / 100.0, t[2] / n / 100.0)
    print(f"{name:24s} {t[3]:6d} {e0.elapsed_time(e1) * 1e3:10.1f} | {pro:23.2f} {loop:8.2f} {epi:9.2f} {pro + loop + epi:7.2f}    | {epi / (pro + loop + epi):.2f}   (span {(t[5] - t[4]) / 100.0:.1f} us; epilogue part 0: rows + transpose {t[6] / n / 100.0:.2f}, math + stores (issue) {t[7] / n / 100.0:.2f})", flush=True)
    del x, w
