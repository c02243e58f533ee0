"""Probe (round 5): FORCED heterogeneous co-residency.  A matrix-bound launch X requests 84 KB of LDS (option lds_min_kb), so that two of
its workgroups never share a CU, while a bandwidth-bound launch Y (<= 76 KB) can sit beside it: every CU then runs one X and one Y
workgroup.  Compared with the same launches run concurrently without the constraint, and one after the other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib as blib
B = int(os.environ.get("B", "256"))
g = torch.Generator().manual_seed(0)
dev = "cuda"


def layer(H, Cin, Cout, k, heavy_epi, lds_kb):
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to(dev))
    w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(dev))
    csc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    out = torch.empty(B, H, H, Cout, device=dev)
    sc = torch.empty_like(out) if heavy_epi else None
    pd = k // 2

    def f(force=False):
        if force and lds_kb:
            blib.set_option("lds_min_kb", lds_kb)
        ops.conv2d_fwd(x, w, padding=(pd, pd), ch_scale=csc, relu=True, out=out, scale_out=sc, want_scale=heavy_epi, track_absmax=True)
        if force and lds_kb:
            blib.set_option("lds_min_kb", 0)
    return f


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


XS = {"3x3 256->256 @14": (14, 256, 256, 3, False, 84), "1x1 1024->256 @14": (14, 1024, 256, 1, False, 84), "3x3 128->128 @28": (28, 128, 128, 3, False, 84)}
YS = {"1x1 64->256 @56 (+t)": (56, 64, 256, 1, True, 0), "1x1 256->64 @56": (56, 256, 64, 1, False, 0), "1x1 256->1024 @14 (+t)": (14, 256, 1024, 1, True, 0)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
fx = {k: layer(*v) for k, v in XS.items()}
fy = {k: layer(*v) for k, v in YS.items()}
for f in list(fx.values()) + list(fy.values()):
    f(); f(True)
torch.cuda.synchronize()


def corun(X, Y, nx, ny, force):
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    s1.wait_stream(cur); s2.wait_stream(cur)
    ix = iy = 0
    while ix < nx or iy < ny:
        if ix < nx and (iy >= ny or ix * ny <= iy * nx):
            with torch.cuda.stream(s1):
                X(force)
            ix += 1
        else:
            with torch.cuda.stream(s2):
                Y()
            iy += 1
    cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


print(f"batch {B}; us per launch alone (X also with the 84 KB request = one X workgroup per CU), then X || Y on two streams")
for kx, X in fx.items():
    for ky, Y in fy.items():
        best = None
        for rnd in range(3):
            tx = timed(X, 20); tx1 = timed(lambda: X(True), 20); ty = timed(Y, 20)
            nx = 40
            ny = max(1, round(nx * tx / ty))
            serial = nx * tx + ny * ty
            free = corun(X, Y, nx, ny, False)
            forced = corun(X, Y, nx, ny, True)
            r = (forced / serial, free / serial, tx, tx1, ty, nx, ny, serial)
            best = r if best is None or r[0] < best[0] else best
        print(f"X {kx:18s} {best[2]:7.1f} us (1/CU: {best[3]:7.1f}) | Y {ky:22s} {best[4]:7.1f} us | {best[5]}x || {best[6]}y: serial {best[7]:8.0f} us, "
              f"concurrent {best[1]:.3f}, concurrent with X exclusive {best[0]:.3f}", flush=True)
