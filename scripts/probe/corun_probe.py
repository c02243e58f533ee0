"""Probe (round 5): what do a matrix-bound and a bandwidth-bound launch cost when they run CONCURRENTLY on two HIP streams, against
one after the other?  The ResNet-50 step is 12.6 ms of matrix-bound + 11.6 ms of bandwidth-bound launches; two sub-batch streams only
overlap 8 % of that.  X = 3x3 256->256 @14^2 (and 1x1 1024->256 @14^2), Y = 1x1 64->256 @56^2 with stored multiplier + maxima (and
256->64 @56^2), sub-batch size 128.  Second part: streaming bandwidth of a float4 copy against the buffer size (does a
producer -> consumer pair that fits the 256 MiB Infinity Cache run faster than HBM?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
B = int(os.environ.get("B", "128"))
g = torch.Generator().manual_seed(0)
dev = "cuda"


def layer(H, Cin, Cout, k, heavy_epi):
    x = ops.ensure_absmax(torch.randn(B, H, H, Cin, generator=g).to(dev))
    w = ops.mark_static((torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(dev))
    csc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    out = torch.empty(B, H, H, Cout, device=dev)
    sc = torch.empty_like(out) if heavy_epi else None
    pd = k // 2
    return lambda: ops.conv2d_fwd(x, w, padding=(pd, pd), ch_scale=csc, relu=True, out=out, scale_out=sc, want_scale=heavy_epi,
                                  track_absmax=True)


def timed(fn, n, stream=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


XS = {"3x3 256->256 @14": (14, 256, 256, 3, False), "1x1 1024->256 @14": (14, 1024, 256, 1, False), "3x3 64->64 @56": (56, 64, 64, 3, False)}
YS = {"1x1 64->256 @56 (+t)": (56, 64, 256, 1, True), "1x1 256->64 @56": (56, 256, 64, 1, False), "1x1 128->512 @28 (+t)": (28, 128, 512, 1, True)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a1, a2 = ops.AbsmaxArena(), ops.AbsmaxArena()
fx = {k: layer(*v) for k, v in XS.items()}
fy = {k: layer(*v) for k, v in YS.items()}
for f in list(fx.values()) + list(fy.values()):
    f(); f()
torch.cuda.synchronize()
print(f"sub-batch {B} images; us per launch alone, then X on stream 1 || Y on stream 2 (launch counts balanced to equal time)")
for kx, X in fx.items():
    for ky, Y in fy.items():
        res = []
        for rnd in range(3):
            tx = timed(X, 20)
            ty = timed(Y, 20)
            nx = 40
            ny = max(1, round(nx * tx / ty))
            cur = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            s1.wait_stream(cur); s2.wait_stream(cur)
            # issue interleaved so that neither queue runs dry
            ix = iy = 0
            while ix < nx or iy < ny:
                if ix < nx and (iy >= ny or ix * ny <= iy * nx):
                    with torch.cuda.stream(s1):
                        X()
                    ix += 1
                else:
                    with torch.cuda.stream(s2):
                        Y()
                    iy += 1
            cur.wait_stream(s1); cur.wait_stream(s2)
            e1.record()
            torch.cuda.synchronize()
            both = e0.elapsed_time(e1) * 1e3
            serial = nx * tx + ny * ty
            res.append((both / serial, tx, ty, nx, ny, both, serial, max(nx * tx, ny * ty)))
        r = min(res)
        print(f"X {kx:20s} {r[1]:7.1f} us | Y {ky:22s} {r[2]:7.1f} us | {r[3]}x || {r[4]}y: {r[5]:9.0f} us vs serial {r[6]:9.0f} (ratio {r[0]:.3f}; "
              f"perfect overlap {r[7] / r[6]:.3f})", flush=True)

print("\nfloat4 copy y = x (read S + write S), and chain x -> y -> z -> x of three S-sized buffers; GB/s of bytes moved")
for mb in (8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    t_copy = timed(lambda: y.copy_(x), 30)
    def chain():
        torch.mul(x, 1.0001, out=y); torch.mul(y, 1.0001, out=z); torch.mul(z, 1.0001, out=x)
    chain()
    t_chain = timed(chain, 10) / 3
    print(f"S = {mb:5d} MiB   copy {2 * mb * 1.048576 / t_copy * 1e3:8.0f} GB/s   chain {2 * mb * 1.048576 / t_chain * 1e3:8.0f} GB/s", flush=True)
    del x, y, z
