"""f16x2 contraction: accuracy against fp64 / the other modes, per-pixel maxima, per-shape timing (development aid)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib as blib
dev = "cuda"

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))

def mm(a, w, mode):
    blib.set_contraction_mode(mode)
    a = a.to(dev); w = ops.mark_static(w.to(dev))
    ops.ensure_absmax(a)
    return ops.matmul_nt(a, w)

if "acc" in sys.argv or len(sys.argv) == 1:
    g = torch.Generator().manual_seed(3)
    a = torch.randn(512, 2304, generator=g) * (torch.rand(512, 1, generator=g) * 3)
    w = torch.randn(256, 2304, generator=g) / 48
    ref = a.double() @ w.double().t()
    for mode in ("f32", "bf16x3", "f16x2"):
        print(mode, "plain", rel(mm(a, w, mode), ref))
    # per-row magnitudes from 1e-30 to 1e30, per-column too
    rs = 10.0 ** (torch.rand(512, 1, generator=g) * 60 - 30)
    cs = 10.0 ** (torch.rand(256, 1, generator=g) * 12 - 6)
    a2, w2 = (a * rs), (w * cs)
    ref2 = a2.double() @ w2.double().t()
    for mode in ("bf16x3", "f16x2"):
        out = mm(a2, w2, mode).double().cpu()
        rowerr = ((out - ref2).norm(dim=1) / ref2.norm(dim=1)).max()
        print(mode, "row/col scaled: worst row relL2", float(rowerr))
    # wide dynamic range inside a row: a few huge outliers
    a3 = a.clone(); a3[:, ::97] *= 1e4; a3[:, 5::131] *= 1e-6
    ref3 = a3.double() @ w.double().t()
    for mode in ("bf16x3", "f16x2"):
        print(mode, "outliers", rel(mm(a3, w, mode), ref3))
    # tiny whole tensor
    for sc in (1e-20, 1e15, 1e-38):
        for mode in ("bf16x3", "f16x2"):
            print(mode, f"scale {sc:g}", rel(mm(a * sc, w, mode), ref * sc))

if "conv" in sys.argv or len(sys.argv) == 1:
    from oracle import bcos_oracle as O
    g = torch.Generator().manual_seed(5)
    for (N, Cin, H, W, Cout, k, s, p) in [(2, 64, 14, 14, 64, 3, 1, 1), (2, 128, 15, 13, 160, 3, 2, 1), (2, 256, 14, 14, 512, 1, 2, 0),
                                           (2, 8, 32, 32, 64, 7, 2, 3), (3, 512, 7, 7, 1000, 1, 1, 0), (2, 64, 28, 28, 256, 1, 1, 0),
                                           (4, 128, 28, 28, 128, 3, 1, 1)]:
        x = torch.randn(N, Cin, H, W, generator=g)
        wt = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        xr = x.clone().requires_grad_(True)
        y_ref, s_ref = O.bcos_conv2d(xr, wt, stride=s, padding=p, detach=True, return_scale=True)
        gy = torch.randn(y_ref.shape, generator=g)
        (gx_ref,) = torch.autograd.grad(y_ref, xr, gy)
        for mode in ("bf16x3", "f16x2"):
            blib.set_contraction_mode(mode)
            xh = ops.ensure_absmax(x.permute(0, 2, 3, 1).contiguous().to(dev))
            wk = ops.mark_static(wt.permute(0, 2, 3, 1).contiguous().to(dev))
            y, sc, _ = ops.conv2d_fwd(xh, wk, stride=(s, s), padding=(p, p), want_scale=True)
            am = ops.absmax_of(y)
            if mode == "f16x2":
                exp = y.abs().amax(dim=-1).reshape(-1).view(torch.int32)
                assert am is not None and torch.equal(am, exp), "absmax mismatch"
            glin = ops.ensure_absmax(ops.mul(gy.permute(0, 2, 3, 1).contiguous().to(dev), sc))
            gx = ops.DgradPlan(wt.to(dev), (s, s), (p, p)).run(glin, H, W)
            print(mode, (N, Cin, H, W, Cout, k, s, p), "y", f"{rel(y.permute(0, 3, 1, 2), y_ref):.2e}", "gx", f"{rel(gx.permute(0, 3, 1, 2), gx_ref):.2e}")

def timeit(f, iters=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

if "time" in sys.argv:
    # (N, H, Cin, Cout, k) ResNet-50 shapes at batch 256
    shapes = [(256, 14, 256, 256, 3), (256, 28, 128, 128, 3), (256, 56, 64, 64, 3), (256, 7, 512, 512, 3),
              (256, 14, 1024, 256, 1), (256, 14, 256, 1024, 1), (256, 28, 512, 128, 1), (256, 28, 128, 512, 1),
              (256, 56, 64, 256, 1), (256, 56, 256, 64, 1), (256, 7, 2048, 512, 1), (256, 7, 512, 2048, 1)]
    tiles = os.environ.get("TILES", "128x128,256x128,128x256").split(",")
    for (N, H, Cin, Cout, k) in shapes:
        x = torch.randn(N, H, H, Cin, device=dev)
        w = ops.mark_static(torch.randn(Cout, k, k, Cin, device=dev) / math.sqrt(Cin * k * k))
        out = torch.empty(N, H, H, Cout, device=dev); sc = torch.empty_like(out)
        fl = 2.0 * N * H * H * Cin * Cout * k * k
        res = []
        blib.set_contraction_mode("bf16x3")
        ms = timeit(lambda: ops.conv2d_fwd(x, w, stride=(1, 1), padding=(k // 2, k // 2), out=out, scale_out=sc, want_scale=True))
        res.append(f"bf16x3 {ms:6.3f} ms {fl/ms/1e9:6.1f} TF")
        blib.set_contraction_mode("f16x2")
        ops.ensure_absmax(x)
        for t in tiles:
            blib.set_option("h2_tile", 2 if t == "128x256" else 1)
            ms = timeit(lambda: ops.conv2d_fwd(x, w, stride=(1, 1), padding=(k // 2, k // 2), out=out, scale_out=sc, want_scale=True))
            res.append(f"h2 {t} {ms:6.3f} ms {fl/ms/1e9:6.1f} TF")
        print((N * H * H, Cin * k * k, Cout, k), " | ".join(res), flush=True)
