"""Quick GPU sanity script (development aid): layer parity vs the CPU oracle, engine parity, rough timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from bcos_hip import ops, synth, engine
from oracle import bcos_oracle as O

dev = "cuda"
torch.manual_seed(0)

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item(), (a - b).abs().max().item()

def check_conv(N, Cin, H, W, Cout, k, s, p, d=1):
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    xr = x.clone().requires_grad_(True)
    y_ref, s_ref = O.bcos_conv2d(xr, w, stride=s, padding=p, dilation=d, b=2, detach=True, return_scale=True)
    gy = torch.randn_like(y_ref)
    (gx_ref,) = torch.autograd.grad(y_ref, xr, gy)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wk = w.permute(0, 2, 3, 1).contiguous().to(dev)
    y, sc, nrm = ops.conv2d_fwd(xh, wk, stride=(s, s), padding=(p, p), dilation=(d, d), want_scale=True, want_norm=True)
    plan = ops.DgradPlan(w.to(dev), (s, s), (p, p), (d, d))
    glin = ops.mul(gy.permute(0, 2, 3, 1).contiguous().to(dev), sc)
    gx = plan.run(glin, H, W)
    torch.cuda.synchronize()
    r1 = rel(y.permute(0, 3, 1, 2), y_ref.detach())
    r2 = rel(sc.permute(0, 3, 1, 2), s_ref.detach().expand_as(y_ref))
    r3 = rel(gx.permute(0, 3, 1, 2), gx_ref)
    print(f"conv N{N} C{Cin}->{Cout} {H}x{W} k{k} s{s} p{p} d{d}: y relL2 {r1[0]:.2e} max {r1[1]:.2e} | s {r2[0]:.2e} | gx {r3[0]:.2e} max {r3[1]:.2e}")

if "layers" in sys.argv or len(sys.argv) == 1:
    check_conv(2, 64, 14, 14, 64, 1, 1, 0)
    check_conv(2, 64, 14, 14, 256, 1, 1, 0)
    check_conv(2, 64, 14, 14, 64, 3, 1, 1)
    check_conv(2, 128, 14, 14, 128, 3, 2, 1)
    check_conv(2, 128, 15, 13, 160, 3, 2, 1)
    check_conv(2, 256, 14, 14, 512, 1, 2, 0)
    check_conv(2, 8, 32, 32, 64, 7, 2, 3)
    check_conv(1, 32, 9, 9, 40, 3, 1, 2, 2)
    check_conv(3, 512, 7, 7, 1000, 1, 1, 0)

def model_parity(arch, n):
    net = synth.build_bcosified_resnet(arch).to(dev)
    x = synth.synthetic_images(max(n, 4)).to(dev)
    with torch.no_grad():
        rec = synth.calibrate(net, x[:4])
    eng = engine.attach(net)
    xs = x[:n]
    out = net.explain_batch(xs)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    fwd = lambda xx, detach: O.resnet_logits(sd, xx, arch, detach=detach)
    ref = O.explain_batch(fwd, xs.cpu())
    print(arch, "logits", rel(out["logits"], ref["logits"]), "argmax equal", bool((out["prediction"].cpu() == ref["prediction"]).all()),
          ref["prediction"].tolist())
    print(arch, "W(x)", rel(out["dynamic_linear_weights"], ref["dynamic_linear_weights"]))
    print(arch, "contrib", rel(out["contribution_map"], ref["contribution_map"]))
    print("logit stats", ref["logits"].std().item(), (ref["logits"].max(1).values - ref["logits"].topk(2, 1).values[:, 1]).tolist())
    # module path (autograd over HIP modules)
    engine.detach(net)
    out2 = net.explain_batch(xs)
    print(arch, "module-path logits", rel(out2["logits"], ref["logits"]), "W(x)", rel(out2["dynamic_linear_weights"], ref["dynamic_linear_weights"]))
    return net

if "model" in sys.argv or len(sys.argv) == 1:
    model_parity("resnet18", 2)
    model_parity("resnet50", 2)

if "time" in sys.argv or len(sys.argv) == 1:
    net = synth.build_bcosified_resnet("resnet50").to(dev)
    x = synth.synthetic_images(64).to(dev)
    with torch.no_grad():
        synth.calibrate(net, x[:8])
    eng = engine.attach(net)
    B = int(os.environ.get("B", "64"))
    xb = synth.synthetic_images(B).to(dev)
    for mode in ("fwd", "explain"):
        fn = (lambda: eng.forward(xb)) if mode == "fwd" else (lambda: eng.explain(xb))
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3): fn()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        gf = 8.611 * (1 if mode == "fwd" else 2) * B
        print(f"R50 B={B} {mode}: {dt*1e3:.1f} ms  {B/dt:.0f} img/s  {gf/dt/1e3:.1f} TFLOP/s")
