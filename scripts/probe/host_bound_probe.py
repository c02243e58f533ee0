"""Is a step host-bound?  Time until engine.explain() RETURNS (all launches issued) against the time until the device has finished.
python scripts/probe/host_bound_probe.py [vit_ti|resnet50] (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_ti"
if arch == "vit_ti":
    from bcos_hip import vit_engine
    net = synth.build_bcosified_vit("simple_vit_ti_patch16_224").cuda().eval()
    eng = vit_engine.attach(net); B = 512
else:
    from bcos_hip import engine
    net = synth.build_bcosified_resnet(arch).cuda().eval()
    with torch.no_grad(): synth.calibrate(net, synth.synthetic_images(8).cuda())
    eng = engine.attach(net); B = 256
x = synth.synthetic_images(B).cuda()
for _ in range(3): eng.explain(x)
torch.cuda.synchronize()
iss, tot = [], []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.explain(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"{arch} batch {B}: launches issued after {min(iss):.2f} ms (median {sorted(iss)[len(iss)//2]:.2f}), device done after {min(tot):.2f} ms (median {sorted(tot)[len(tot)//2]:.2f})")
