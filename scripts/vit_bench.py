import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth, vit_engine, ops
B = int(os.environ.get("B", "512"))
net = synth.build_bcosified_vit().to("cuda")
x = synth.synthetic_images(B).to("cuda")
with torch.no_grad(): synth.calibrate(net, x[:8])
eng = vit_engine.attach(net)
for mode, fn in (("fwd", lambda: eng.forward(x)), ("explain", lambda: eng.explain(x))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"ViT-Ti B={B} {mode}: {dt*1e3:.1f} ms {B/dt:.0f} img/s  (B-cos GEMMs {1.752*(1 if mode=='fwd' else 2)*B/dt/1e3:.1f} TF/s)")
