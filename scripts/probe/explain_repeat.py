"""Is a whole explanation pass reproducible run to run (optionally while other processes share the GPU)?  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import engine, synth
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
B = int(os.environ.get("B", "128"))
net = synth.build_bcosified_resnet("resnet50").to("cuda")
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
eng = engine.attach(net)
x = synth.synthetic_images(B, seed=4321).to("cuda")
ref = None
bad = 0
for it in range(int(os.environ.get("ITERS", "6"))):
    out = eng.explain(x, want_weights=True)
    cur = {k: out[k].clone() for k in ("logits", "contribution_map", "dynamic_linear_weights")}
    if ref is None:
        ref = cur
        continue
    for k in cur:
        if not torch.equal(cur[k], ref[k]):
            bad += 1
            d = (cur[k] - ref[k]).abs()
            imgs = (d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()
            print(f"[{tag}] iter {it} {k}: max abs {float(d.max()):.3e}, images {imgs[:10]} ({len(imgs)})", flush=True)
print(f"[{tag}] done, {bad} mismatches", flush=True)
