"""How wide is the dynamic range INSIDE an image at the operands of the launches that use per-image scales (3 x 3 layers, stem
gradient), forward and explanation pass, on smooth and structured images?  Prints, per launch, max over images of
exponent(image max) - exponent(smallest nonzero per-pixel max); > 12 means the tile ladder of the input-patch loop has work to do."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import engine, ops, synth
dev = "cuda"
arch = os.environ.get("ARCH", "resnet50")
net = synth.build_bcosified_resnet(arch).to(dev)
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to(dev))
eng = engine.attach(net)
log = []
orig = ops.image_absmax
def spy(am, n, hw):
    out = orig(am, n, hw)
    log.append((n, hw, out))
    return out
ops.image_absmax = spy
for name, x in (("smooth", synth.synthetic_images(8, seed=5)), ("structured", synth.structured_images(8))):
    log.clear()
    eng.explain(x.to(dev))
    torch.cuda.synchronize()
    rows = []
    for (n, hw, out) in log:
        mx, mn = out[0].cpu().long(), out[1].cpu().long()
        mn = torch.where(mn < 0, mx, mn)          # (all-zero image)
        rows.append((hw, int(((mx >> 23) - (mn >> 23)).max())))
    print(name, "launches with image scales:", len(rows), " exponent range per launch (pixels per image : bits):",
          " ".join(f"{hw}:{d}" for hw, d in rows), flush=True)
