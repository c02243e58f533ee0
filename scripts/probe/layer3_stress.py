"""Repeat the calibration of CLIP RN50's layer3 alone (module path, 8 images) and compare the resulting parameters with the
first repetition, while other copies of this script share the GPU.  (development aid)
MODE=var: additionally recompute every BatchNorm input variance twice and compare (is the torch reduction the flaky op?)"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
iters = int(os.environ.get("ITERS", "30"))
net = synth.build_bcosified_clip_rn50().to("cuda")
layer3 = net.model.layer3
state0 = copy.deepcopy(layer3.state_dict())
grab = {}
h = layer3.register_forward_pre_hook(lambda m, args: grab.setdefault("x", args[0].detach().clone()))
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
h.remove()
x3 = grab["x"]
ref, bad = None, 0
for it in range(iters):
    layer3.load_state_dict(state0)
    with torch.no_grad():
        synth.calibrate(layer3, x3)
    dig = {k: v.double().abs().sum().item() for k, v in layer3.state_dict().items() if v.dtype.is_floating_point}
    if ref is None:
        ref = dig
        continue
    diff = [k for k in dig if dig[k] != ref[k]]
    if diff:
        bad += 1
        print(f"[{tag}] iter {it}: {len(diff)} entries differ, first {diff[0]} ({dig[diff[0]]!r} vs {ref[diff[0]]!r})", flush=True)
print(f"[{tag}] done: {bad} of {iters - 1} repetitions differ", flush=True)
