"""Is synth.calibrate reproducible while other processes share the GPU?  (development aid)
for i in $(seq 8); do python scripts/probe/calibrate_stress.py $i & done; wait"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
reps = int(os.environ.get("REPS", "8"))
arch = os.environ.get("ARCH", "clip")
ref = None
bad = 0
for rep in range(reps):
    net = (synth.build_bcosified_clip_rn50() if arch == "clip" else synth.build_bcosified_resnet(arch)).to("cuda")
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
    dig = {k: v.double().abs().sum().item() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    if ref is None:
        ref = dig
        continue
    diff = [k for k in dig if dig[k] != ref[k]]
    if diff:
        bad += 1
        print(f"[{tag}] rep {rep}: {len(diff)} entries differ, first: {diff[:3]}  ({dig[diff[0]]!r} vs {ref[diff[0]]!r})", flush=True)
print(f"[{tag}] done: {bad} of {reps - 1} repeats differ from the first; digest of last entry {list(ref.items())[-1]}", flush=True)
