"""Kernel list of ONE forward + explanation step (development aid): counts and total time per kernel name, torch.profiler."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from bcos_hip import engine, synth
net = synth.build_bcosified_resnet(os.environ.get("ARCH", "resnet50")).to("cuda")
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
eng = engine.attach(net)
x = synth.synthetic_images(int(os.environ.get("B", "256")), seed=1000).to("cuda")
for _ in range(3):
    eng.explain(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    eng.explain(x)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith("CUDA"):
        agg[ev.name][0] += 1
        agg[ev.name][1] += ev.device_time if hasattr(ev, "device_time") else ev.cuda_time
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total device time {tot / 1e3:.2f} ms over {sum(v[0] for _, v in rows)} kernels")
for name, (cnt, us) in rows[:40]:
    if "tapconv_kernel" in name:
        continue
    print(f"{cnt:5d} x {us / max(cnt, 1):8.1f} us = {us / 1e3:7.3f} ms  {name[:110]}")
