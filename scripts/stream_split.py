"""Experiment: explain two half batches concurrently on two HIP streams (fills launch tails)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth, engine
net = synth.build_bcosified_resnet("resnet50").to("cuda")
with torch.no_grad(): synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
eng = engine.attach(net)
x = synth.synthetic_images(256, seed=1000).to("cuda")
def run_single(): return eng.explain(x)
streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("NS", "2")))]
def run_split():
    n = len(streams); outs = [None] * n
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            lo, hi = i * 256 // n, (i + 1) * 256 // n
            outs[i] = eng.explain(x[lo:hi])
    for s in streams: cur.wait_stream(s)
    return outs
for name, fn in (("single", run_single), ("split", run_split), ("single", run_single), ("split", run_split)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms/step {256/dt:.0f} img/s")
