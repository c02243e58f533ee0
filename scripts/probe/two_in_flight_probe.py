"""Two batches in flight: consecutive explain() calls issued on two alternating caller streams, each with its own sub-batch streams and
maxima arenas (`engine.lane`), so that the tail of one step overlaps the head of the next and their launches sit at different phases of
the network.  Development probe: what would a serving loop with two batches in flight gain over the step-after-step loop bench.py times?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import engine, synth
net = synth.build_bcosified_resnet("resnet50").cuda().eval()
with torch.no_grad(): synth.calibrate(net, synth.synthetic_images(8).cuda())
eng = engine.attach(net)
B, K = 256, 20
xs = [synth.synthetic_images(B, seed=s).cuda() for s in (1, 2)]
lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cur = torch.cuda.current_stream()
    outs = []
    for k in range(K):
        if two:
            l = k & 1
            lanes[l].wait_stream(cur)
            with torch.cuda.stream(lanes[l]):
                eng.lane = l
                outs.append(eng.explain(xs[l]))
        else:
            eng.lane = 0
            outs.append(eng.explain(xs[k & 1]))
        if len(outs) > 2: outs.pop(0)
    if two:
        for s in lanes: cur.wait_stream(s)
    torch.cuda.synchronize()
    eng.lane = 0
    return (time.perf_counter() - t0) / K * 1e3
for _ in range(2): run(False); run(True)
for rnd in range(3):
    print(f"step after step {run(False):.2f} ms   two in flight {run(True):.2f} ms", flush=True)
