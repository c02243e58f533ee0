"""Experiment: the batch split into S independent sub-batches, each forward + explanation pass issued on its own HIP stream
(images are independent in eval mode), against the single-stream pass.  Overlaps HBM-bound launches of one sub-batch with
MFMA-bound launches of another and fills the tails of short launches.  Usage: python scripts/dual_stream.py [S ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import engine, synth

B = int(os.environ.get("B", "256"))
arch = os.environ.get("ARCH", "resnet50")
dev = "cuda"
net = synth.build_bcosified_resnet(arch).to(dev)
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to(dev))
x = synth.synthetic_images(B, seed=1000).to(dev)
splits = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
ref = None
for S in splits:
    engs = [engine.ResNetEngine(net) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    parts = list(x.chunk(S))

    def step():
        outs = []
        cur = torch.cuda.current_stream()
        for eng, st, xp in zip(engs, streams, parts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(eng.explain(xp))
        for st in streams:
            cur.wait_stream(st)
        return outs

    for _ in range(3):
        outs = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    logits = torch.cat([o["logits"] for o in outs])
    maps = torch.cat([o["contribution_map"] for o in outs])
    if ref is None:
        ref = (logits, maps)
    same = torch.equal(logits, ref[0]) and torch.equal(maps, ref[1])
    print(f"streams {S}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} images/s, identical to 1 stream: {same}", flush=True)
