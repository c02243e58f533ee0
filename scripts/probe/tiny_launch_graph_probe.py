"""How much device time do ~400 tiny dependent launches cost in a stream, issued eagerly against replayed from a hipGraph?
(the weight-image preparation of a training step: scripts/probe/train_host_probe.py; profiles/r06_small_probes.txt (14))
usage: python scripts/probe/tiny_launch_graph_probe.py   (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "b-cosification_amd"))
import torch
from bcos_hip import lib, ops
lib.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
shapes = [(64, 64, 1, 1), (64, 64, 3, 3), (256, 64, 1, 1), (128, 256, 1, 1), (128, 128, 3, 3), (512, 128, 1, 1), (256, 512, 1, 1),
          (256, 256, 3, 3), (1024, 256, 1, 1), (512, 1024, 1, 1), (512, 512, 3, 3), (2048, 512, 1, 1)]
ws = [torch.randn(s, device=dev) * 0.05 for s in shapes for _ in range(4)]           # 48 layers

def prep():
    keep = []
    for w in ws:
        wk = ops.mark_static(w.permute(0, 2, 3, 1).contiguous())
        keep.append(ops.split_weights_f16x2(wk, w.shape[2] * w.shape[3]))
        wt = ops.mark_static(w.flip(2, 3).permute(1, 2, 3, 0).contiguous())
        keep.append(ops.split_weights_f16x2(wt, w.shape[2] * w.shape[3]))
    return keep

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, 1e3 * (t1 - t0) / n

d, h = timed(prep)
print(f"eager : device {d:.3f} ms   host issue {h:.3f} ms per pass ({len(ws) * 2} images)")
g = torch.cuda.CUDAGraph()
prep(); torch.cuda.synchronize()
with torch.cuda.graph(g):
    static = prep()
d, h = timed(g.replay)
print(f"graph : device {d:.3f} ms   host issue {h:.3f} ms per replay")
