"""Upper bound of any attention speed-up on the ViT-Ti step: the step with the attention launches knocked out (outputs unwritten: timing
only) against the real step, three-stream, batch 512."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd"))
import torch
from bcos_hip import ops, synth, vit_engine
net = synth.build_bcosified_vit(seed=0).to("cuda")
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
eng = vit_engine.attach(net)
x = synth.synthetic_images(512, seed=1000).to("cuda")
def run(n=12):
    for _ in range(3): eng.explain(x)
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.explain(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
real = run()
f0, b0 = ops.attention_fwd, ops.attention_bwd_v
def fwd_ko(qkv, heads, scale, want_stats=False, want_absmax=False):
    B, T, ti = qkv.shape
    out = torch.empty((B, T, ti // 3), device=qkv.device)
    if want_absmax: ops.ensure_absmax(out)
    return out, (torch.empty((B, heads, T, 2), device=qkv.device) if want_stats else None)
def bwd_ko(qkv, stats, gout, heads, scale, want_absmax=False):
    gv = torch.empty_like(gout)
    if want_absmax: ops.ensure_absmax(gv)
    return gv
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
ops.attention_fwd, ops.attention_bwd_v = fwd_ko, bwd_ko
ko = run()
ops.attention_fwd, ops.attention_bwd_v = f0, b0
print(f"real step {real:.3f} ms   attention knocked out {ko:.3f} ms   (bound on any attention speed-up: {real - ko:.3f} ms)")
