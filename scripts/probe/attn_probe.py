"""Time of the attention kernels at the ViT-Ti shapes of the bench (development probe): python scripts/probe/attn_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
B, T, H = int(os.environ.get("B", "256")), 197, 3
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * H * 64, generator=g).to("cuda")
go = torch.randn(B, T, H * 64, generator=g).to("cuda")
out, stats = ops.attention_fwd(qkv, H, 0.125, want_stats=True, want_absmax=True)
def timeit(f):
    best = 1e9
    for _ in range(3):
        f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return best
print(f"B={B}: fwd {timeit(lambda: ops.attention_fwd(qkv, H, 0.125, want_stats=True, want_absmax=True)):.1f} us   "
      f"bwd_v {timeit(lambda: ops.attention_bwd_v(qkv, stats, go, H, 0.125, want_absmax=True)):.1f} us", flush=True)
