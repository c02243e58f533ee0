"""one multi-tap forward launch (default 3x3 256 -> 256 @ 14^2, batch 256) for the counter passes: HH, CIN, COUT, KK, ST"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
N, H, Cin, Cout = int(os.environ.get("B", "256")), int(os.environ.get("HH", "14")), int(os.environ.get("CIN", "256")), int(os.environ.get("COUT", "256"))
k, st = int(os.environ.get("KK", "3")), int(os.environ.get("ST", "1"))
x = ops.ensure_absmax(torch.randn(N, H, H, Cin, device="cuda")); w = ops.mark_static(torch.randn(Cout, k, k, Cin, device="cuda") / (k * k * Cin) ** 0.5)
for _ in range(3):
    ops.conv2d_fwd(x, w, stride=(st, st), padding=(k // 2, k // 2), relu=True, want_scale=False, want_norm=False, track_absmax=False)
torch.cuda.synchronize()
