"""one forward-type HBM-bound launch (64 -> 256 @ 56^2, residual + ReLU + scale_out) for the counter passes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
from bcos_hip.lib import BCOS_EPI_SCALE_GATE_LSB
N, H, Cin, Cout = 256, int(os.environ.get("HH", "56")), int(os.environ.get("CIN", "64")), int(os.environ.get("COUT", "256"))
x = ops.ensure_absmax(torch.randn(N, H, H, Cin, device="cuda")); w = ops.mark_static(torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5)
res = torch.randn(N, H, H, Cout, device="cuda"); out = torch.empty_like(res); sc = torch.empty_like(res)
csc = torch.rand(Cout, device="cuda") + 0.5
for _ in range(3):
    ops.conv2d_fwd(x, w, ch_scale=csc, addend=res, relu=True, out=out, scale_out=sc, want_scale=True, flags=BCOS_EPI_SCALE_GATE_LSB,
                   track_absmax=False)
torch.cuda.synchronize()
