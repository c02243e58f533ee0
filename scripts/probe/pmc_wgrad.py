"""One weight-gradient launch for the SQ counter passes (scripts/probe/pmc_wgrad.sh): SHAPE=H,Cin,Cout,k (batch 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd"))
import torch
from bcos_hip import ops
H, Cin, Cout, k = [int(v) for v in os.environ.get("SHAPE", "14,256,1024,1").split(",")]
x = torch.randn(64, H, H, Cin, device="cuda"); g = torch.randn(64, H, H, Cout, device="cuda")
for _ in range(3):
    ops.conv2d_wgrad(g, x, Cin, Cout, (k, k), (1, 1), (k // 2, k // 2), (1, 1))
torch.cuda.synchronize()
