import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
M, K, N = [int(v) for v in os.environ.get("MKN", "8192,4096,4096").split(",")]
a = torch.randn(M, K, device="cuda"); w = ops.mark_static(torch.randn(N, K, device="cuda") / K ** 0.5)
ops.ensure_absmax(a)
out = torch.empty(M, N, device="cuda")
for _ in range(3):
    ops.matmul_nt(a, w, out=out)
torch.cuda.synchronize()
