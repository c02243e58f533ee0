"""GEMM micro-benchmark of the tapconv kernel in linear mode (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
dev = "cuda"
def bench(M, K, N, bcos=False, iters=10):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    if os.environ.get("STATIC"): ops.mark_static(w)
    out = torch.empty(M, N, device=dev)
    f = (lambda: ops.linear_fwd(a, w, out=out)) if bcos else (lambda: ops.matmul_nt(a, w, out=out))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0*M*K*N/ms/1e9
shapes = [(8192, 8192, 8192), (65536, 2304, 256), (50176, 2304, 256), (802816, 576, 64), (802816, 64, 256)]
if os.environ.get("QUICK"): shapes = shapes[:3]
print(os.environ.get("BCOS_HIP_LIB", "default").split("/")[-1], " ".join(f"{bench(*s):6.1f}" for s in shapes), f"bcos8k {bench(8192, 8192, 8192, bcos=True):6.1f}")
