import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
def bench(M, K, N, iters=10):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.empty(M, N, device="cuda")
    f = lambda: ops.matmul_nt(a, w, out=out)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N) in ((131072, 256), (131072, 64)):
    print(f"M={M} N={N}")
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        ms = bench(M, K, N)
        print(f"  K={K:5d}: {ms*1e3:8.1f} us  {2.0*M*K*N/ms/1e9:6.1f} TF/s   per-Kstep {ms*1e3/(K/32):6.2f} us")
