import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib
def bench(M, K, N, iters=10):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.empty(M, N, device="cuda")
    f = lambda: ops.matmul_nt(a, w, out=out)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return 2.0*M*K*N/(e0.elapsed_time(e1)/iters)/1e9
torch.manual_seed(0)
M, K, N = 1024, 2304, 256
a = torch.randn(M, K) * torch.rand(M, 1) * 3; w = torch.randn(N, K) / K ** 0.5
ref = (a.double() @ w.double().t())
for mode in ("f32", "bf16x3"):
    lib.set_contraction_mode(mode)
    out = ops.matmul_nt(a.cuda(), w.cuda()).cpu().double()
    err = (out - ref).abs().max() / ref.abs().max(); rel = (out - ref).norm() / ref.norm()
    print(mode, f"max-err/max {err:.2e} relL2 {rel:.2e}", " TF/s:", " ".join(f"{bench(*s):6.1f}" for s in [(8192, 8192, 8192), (65536, 2304, 256), (50176, 2304, 256), (802816, 576, 64), (802816, 64, 256), (12544, 4608, 512)]))
