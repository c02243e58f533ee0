"""time one MFMA-bound shape for knock-out builds (development aid)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
dev = "cuda"
from bcos_hip import lib as blib
blib.set_option("h2_tile", 2 if os.environ.get("TILE", "128x128") == "128x256" else 1)
for (N, H, Cin, Cout, k) in [(256, 14, 256, 256, 3), (256, 14, 1024, 256, 1)]:
    x = torch.randn(N, H, H, Cin, device=dev); ops.ensure_absmax(x)
    w = ops.mark_static(torch.randn(Cout, k, k, Cin, device=dev) / math.sqrt(Cin * k * k))
    out = torch.empty(N, H, H, Cout, device=dev)
    f = lambda: ops.conv2d_fwd(x, w, stride=(1, 1), padding=(k // 2, k // 2), out=out, mode=0, b=1.0)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("BCOS_HIP_LIB", "default").split("/")[-1], (N * H * H, Cin * k * k, Cout), f"{e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
