"""HBM throughput of plain streaming patterns at the size of the 56x56x256 activations (development aid): what the
HBM-bound tapconv launches can hope for."""
import torch
n = 802816 * 256
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.empty_like(a); d = torch.empty_like(a)
small = torch.randn(802816 * 64, device="cuda")
def t(f, bytes_, name):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:34s} {ms:7.3f} ms  {bytes_ / ms / 1e6:7.0f} GB/s")
t(lambda: c.copy_(a), 8 * n, "copy 1R 1W")
t(lambda: torch.add(a, b, out=c), 12 * n, "add 2R 1W")
t(lambda: torch.mul(a, 2.0, out=c), 8 * n, "scale 1R 1W")
def two_out():
    torch.mul(a, 2.0, out=c); torch.add(a, 1.0, out=d)
t(two_out, 16 * n, "2 kernels (1R 1W each)")
t(lambda: torch.addcmul(a, b, c, out=d), 16 * n, "addcmul 3R 1W")
t(lambda: a.sum(), 4 * n, "sum 1R")
t(lambda: c.fill_(1.0), 4 * n, "fill 1W")
