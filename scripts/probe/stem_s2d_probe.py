"""What would the 7x7 / 2 stem forward cost as a 4x4 / 1 convolution over a space-to-depth input on the input-patch loop?  Proxy: the existing
4 x 4-tap, <= 32-column patch configuration (the stem gradient's) run FORWARD over a [256, 112, 112, 32] input with 32 output columns (the stem
has 64: at most twice this, the patch refill shared), next to the stem forward as it is (7x7 / 2 over [256, 224, 224, 8]).  Development probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib as blib
from bcos_hip.lib import BCOS_CONV_EPS
B = 256
g = torch.Generator().manual_seed(0)
def timeit(f):
    best = 1e9
    for _ in range(4):
        f(); f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best
# the stem as it is
x8 = ops.ensure_absmax(torch.rand(B, 224, 224, 8, generator=g).to("cuda"))
w7 = ops.mark_static((torch.randn(64, 7, 7, 8, generator=g) / 20).to("cuda"))
csc = (torch.rand(64, generator=g) + 0.5).to("cuda")
out = torch.empty(B, 112, 112, 64, device="cuda"); sc = torch.empty_like(out)
t_now = timeit(lambda: ops.conv2d_fwd(x8, w7, stride=(2, 2), padding=(3, 3), ch_scale=csc, relu=True, out=out, scale_out=sc, want_scale=True, track_absmax=False))
# proxy: 4x4 taps over the space-to-depth input, 32 columns, on the patch loop
xs = ops.ensure_absmax(torch.rand(B, 112, 112, 32, generator=g).to("cuda"))
res = {}
for cout in (32, 64):
    w4 = ops.mark_static((torch.randn(cout, 4, 4, 32, generator=g) / 20).to("cuda"))
    c2 = (torch.rand(cout, generator=g) + 0.5).to("cuda")
    o2 = torch.empty(B, 112, 112, cout, device="cuda"); s2 = torch.empty_like(o2)
    geom = dict(N=B, H=112, W=112, C=32, P=112, Q=112, in_sh=1, in_sw=1, dh0=-2, dw0=-2, dstep_h=1, dstep_w=1, TH=4, TW=4, OH=112, OW=112,
                out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=cout)
    res[cout] = timeit(lambda: ops.tapconv(xs, w4, geom, out=o2, scale_out=s2, ch_scale=c2, bcos_mode=BCOS_CONV_EPS, b=2.0, relu=1, track_absmax=False))
print(f"stem forward now (7x7/2, K = 392, N = 64): {t_now:.0f} us;  4x4/1 over s2d input, K = 512: N = 32 {res[32]:.0f} us (patch loop), N = 64 {res[64]:.0f} us (per-tap loop: no 64-column 16-tap patch configuration)")
