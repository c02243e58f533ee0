"""Time 1x1 forward launches (B-cos scale + ReLU + stored multiplier) of given shapes in the f16x2 and bf16x3 contraction modes
(development aid):  SHAPES="802816,256,64;200704,512,128" python scripts/mode_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
from bcos_hip.lib import BCOS_LINEAR_EPS
shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get(
    "SHAPES", "802816,256,64;802816,256,128;200704,512,128;200704,512,256;50176,1024,256;50176,1024,512;12544,2048,512").split(";")]
for rows, cin, cout in shapes:
    x = ops.ensure_absmax(torch.randn(rows, cin, device="cuda"))
    w = ops.mark_static(torch.randn(cout, cin, device="cuda") / cin ** 0.5)
    g = dict(N=1, H=1, W=rows, C=cin, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1, TH=1, TW=1, OH=1, OW=rows,
             out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=cout)
    y = torch.empty(rows, cout, device="cuda"); t = torch.empty_like(y)
    res = {}
    for mode in ("f16x2", "bf16x3", "f16x2", "bf16x3"):
        for _ in range(3):
            ops.tapconv(x, w, g, out=y, scale_out=t, bcos_mode=BCOS_LINEAR_EPS, b=2.0, relu=1, contraction=mode, track_absmax=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.tapconv(x, w, g, out=y, scale_out=t, bcos_mode=BCOS_LINEAR_EPS, b=2.0, relu=1, contraction=mode, track_absmax=False)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) * 100)
    nbytes = 4 * (rows * cin + 2 * rows * cout)
    print(f"M={rows} K={cin} N={cout}: " + "  ".join(f"{m} {min(v):.1f} us ({nbytes / min(v) / 1e6:.2f} TB/s, {2e-6 * rows * cin * cout / min(v):.0f} TF/s)" for m, v in res.items()))
    del x, w, y, t
