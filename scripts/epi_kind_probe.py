"""Time one launch shape with the specialised and with the general epilogue (development aid):
   ROWS=100352 CIN=192 COUT=768 RELU=2 python scripts/epi_kind_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
from bcos_hip import lib as _l
from bcos_hip.lib import BCOS_LINEAR_EPS
rows, cin, cout, relu = (int(os.environ.get(k, d)) for k, d in (("ROWS", "100352"), ("CIN", "192"), ("COUT", "768"), ("RELU", "2")))
x = torch.randn(rows, cin, device="cuda")
w = ops.mark_static(torch.randn(cout, cin, device="cuda") / cin ** 0.5)
g = dict(N=1, H=1, W=rows, C=cin, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1, TH=1, TW=1, OH=1, OW=rows,
         out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=cout)
y = torch.empty(rows, cout, device="cuda"); t = torch.empty_like(y)
for generic in (False, True, False, True):
    _l.set_option("epi_generic", 1 if generic else 0)
    with ops.no_absmax():
        for _ in range(3):
            ops.tapconv(x, w, g, out=y, scale_out=t, bcos_mode=BCOS_LINEAR_EPS, b=2.0, relu=relu)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.tapconv(x, w, g, out=y, scale_out=t, bcos_mode=BCOS_LINEAR_EPS, b=2.0, relu=relu)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    nbytes = 4 * (rows * cin + 2 * rows * cout)
    print(f"{'general' if generic else 'specialised'} epilogue: {us:.1f} us, {nbytes / us / 1e6:.2f} TB/s algorithmic")
