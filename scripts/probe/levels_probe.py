"""One 3x3 launch of the input-patch loop on an image with a wide dynamic range (several operand-scale levels per tile): finite?
close to fp64?  Development probe for the out-of-line level passes (csrc/bcos_tapconv.hip: tile_body_p_more)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, lib as blib
dev = "cuda"
g = torch.Generator().manual_seed(0)
for (N, H, Cin, Cout) in [(2, 14, 64, 128), (2, 14, 256, 256), (2, 56, 64, 64)]:
    x = torch.randn(N, H, H, Cin, generator=g)
    x[:, : H // 2] *= 1e-8
    x = ops.ensure_absmax(x.to(dev))
    w = ops.mark_static((torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev))
    for lv in (0, 1):
        blib.set_option("patch_levels", lv)
        y = ops.conv2d_fwd(x, w, stride=(1, 1), padding=(1, 1), relu=False)[0]
        torch.cuda.synchronize()
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1)
        nrm = (torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2) ** 2, torch.ones(1, Cin, 3, 3, device=dev, dtype=torch.float64), padding=1) + 1e-6).sqrt()
        y64 = (ref * ref.abs() / nrm).permute(0, 2, 3, 1)
        top = ((y[:, : H // 2 - 1].double() - y64[:, : H // 2 - 1]).norm() / y64[:, : H // 2 - 1].norm()).item()
        bot = ((y[:, H // 2 + 1:].double() - y64[:, H // 2 + 1:]).norm() / y64[:, H // 2 + 1:].norm()).item()
        print((N, H, Cin, Cout), "levels", lv, "finite", bool(torch.isfinite(y).all()), f"dark half relL2 {top:.2e}  bright half {bot:.2e}", flush=True)
