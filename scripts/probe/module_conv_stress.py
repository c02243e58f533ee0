"""Is the nn.Module-path B-cos convolution reproducible while other processes share the GPU?  (development aid)
Run N copies at once:  for i in $(seq 8); do python scripts/probe/module_conv_stress.py $i & done; wait
Each iteration restores the weight, runs the layer, rescales the weight in place (what synth.calibrate does), runs it again and
compares both outputs bit for bit with those of iteration 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos.modules.bcosifyconv2d import BcosifyConv2d
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
iters = int(os.environ.get("ITERS", "150"))
dev = "cuda"
shapes = [(8, 14, 256, 1024, 1, 0), (8, 14, 1024, 256, 1, 0), (8, 14, 256, 256, 3, 1), (8, 7, 2048, 512, 1, 0), (8, 7, 512, 2048, 1, 0),
          (8, 28, 128, 512, 1, 0), (8, 56, 64, 256, 1, 0)]
g = torch.Generator().manual_seed(7)
bad = 0
with torch.no_grad():
    for (N, H, Cin, Cout, k, p) in shapes:
        m = BcosifyConv2d(Cin, Cout, k, 1, p, b=2).to(dev).eval()
        w0 = (torch.randn(m.linear.weight.shape, generator=g) / (k * k * Cin) ** 0.5).to(dev)
        x = torch.randn(N, Cin, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        ref = None
        for it in range(iters):
            m.linear.weight.copy_(w0)
            y1 = m(x)
            gain = y1.pow(2).mean().sqrt().clamp_min(1e-30).pow(-0.5)
            m.linear.weight.mul_(gain)
            y2 = m(x)
            var = y2.var(dim=(0, 2, 3), unbiased=False)
            cur = (y1.clone(), y2.clone(), var.clone())
            if ref is None:
                ref = cur
                continue
            for j, (a, b) in enumerate(zip(cur, ref)):
                if not torch.equal(a, b):
                    bad += 1
                    d = (a - b).abs()
                    print(f"[{tag}] MISMATCH shape {(N, H, Cin, Cout, k)} iter {it} output {j}: max abs {float(d.max()):.3e} "
                          f"in {int((d > 0).sum())} of {d.numel()} elements, rows {(d > 0).nonzero()[:3].tolist()}", flush=True)
                    break
print(f"[{tag}] done, {bad} mismatching iterations", flush=True)
