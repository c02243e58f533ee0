"""Are the 3x3 launches (input-patch loop) reproducible while other processes share the GPU?  (development aid)
Run N copies at once:  for i in $(seq 8); do python scripts/probe/patch_stress.py $i & done; wait"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
iters = int(os.environ.get("ITERS", "100"))
dev = "cuda"
B = int(os.environ.get("B", "64"))
shapes = [(B, 14, 256, 256), (B, 28, 128, 128), (B, 56, 64, 64), (B, 7, 512, 512)]
g = torch.Generator().manual_seed(7)
bad = 0
for (N, H, Cin, Cout) in shapes:
    x = ops.ensure_absmax(torch.randn(N, H, H, Cin, generator=g).to(dev))
    w = ops.mark_static((torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev))
    ref = None
    for it in range(iters):
        y, sc, nrm = ops.conv2d_fwd(x, w, stride=(1, 1), padding=(1, 1), relu=True, want_scale=False, want_norm=True, track_absmax=True)
        cur = (y.clone(), nrm.clone(), ops.absmax_of(y).clone())
        if ref is None:
            ref = cur
            continue
        for j, (a, b) in enumerate(zip(cur, ref)):
            if not torch.equal(a, b):
                bad += 1
                d = (a.float() - b.float()).abs()
                idx = (d > 0).nonzero()
                print(f"[{tag}] MISMATCH shape {(N, H, Cin, Cout)} iter {it} output {j}: max abs {float(d.max()):.3e} in {int((d > 0).sum())} of "
                      f"{d.numel()} elements, first {idx[:3].tolist()} last {idx[-1].tolist()}", flush=True)
                break
print(f"[{tag}] done, {bad} mismatching iterations", flush=True)
