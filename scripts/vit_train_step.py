"""One training step (BCE) of the B-cosified ViT-Ti on the HIP kernels (development aid): B=64 python scripts/vit_train_step.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from bcos_hip import synth
B = int(os.environ.get("B", "64"))
net = synth.build_bcosified_vit().to("cuda")
x = synth.synthetic_images(B).to("cuda")
with torch.no_grad():
    synth.calibrate(net, x[:8])
net.train()
target = F.one_hot(torch.randint(0, 1000, (B,)), 1000).float().cuda()
params = [p for p in net.parameters() if p.requires_grad]
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    loss = F.binary_cross_entropy_with_logits(net(x), target)
    grads = torch.autograd.grad(loss, params)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"step {it}: loss {float(loss):.5f}, {dt * 1e3:.1f} ms, {B / dt:.0f} images/s, finite grads {all(torch.isfinite(g).all() for g in grads)}")
