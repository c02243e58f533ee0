"""The oracle (plain torch ops, fp32) executed by PyTorch-ROCm ON THE DEVICE: what running the reference's algorithm eagerly on this
MI355X gives (MIOpen / rocBLAS fp32 convolutions + autograd).  usage: python scripts/probe/eager_gpu_probe.py [resnet50|resnet18] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
from oracle import bcos_oracle as O
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
net = synth.build_bcosified_resnet(arch, seed=0)
with torch.no_grad():
    pass
sd = {k: v.detach().to(dev) for k, v in net.state_dict().items()}
fwd = lambda xx, detach: O.resnet_logits(sd, xx, arch, detach=detach)
x = synth.synthetic_images(B, seed=321).to(dev)
CH = int(os.environ.get("CH", "64"))
def run(explain=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for lo in range(0, B, CH):
        if explain: O.explain_batch(fwd, x[lo:lo + CH])
        else:
            with torch.no_grad(): fwd(x[lo:lo + CH], False)
    torch.cuda.synchronize(); return time.perf_counter() - t0
run(); run(False)
te = min(run() for _ in range(3)); tf = min(run(False) for _ in range(3))
print(f"{arch} batch {B} (chunks of {CH}) torch eager fp32 on the device: forward+explanation {B / te:.1f} images/s ({1e3 * te:.1f} ms), forward {B / tf:.1f} images/s")
