"""Which stream bounds a ResNet-50 training step through the plan?  The step as it is, with the parameter gradients on the caller's stream
(BCOS_TRAIN_SIDE_STREAM=0 in the environment), and with the convolution weights frozen (no weight-gradient launches at all: the main
stream alone).  usage: python scripts/probe/train_critical_path_probe.py [arch]   (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "b-cosification_amd"))
import torch, torch.nn.functional as F
from bcos_hip import synth, lib, engine
lib.load()
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dev = torch.device("cuda", 0)
net = synth.build_bcosified_resnet(arch, seed=0).to(dev)
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8, seed=123).to(dev))
engine.attach(net); net.train()
x = synth.synthetic_images(64, seed=1).to(dev)
target = F.one_hot(torch.randint(0, 1000, (64,)), 1000).float().to(dev)

def measure(tag):
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9)
    def step():
        opt.zero_grad(set_to_none=True)
        F.binary_cross_entropy_with_logits(net(x), target).backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")

measure("as built (side stream: " + os.environ.get("BCOS_TRAIN_SIDE_STREAM", "1") + ")")
for m in net.modules():
    if isinstance(m, torch.nn.Conv2d):
        m.weight.requires_grad_(False)
measure("convolution weights frozen (no weight-gradient launches)")
