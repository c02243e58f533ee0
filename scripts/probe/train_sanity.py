"""Sanity of the training plans over a few dozen steps (development aid): the BCE loss of a fixed synthetic batch must fall and stay finite,
through the plan and per layer alike.  usage: python scripts/probe/train_sanity.py [vit_ti|resnet18|resnet50] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "b-cosification_amd"))
import torch, torch.nn.functional as F
from bcos_hip import synth, lib
lib.load()
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_ti"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
def build(plan):
    if arch == "vit_ti":
        from bcos_hip import vit_engine as E
        net = synth.build_bcosified_vit(seed=0).to(dev)
    else:
        from bcos_hip import engine as E
        net = synth.build_bcosified_resnet(arch, seed=0).to(dev)
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8, seed=123).to(dev))
    if plan: E.attach(net)
    return net.train()
x = synth.synthetic_images(32, seed=1).to(dev)
tgt = F.one_hot(torch.arange(32) % 1000, 1000).float().to(dev)
for plan in (True, False):
    net = build(plan)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=float(os.environ.get("LR", "0.05")), momentum=0.9)
    losses = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        out = net(x)
        loss = F.binary_cross_entropy_with_logits(out, tgt)
        loss.backward(); opt.step()
        losses.append(float(loss))
    node = type(out.grad_fn).__name__
    print(f"{arch} plan={plan} ({node}): " + " ".join(f"{l:.5f}" for l in losses[::max(1, steps // 8)]) + f" -> {losses[-1]:.5f}")
    assert all(l == l and l < 1e3 for l in losses) and losses[-1] < losses[0]
