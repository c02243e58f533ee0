"""A whole training step (forward, BCE loss, backward through the plan, SGD update) captured into ONE hipGraph and replayed -- torch's
whole-network capture recipe (static input / target buffers, warm-up on a side stream, optimizer with capturable state) -- against the
same step issued eagerly.  ViT-Ti training is host-bound (issue 10.6 of 11.0 ms, scripts/probe/train_host_probe.py): what does the
replay return?  usage: python scripts/probe/train_graph_probe.py [vit_ti|resnet18|resnet50]   (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "b-cosification_amd"))
import torch, torch.nn.functional as F
from bcos_hip import synth, lib
lib.load()
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_ti"
dev = torch.device("cuda", 0)
if arch == "vit_ti":
    from bcos_hip import vit_engine
    net = synth.build_bcosified_vit(seed=0).to(dev); attach = vit_engine.attach
else:
    from bcos_hip import engine
    net = synth.build_bcosified_resnet(arch, seed=0).to(dev); attach = engine.attach
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8, seed=123).to(dev))
attach(net); net.train()
B = 64
x = synth.synthetic_images(B, seed=1).to(dev)
target = F.one_hot(torch.randint(0, 1000, (B,)), 1000).float().to(dev)
params = [p for p in net.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-4, momentum=0.9)

def step():
    opt.zero_grad(set_to_none=True)
    loss = F.binary_cross_entropy_with_logits(net(x), target)
    loss.backward(); opt.step()
    return loss

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(3): step()
print(f"{arch} eager : {timed(step):.2f} ms per step")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss = F.binary_cross_entropy_with_logits(net(x), target)
    loss.backward(); opt.step()
l0 = float(loss)
print(f"{arch} graph : {timed(g.replay):.2f} ms per step   (captured loss {l0:.6f}, after the replays {float(loss):.6f})")
