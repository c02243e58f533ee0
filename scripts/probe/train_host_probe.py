"""Host issue time against device time of a training step through the plans (is the step launch-bound?), and where the Python time goes.
usage: python scripts/probe/train_host_probe.py [vit_ti|resnet50|...]   (on the GPU box)"""
import cProfile, pstats, sys, time, os
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
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record(); step(); e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"issue {1e3*(t1-t0):.2f} ms   device {e0.elapsed_time(e1):.2f} ms   wall {1e3*(t2-t0):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
