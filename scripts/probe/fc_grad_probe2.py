import sys, os, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd"))
import numpy as np, torch, torch.nn.functional as F
from bcos_hip import synth, engine
DEV = "cuda"
def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))
gd = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(gd, "resnet14b_train_step.json")))
data = np.load(os.path.join(gd, "resnet14b_train_step.npz"))
for path in ("layers", "plan"):
    net = synth.build_bcosified_resnet("resnet14b", seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    net = net.to(DEV)
    if path == "plan":
        engine.attach(net)
    x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV).requires_grad_(True)
    target = F.one_hot(torch.tensor(meta["labels"]), 1000).float().to(DEV)
    net.train()
    logits = net(x)
    print(path, "logits", rel(logits, data["output"]))
    loss = F.binary_cross_entropy_with_logits(logits, target)
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    names = [n for n, _ in named]
    grads = torch.autograd.grad(loss, [x] + [p for _, p in named])
    gfc = grads[1 + names.index("model.fc.linear.weight")].view(1000, 2048)
    ref = torch.from_numpy(data["gradrows/model.fc.linear.weight"]).view(-1, 2048)
    print(path, "fc rows rel", rel(gfc[:ref.shape[0]], ref))
    for r in (0, 1, 5, 11, 20):
        print("   row", r, "norm ours", float(gfc[r].norm()), "ref", float(ref[r].norm()), "rel", rel(gfc[r], ref[r]))
    # the logits' own rows: which images drive row 0 / 11
    lg = logits.detach().cpu()
    print("   logits at labels", [float(lg[i, l]) for i, l in enumerate(meta["labels"])], "ref", [float(data["output"][i, l]) for i, l in enumerate(meta["labels"])])
    if path == "plan":
        engine.detach(net)

# full-row comparison against the CPU emulation of the same path (tests/cpu_emulation.py: held to the reference fixture at 1e-4 on CPU)
sys.path.insert(0, os.path.join(ROOT, "tests"))
net = synth.build_bcosified_resnet("resnet14b", seed=meta["weight_seed"])
synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
net = net.to(DEV)
x = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV).requires_grad_(True)
target = F.one_hot(torch.tensor(meta["labels"]), 1000).float().to(DEV)
net.train()
loss = F.binary_cross_entropy_with_logits(net(x), target)
named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
names = [n for n, _ in named]
gdev = torch.autograd.grad(loss, [p for _, p in named])[names.index("model.fc.linear.weight")].view(1000, 2048).cpu()
import cpu_emulation
class MP:
    def setattr(self, obj, name, val, raising=True): setattr(obj, name, val)
    def setitem(self, d, k, v): d[k] = v
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=True): os.environ.pop(k, None)
cpu_emulation.install(MP())
netc = synth.build_bcosified_resnet("resnet14b", seed=meta["weight_seed"])
synth.apply_calibration(netc, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
xc = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).requires_grad_(True)
netc.train()
lossc = F.binary_cross_entropy_with_logits(netc(xc), target.cpu())
gcpu = torch.autograd.grad(lossc, [p for _, p in netc.named_parameters() if p.requires_grad])[names.index("model.fc.linear.weight")].view(1000, 2048)
print("emulated norm", float(gcpu.norm()), "device norm", float(gdev.norm()), "fixture", float(data["grad_norms"][names.index("model.fc.linear.weight")]))
rn = (gdev - gcpu).norm(dim=1) / gcpu.norm(dim=1).clamp_min(1e-30)
top = torch.topk(rn * gcpu.norm(dim=1), 8).indices.tolist()
for r in top:
    print("   row", r, "dev", float(gdev[r].norm()), "cpu", float(gcpu[r].norm()), "rel", float(rn[r]))
