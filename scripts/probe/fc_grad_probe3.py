import sys, os, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from bcos_hip import synth, engine
DEV = "cuda"
gd = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(gd, "resnet14b_train_step.json")))
data = np.load(os.path.join(gd, "resnet14b_train_step.npz"))
def build():
    net = synth.build_bcosified_resnet("resnet14b", seed=meta["weight_seed"])
    synth.apply_calibration(net, {k: torch.from_numpy(data["calib/" + k]) for k in meta["calib_order"]})
    return net
x0 = synth.synthetic_images(4, seed=meta["image_seed"], size=meta["size"]).to(DEV)
target = F.one_hot(torch.tensor(meta["labels"]), 1000).float().to(DEV)
res = {}
for variant in ("A: as the checker", "B: no float(loss) sync", "C: no gx"):
    net = build().to(DEV)
    net.train()
    x = x0.clone().requires_grad_(True)
    out = net(x)
    loss = F.binary_cross_entropy_with_logits(out, target)
    if variant.startswith("A"):
        _ = float(loss.detach())
    named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    names = [n for n, _ in named]
    grads = torch.autograd.grad(loss, ([x] if not variant.startswith("C") else []) + [p for _, p in named])
    off = 0 if variant.startswith("C") else 1
    gfc = grads[off + names.index("model.fc.linear.weight")].detach().cpu().double().view(1000, 2048)
    res[variant] = gfc
    print(variant, "fc norm", float(gfc.norm()), "fixture", float(data["grad_norms"][names.index("model.fc.linear.weight")]))
a, c = res["A: as the checker"], res["C: no gx"]
d = (a - c).norm(dim=1)
top = torch.topk(d, 6).indices.tolist()
print("rows that differ between A and C:", [(r, float(a[r].norm()), float(c[r].norm())) for r in top])
print("columns: max |A - C| per column block of 128:", [float((a - c)[:, k:k + 128].abs().max()) for k in range(0, 2048, 128)])
