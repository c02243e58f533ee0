import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from bcos_hip import synth, engine
from oracle import bcos_oracle as O
import test_gpu_parity as T
g = os.path.join(ROOT, "tests", "golden")
net, meta, data = T._golden_net(g, "resnet18_e2e")
x = synth.synthetic_images(8, seed=123).to("cuda")
eng = engine.attach(net)
out = eng.explain(x)
def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())
for i in range(8):
    print(i, "contrib rel", rel(out["contribution_map"][i], data["contribution_map"][i]), "norm", float(np.linalg.norm(data["contribution_map"][i])))
print("flips", T._gate_flips(net, eng, x, "resnet18"))
sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
ref = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, "resnet18", detach=detach), x.cpu())
print("oracle-on-this-host vs golden", rel(ref["contribution_map"], data["contribution_map"]), "hip vs oracle-here", rel(out["contribution_map"], ref["contribution_map"]))
pre = T._oracle_pre_activations(net, x, "resnet18")
_, st = eng._run_forward(x, keep=True)
ours = [st["t0"]] + [t for rec in st["blocks"] for t in rec["ts"]]
for li, (p, t) in enumerate(zip(pre, ours)):
    d = (p > 0) != (t.permute(0, 3, 1, 2).cpu() != 0)
    if d.any():
        idx = d.nonzero()
        print("layer", li, "flips", int(d.sum()), "pre", p[d][:5].tolist(), "rms", float(p.pow(2).mean().sqrt()), "img", idx[:5, 0].tolist())
