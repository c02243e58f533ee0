"""layer3_stress.py with the tensors around block 4's conv3 / bn3 kept: on a differing repetition, say what differs.  (development aid)"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
iters = int(os.environ.get("ITERS", "40"))
net = synth.build_bcosified_clip_rn50().to("cuda")
layer3 = net.model.layer3
state0 = copy.deepcopy(layer3.state_dict())
grab = {}
h = layer3.register_forward_pre_hook(lambda m, args: grab.setdefault("x", args[0].detach().clone()))
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
h.remove()
x3 = grab["x"]
blocks = list(layer3.children())
ref, bad = None, 0
for it in range(iters):
    layer3.load_state_dict(state0)
    rec = []
    hooks = []
    for bi, blk in enumerate(blocks):
        hooks.append(blk.conv3.register_forward_hook(lambda m, a, o, bi=bi: rec.append((f"{bi}.conv3", a[0].detach().clone(), m.linear.weight.detach().clone(), o.detach().clone()))))
        hooks.append(blk.bn3.register_forward_hook(lambda m, a, o, bi=bi: rec.append((f"{bi}.bn3", a[0].detach().clone(), m.running_var.detach().clone(), o.detach().clone()))))
    with torch.no_grad():
        synth.calibrate(layer3, x3)
    for hk in hooks:
        hk.remove()
    torch.cuda.synchronize()
    if ref is None:
        ref = rec
        continue
    for (n0, a0, w0, o0), (n1, a1, w1, o1) in zip(ref, rec):
        same = (torch.equal(a0, a1), torch.equal(w0, w1), torch.equal(o0, o1))
        if not all(same):
            bad += 1
            t0, t1 = (a0, a1) if not same[0] else ((w0, w1) if not same[1] else (o0, o1))
            d = (t0 - t1).abs()
            nz = d > 0
            msg = f"[{tag}] iter {it} first difference at {n0}: input same {same[0]}, weight/var same {same[1]}, output same {same[2]}; " \
                  f"{int(nz.sum())} of {d.numel()} elements differ, max {float(d.max()):.3e}"
            if t0.dim() == 4:
                pix = nz.any(1).flatten().nonzero().flatten()
                ch = nz.any(0).any(-1).any(-1).nonzero().flatten()
                msg += f"; pixels {pix[:4].tolist()}..{pix[-2:].tolist()} ({pix.numel()}), channels {ch[:8].tolist()}..{ch[-2:].tolist()} ({ch.numel()})"
                msg += f"; recomputed var equal to hooked var: n/a"
            else:
                idx = nz.nonzero().flatten()
                msg += f"; indices {idx[:8].tolist()} ({idx.numel()}); values {t0[idx[:3]].tolist()} vs {t1[idx[:3]].tolist()}"
                # is the variance of the (identical?) input reproducible now?
                v_again = a1.var(dim=(0, 2, 3), unbiased=False)
                msg += f"; var(input) recomputed equals ref var {bool(torch.equal(v_again, w0))}, equals this var {bool(torch.equal(v_again, w1))}"
            print(msg, flush=True)
            break
print(f"[{tag}] done: {bad} of {iters - 1} repetitions differ", flush=True)
