"""Follow-up of calibrate_stress.py: keep every output of the layer3 conv3 modules (and the input of the second call) during
calibration and report where two calibrations of the same process first differ.  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
reps = int(os.environ.get("REPS", "2"))
runs = []
for rep in range(reps):
    net = synth.build_bcosified_clip_rn50().to("cuda")
    log = []
    hooks = []
    for name, m in net.named_modules():
        if name.startswith("model.layer3.") and name.endswith("conv3"):
            hooks.append(m.register_forward_hook(lambda mod, args, out, name=name: log.append((name, args[0].detach().clone(), out.detach().clone(),
                                                                                             mod.linear.weight.detach().clone()))))
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
    for h in hooks:
        h.remove()
    runs.append(log)
    torch.cuda.synchronize()
bad = 0
for rep in range(1, reps):
    for (n0, x0, y0, w0), (n1, x1, y1, w1) in zip(runs[0], runs[rep]):
        same = (torch.equal(x0, x1), torch.equal(w0, w1), torch.equal(y0, y1))
        if not all(same):
            bad += 1
            d = (y0 - y1).abs()
            nz = (d > 0)
            # y is NCHW-shaped channels_last: report which pixels / channels are touched
            pix = nz.any(1).flatten().nonzero().flatten()
            ch = nz.any(0).any(-1).any(-1).nonzero().flatten()
            print(f"[{tag}] rep {rep} {n0}: input same {same[0]}, weight same {same[1]}, output same {same[2]}; {int(nz.sum())} of {d.numel()} "
                  f"elements differ (max {float(d.max()):.3e}); pixels {pix[:6].tolist()}..{pix[-3:].tolist()} ({pix.numel()}), "
                  f"channels {ch[:6].tolist()}..{ch[-3:].tolist()} ({ch.numel()})", flush=True)
            break
print(f"[{tag}] done: {bad} differing repeats", flush=True)
