"""Which module first produces a different output for the same input when calibration is repeated in one process?  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth
tag = sys.argv[1] if len(sys.argv) > 1 else "0"
reps = int(os.environ.get("REPS", "2"))
SYNC = os.environ.get("SYNC", "0") == "1"


def dig(t):
    return t.detach().double().abs().sum()          # stays on the device: no sync inside the pass


runs = []
for rep in range(reps):
    net = synth.build_bcosified_clip_rn50().to("cuda")
    log = []
    hooks = []
    for name, m in net.named_modules():
        if not list(m.children()):
            def hook(mod, args, out, name=name):
                if isinstance(out, torch.Tensor) and args and isinstance(args[0], torch.Tensor):
                    log.append((name, type(mod).__name__, dig(args[0]), dig(out)))
                    if SYNC:
                        torch.cuda.synchronize()
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    runs.append([(n, t, float(a), float(b)) for n, t, a, b in log])
bad = 0
for rep in range(1, reps):
    for i, ((n0, t0, a0, b0), (n1, t1, a1, b1)) in enumerate(zip(runs[0], runs[rep])):
        if (a0, b0) != (a1, b1):
            bad += 1
            print(f"[{tag}] rep {rep}: first difference at call {i} of {len(runs[0])}: {n0} ({t0}): input digest {a0!r} vs {a1!r}, "
                  f"output digest {b0!r} vs {b1!r}; previous call: {runs[0][i - 1][:2]}", flush=True)
            break
print(f"[{tag}] done: {bad} differing repeats", flush=True)
