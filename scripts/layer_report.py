"""Per-launch report of one forward+explanation step (development aid): geometry, time, TFLOP/s, GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, synth, engine

B = int(os.environ.get("B", "256"))
arch = os.environ.get("ARCH", "resnet50")
dev = "cuda"
net = synth.build_bcosified_resnet(arch).to(dev)
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to(dev))
eng = engine.attach(net)
x = synth.synthetic_images(B, seed=1000).to(dev)
records = []
orig = ops.tapconv
def hooked(a, wt, geom, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(a, wt, geom, **kw); e1.record()
    g = geom
    M = g["N"] * g["P"] * g["Q"]; K = g["TH"] * g["TW"] * g["C"]; Nn = g["Cout"]
    nbytes = 4 * (M * Nn * sum(1 for k in ("out", "out2", "scale_out", "addend", "mul", "mul2", "gate2") if kw.get(k) is not None))
    nbytes += 4 * g["N"] * g["H"] * g["W"] * g["C"] + 4 * K * Nn
    records.append((e0, e1, M, K, Nn, g["TH"], g["in_sh"], g["out_sh"], nbytes, "fwd" if kw.get("bcos_mode", 0) else "bwd", 2.0 * M * K * Nn))
ops.tapconv = hooked
orig_group = ops.tapconv_group
def hooked_group(a, wts, geoms, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig_group(a, wts, geoms, **kw); e1.record()
    g = geoms[0]
    M = sum(q["N"] * q["P"] * q["Q"] for q in geoms); K = max(q["TH"] * q["TW"] * q["C"] for q in geoms); Nn = g["Cout"]
    fl = sum(2.0 * q["N"] * q["P"] * q["Q"] * q["TH"] * q["TW"] * q["C"] * q["Cout"] for q in geoms)
    nbytes = 4 * (g["N"] * g["H"] * g["W"] * g["C"] + M * Nn)
    records.append((e0, e1, M, K, Nn, 0, 1, g["out_sh"], nbytes, "bwd", fl))
ops.tapconv_group = hooked_group
for _ in range(2):
    records.clear()
    eng.explain(x)
torch.cuda.synchronize()
tot = 0; totf = 0
agg = {}
for (e0, e1, M, K, Nn, TH, ins, outs, nbytes, kind, fl) in records:
    ms = e0.elapsed_time(e1)
    key = (kind, M, K, Nn, TH, ins, outs)
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl; a[3] += nbytes
    tot += ms; totf += fl
rows_csv = []
print(f"{'kind':4} {'M':>8} {'K':>6} {'N':>5} k s/os  cnt    ms     TF/s    GB/s   %time")
for key, (cnt, ms, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, M, K, Nn, TH, ins, outs = key
    print(f"{kind:4} {M:8d} {K:6d} {Nn:5d} {TH} {ins}/{outs} {cnt:4d} {ms:7.3f} {fl/ms/1e9:7.1f} {nb/ms/1e6:7.0f} {100*ms/tot:6.1f}")
    tf, gbs = fl / ms / 1e9, nb / ms / 1e6
    bound = "hbm" if fl / max(nb, 1) < 50.0 else "mfma"       # same rule as bench.py: algorithmic FLOP/B against ~50
    rows_csv.append(dict(kind=kind, M=M, K=K, N=Nn, taps_h=TH, in_stride=ins, out_stride=outs, launches=cnt, ms=round(ms, 4),
                         tflops=round(tf, 1), gbps=round(gbs, 0), bound=bound, frac_of_fp32_mfma_peak=round(tf / 157.3, 3),
                         hbm_frac_of_8TBps=round(gbs / 8000.0, 3), pct_of_contraction_time=round(100 * ms / tot, 2)))
if os.environ.get("LAYERS_CSV"):
    import csv
    with open(os.environ["LAYERS_CSV"], "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows_csv[0].keys()))
        wr.writeheader()
        wr.writerows(rows_csv)
print(f"total tapconv {tot:.2f} ms, {totf/1e9:.0f} GFLOP executed ({totf/tot/1e9:.1f} TF/s executed); algorithmic {17.22*B:.0f} GFLOP -> {17.22*B/tot:.1f} TF/s")
