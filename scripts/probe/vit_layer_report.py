"""Per-launch report of one ViT-Ti forward+explanation step (development aid): geometry, epilogue tensors, time, TFLOP/s, GB/s.
Run with BCOS_VIT_SUBBATCH_STREAMS=1 (an event pair must time its own launch only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import ops, synth, vit_engine
B = int(os.environ.get("B", "512"))
net = synth.build_bcosified_vit(seed=0).to("cuda")
with torch.no_grad():
    synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
eng = vit_engine.attach(net)
x = synth.synthetic_images(B, seed=1000).to("cuda")
records = []
orig = ops.tapconv
def hooked(a, wt, geom, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(a, wt, geom, **kw); e1.record()
    g = geom
    M = g["N"] * g["P"] * g["Q"]; K = g["TH"] * g["TW"] * g["C"]; Nn = g["Cout"]
    epi = tuple(k for k in ("out", "out2", "scale_out", "addend", "mul", "mul2", "row_scale", "a_sumsq", "bias") if kw.get(k) is not None)
    nbytes = 4 * (M * Nn * sum(1 for k in ("out", "out2", "scale_out", "addend", "mul", "mul2") if kw.get(k) is not None) + M * K + K * Nn)
    records.append((e0, e1, M, K, Nn, "fwd" if kw.get("bcos_mode", 0) else "lin", int(kw.get("relu", 0) or 0), epi, nbytes, 2.0 * M * K * Nn))
ops.tapconv = hooked
for _ in range(2):
    records.clear()
    eng.explain(x)
torch.cuda.synchronize()
agg = {}
tot = 0
for (e0, e1, M, K, Nn, kind, relu, epi, nb, fl) in records:
    ms = e0.elapsed_time(e1)
    a = agg.setdefault((kind, M, K, Nn, relu, epi), [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl; a[3] += nb
    tot += ms
print(f"{'kind':4} {'M':>7} {'K':>5} {'N':>5} act cnt     ms    us/launch  TF/s   GB/s  epilogue tensors")
for key, (cnt, ms, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, M, K, Nn, relu, epi = key
    print(f"{kind:4} {M:7d} {K:5d} {Nn:5d}  {relu}  {cnt:3d} {ms:7.3f} {1e3 * ms / cnt:9.1f} {fl / ms / 1e9:7.1f} {nb / ms / 1e6:6.0f}  {','.join(epi)}")
print(f"total contraction {tot:.2f} ms over {len(records)} launches")
