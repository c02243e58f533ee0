import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bcos_hip import clip_head, engine, synth
from oracle import bcos_oracle as O
G = os.path.join(ROOT, "tests", "golden")
meta = json.load(open(os.path.join(G, "clip_zeroshot_attr.json")))
calib = np.load(os.path.join(G, "clip_rn50.npz")); cmeta = json.load(open(os.path.join(G, "clip_rn50.json")))
record = {k: torch.from_numpy(calib["calib/" + k]) for k in cmeta["calib_order"]}
x = synth.synthetic_images(4, seed=123).cuda()
wt = torch.randn(1024, 16, generator=torch.Generator().manual_seed(99)); w1 = (wt[:, 3:4] / wt[:, 3:4].norm()).cuda()
net = synth.build_bcosified_clip_rn50(attn_unpool=True); synth.apply_calibration(net, record); net = net.cuda()
eng = engine.attach(net)
sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
log = []
with torch.no_grad(): O.clip_rn50_embed(sd, x[:1].cpu(), detach=True, attn_unpool=True, gate_log=log)
gates = [(p > 0).float().permute(0, 2, 3, 1).contiguous().cuda() for p in log]
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
for pc, nm in ((1, False), (2, False), (0, False), (2, True)):
    gr, vr = O.zeroshot_attribution(lambda xx, detach: O.clip_rn50_embed(sd, xx, detach=detach, attn_unpool=True), x[:1].cpu(), w1.cpu(), attn_unpool=True, pool_cosine=pc, norm_max_cosine=nm)
    pu = clip_head.zeroshot_attribution(eng, x[:1], w1, pool_cosine=pc, norm_max_cosine=nm, gates=[g.clone() for g in gates])
    print(os.environ.get("BCOS_CONTRACTION", "default"), pc, nm, "W", rel(pu["dynamic_linear_weights"], gr), "val", rel(pu["logits"].view(-1), vr))
