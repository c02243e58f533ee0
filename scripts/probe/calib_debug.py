import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "b-cosification_amd")); sys.path.insert(0, ROOT)
import torch
from bcos_hip import synth, ops
tag = sys.argv[1]
for rnd in range(3):
    net = synth.build_bcosified_resnet("resnet50").to("cuda")
    with torch.no_grad():
        rec = synth.calibrate(net, synth.synthetic_images(8).to("cuda"))
    sd = net.state_dict()
    keys = ["model.conv1.linear.weight", "model.bn1.running_var", "model.layer1.0.conv1.linear.weight", "model.layer1.0.bn1.running_var", "model.layer4.2.bn3.running_var"]
    print(tag, rnd, [f"{float(sd[k].double().abs().sum()):.17g}" for k in keys], flush=True)
    del net
