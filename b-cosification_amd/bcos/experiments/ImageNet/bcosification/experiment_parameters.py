"""Experiment table of the B-cosified torchvision ResNets: names and `model` sections only (the inference-side part
of the reference's bcos/experiments/ImageNet/bcosification/experiment_parameters.py:41-131; data, optimiser,
schedule and criterion entries belong to training and are not restated).  Names: resnet_{18,50}[ _V1][-seed=N]."""
import copy
import math

from torch import nn

__all__ = ["CONFIGS"]

NUM_CLASSES = 1000
SEEDS = (5, 420, 1337)


def _model(depth: int, weight: str) -> dict:
    if weight == "V2":
        weights = f"ResNet{depth}_Weights.DEFAULT"
    else:
        weights = "IMAGENET1K_V1" if depth == 50 else None
    return dict(
        is_bcos=True,
        name=f"resnet{depth}",
        last_layer_name="fc",                      # the fc layer becomes a 1x1 B-cos conv in front of the global pool
        weights=weights,
        args=dict(num_classes=NUM_CLASSES, norm_layer=None, logit_bias=-math.log(NUM_CLASSES - 1)),
        bcos_args=dict(b=2, max_out=1),
        bcosify_args=dict(fix_b=True, use_bias=False, norm_layer="BnUncV2", manual_optim=False, gap=True, act_layer=True),
        standard_changes={"maxpool": nn.AvgPool2d(kernel_size=3, stride=2, padding=1)},
    )


CONFIGS = {}
for _depth in (18, 50):
    for _weight in ("V2", "V1"):
        CONFIGS[f"resnet_{_depth}" + ("_V1" if _weight == "V1" else "")] = dict(model=_model(_depth, _weight), seed=None)
for _name, _cfg in list(CONFIGS.items()):
    for _seed in SEEDS:
        _c = copy.deepcopy(_cfg)
        _c["seed"] = _seed
        CONFIGS[f"{_name}-seed={_seed}"] = _c
