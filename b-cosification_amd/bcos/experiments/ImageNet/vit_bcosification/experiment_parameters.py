"""Experiment table of the B-cosified SimpleViTs: names and `model` sections only (reference
bcos/experiments/ImageNet/vit_bcosification/experiment_parameters.py:108-227).  The reference derives the "bcosifyv2"
entries from its "bcos_<arch>" entries, so their names carry that prefix:
    bcosifyv2_bcos_<arch>[_random][_<lr>][_lrWarmup][_useBias][_noGelu][_gapReorder][-seed=N]
Only the flags that reach the model (`random`, `useBias`, `noGelu`, `gapReorder`) change the section built here; the
learning-rate flags select training settings and only vary the name."""
import copy
import itertools
import math

__all__ = ["CONFIGS", "SIMPLE_VIT_ARCHS"]

NUM_CLASSES = 1000
SEEDS = (5, 420, 1337)
# plain and conv-stem SimpleViTs, as in the reference table (the conv stems run on BcosifyConv2d + DetachableGroupNorm2d)
SIMPLE_VIT_ARCHS = ["simple_vit_ti_patch16_224", "simple_vit_s_patch16_224", "simple_vit_b_patch16_224",
                    "simple_vit_l_patch16_224", "vitc_ti_patch1_14", "vitc_s_patch1_14", "vitc_b_patch1_14",
                    "vitc_l_patch1_14"]


def _model(arch: str, weight: str, use_bias: bool, gelu: bool, gap_reorder: bool) -> dict:
    return dict(
        is_bcos=True,
        name=arch,
        weights="pretrained" if weight == "pretrained" else None,
        args=dict(num_classes=NUM_CLASSES, channels=6, gap_reorder=gap_reorder),
        bcos_args=dict(b=2, max_out=1),
        bcosify_args=dict(fix_b=True, use_bias=use_bias),
        logit_layer=True,
        act_layer=gelu,
        logit_bias=math.log(1 / (NUM_CLASSES - 1)),
    )


CONFIGS = {}
for _arch in SIMPLE_VIT_ARCHS:
    for _weight, _warm, _lr, _gelu, _bias, _gap in itertools.product(
            ("pretrained", "random"), ("lrWarmup", "noLrWarmup"), (1e-2, 1e-3, 1e-4, 1e-5), ("gelu", "noGelu"),
            ("useBias", "noBias"), ("gapReorder", "noGapReorder")):
        _name = (f"bcosifyv2_bcos_{_arch}" + ("_random" if _weight == "random" else "") +
                 (f"_{_lr}" if _lr in (1e-3, 1e-2, 1e-5) else "") + ("_lrWarmup" if _warm == "lrWarmup" else "") +
                 ("_useBias" if _bias == "useBias" else "") + ("_noGelu" if _gelu == "noGelu" else "") +
                 ("_gapReorder" if _gap == "gapReorder" else ""))
        CONFIGS[_name] = dict(model=_model(_arch, _weight, _bias == "useBias", _gelu == "gelu", _gap == "gapReorder"), seed=None)
for _name, _cfg in list(CONFIGS.items()):
    for _seed in SEEDS:
        _c = copy.deepcopy(_cfg)
        _c["seed"] = _seed
        CONFIGS[f"{_name}-seed={_seed}"] = _c
