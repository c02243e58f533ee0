"""Experiment table of the B-cosified CLIP RN50 image encoder: names and `model` sections only (reference
bcos/experiments/ImageNet/clip_bcosification/experiment_parameters.py:41-109).
Names: resnet_50_clip_b2_noBias_randomResizedCrop[_cyclicLR]_sigLip_ImageNet_bcosification[-seed=N]."""
import copy

__all__ = ["CONFIGS"]

SEEDS = (420, 1337)


def _model(depth: int, sched: str) -> dict:
    return dict(
        is_bcos=True,
        name=f"resnet{depth}clip",
        weights="clip",
        bcos_args=dict(b=2, max_out=1),
        bcosify_args=dict(clip_kd=True, fix_b=True, norm_layer="BnUncV2", schDLR=sched, use_bias=False),
    )


CONFIGS = {}
for _sched in ("cosineAnnealingLR", "cyclicLR"):
    _name = "resnet_50_clip_b2_noBias_randomResizedCrop" + ("_cyclicLR" if _sched == "cyclicLR" else "") + \
            "_sigLip_ImageNet_bcosification"
    CONFIGS[_name] = dict(clip_kd=True, model=_model(50, _sched), seed=None)
for _name, _cfg in list(CONFIGS.items()):
    for _seed in SEEDS:
        _c = copy.deepcopy(_cfg)
        _c["seed"] = _seed
        CONFIGS[f"{_name}-seed={_seed}"] = _c
