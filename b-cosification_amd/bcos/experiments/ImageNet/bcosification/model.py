"""get_model for B-cosified torchvision ResNets (reference bcos/experiments/ImageNet/bcosification/model.py:15-57).

Same `model_config` contract: name ('resnet18' | 'resnet50'), weights, last_layer_name, bcos_args, bcosify_args,
standard_changes.  Pretrained torchvision weights cannot be downloaded here: pass a state dict through
`model_config["state_dict"]` (torchvision key names) or load a B-cosified checkpoint into the result afterwards.
"""
import warnings

from torch import nn

from bcos.models.standard_models import BasicBlock, Bottleneck, ResNetBcos
from bcos.modules.pooling import use_hip_pools
from bcosify import BcosifyNetwork

__all__ = ["get_model"]

_SPECS = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet34": (BasicBlock, [3, 4, 6, 3]), "resnet50": (Bottleneck, [3, 4, 6, 3])}


def get_torch_model_modified(arch_name: str, model_config):
    block, layers = _SPECS[arch_name]
    model = ResNetBcos(block, layers)
    sd = model_config.get("state_dict")
    if sd is not None:
        model.load_state_dict(sd)
    elif model_config.get("weights"):
        warnings.warn(f"weights={model_config['weights']!r} requested but no network access: standard-model weights are "
                      "randomly initialised; pass model_config['state_dict'] or load a checkpoint afterwards")
    return model


def get_model(model_config) -> nn.Module:
    assert model_config.get("is_bcos", False), "Should be true!"
    model = BcosifyNetwork(get_torch_model_modified(model_config["name"], model_config), model_config,
                           add_channels=True, logit_layer=True)
    for k, v in (model_config.get("standard_changes") or {}).items():     # e.g. maxpool -> nn.AvgPool2d(3, 2, 1)
        setattr(model.model, k, v)
    for mod in model.modules():                                            # "Removing bias parameters (making None)"
        if hasattr(mod, "bias") and mod.bias is not None:
            mod.bias = None
    return use_hip_pools(model)
