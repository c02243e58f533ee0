"""get_model for the B-cosified CLIP RN50 image encoder (reference
bcos/experiments/ImageNet/clip_bcosification/model.py:8-25).  `clip.load("RN50")` needs a download; here the vision
tower is built locally (and loaded from `model_config["state_dict"]`, OpenAI `visual.*` key names, if given)."""
from torch import nn

from bcos.modules.pooling import use_hip_pools
from bcosify import BcosifyNetwork
from CLIP.clip.model import ModifiedResNet

__all__ = ["get_model"]


def get_model(model_config) -> nn.Module:
    assert model_config.get("is_bcos", False), "Should be true!"
    visual = ModifiedResNet([3, 4, 6, 3], 1024, 32, input_resolution=224, width=64).float()
    if model_config.get("state_dict") is not None:
        visual.load_state_dict(model_config["state_dict"])
    model = BcosifyNetwork(visual, model_config, add_channels=True, logit_layer=False)
    for mod in model.modules():
        if hasattr(mod, "bias") and mod.bias is not None:
            mod.bias = None
        if hasattr(mod, "positional_embedding") and mod.positional_embedding is not None:
            mod.positional_embedding = None
    return use_hip_pools(model)
