"""get_model for B-cosified SimpleViTs (reference bcos/experiments/ImageNet/vit_bcosification/model.py:7-31).
The reference fetches the standard model from torch.hub ("B-cos/B-cos-v2", standard_<arch>); without network access the
standard model is built locally (nn.Linear / nn.LayerNorm / nn.GELU, 3 channels) and, if given, loaded from
`model_config["state_dict"]`."""
from torch import nn

import bcos.models.vit as vit
from bcosify_vit import BcosifyNetwork

__all__ = ["get_model"]


def get_model(model_config):
    arch_name = model_config["name"]
    args = model_config["args"]
    # (the conv-stem archs take nn.Conv2d + a one-group nn.GroupNorm, the standard counterpart of DetachableGNLayerNorm2d)
    model = getattr(vit, arch_name)(channels=3, linear_layer=nn.Linear, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                                    conv2d_layer=nn.Conv2d, norm2d_layer=lambda c: nn.GroupNorm(1, c),
                                    num_classes=args.get("num_classes", 1000))
    if model_config.get("state_dict") is not None:
        model.load_state_dict(model_config["state_dict"])
    model = BcosifyNetwork(model, model_config, add_channels=True, logit_layer=model_config.get("logit_layer", False))
    if not model_config.get("bcosify_args", {}).get("use_bias", False):
        for mod in model.modules():
            if hasattr(mod, "bias") and mod.bias is not None:
                mod.bias = None
    if args.get("gap_reorder", False):
        model.model.gap_reorder = True
    return model
