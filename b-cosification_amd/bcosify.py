"""B-cosification of convolutional networks (ResNets, CLIP's ModifiedResNet) -- the conversion surface of
shrebox/B-cosification's `bcosify.py:22-114`, producing HIP-backed layers.

    model = BcosifyNetwork(torch_resnet, model_config, add_channels=True, logit_layer=True)

walks the module tree once and swaps
    nn.Conv2d        -> BcosifyConv2d            nn.Linear        -> BcosifyLinear (except q/k/v_proj)
    last `fc`        -> 1x1 BcosifyConv2d (applied before the global average pool when `gap`)
    nn.BatchNorm2d   -> BatchNormUncentered2d    nn.Sequential    -> BcosSequential
    CLIP AttentionPool2d -> BcosAttentionPool2d  (clip_kd)
after widening the 3-channel stem to the 6-channel (r,g,b,1-r,1-g,1-b) encoding.  The `model_config` keys
are the reference's (SURVEY.md section 5 "Config / flags").
"""
import math
import warnings

import torch
import torch.nn as nn

from bcos.common import BcosUtilMixin
from bcos.modules import BcosSequential, LogitLayer
from bcos.modules.bcosifyconv2d import BcosifyConv2d
from bcos.modules.bcosifylinear import BcosifyLinear
from bcos.modules.norms.uncentered_norms import BatchNormUncentered2d

IMAGENET_MEAN_ADDINVERSE = (0.485, 0.456, 0.406, 0.515, 0.544, 0.594)
IMAGENET_STD_ADDINVERSE = (0.229, 0.224, 0.225, 0.229, 0.224, 0.225)

CLIP_MEAN_ADDINVERSE = (0.48145466, 0.4578275, 0.40821073, 0.51854534, 0.5421725, 0.59178927)
CLIP_MEAN_ZERO = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
CLIP_STD_ADDINVERSE = (0.26862954, 0.26130258, 0.27577711, 0.26862954, 0.26130258, 0.27577711)


class Normalize6(nn.Module):
    """Per-channel (x - mean) / std on [..., C, H, W] -- what the reference obtains from
    torchvision.transforms.Normalize (bcosify.py:38-43).  Kept as `mean` / `std` attributes (tuples) like
    torchvision's class; the fused engine folds it into its input-preparation kernel."""

    def __init__(self, mean, std):
        super().__init__()
        self.mean = tuple(mean)
        self.std = tuple(std)

    def forward(self, x):
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - mean) / std

    def __repr__(self):
        return f"Normalize(mean={self.mean}, std={self.std})"


def select_normalization(model_config):
    """mean / std of the 0-th layer as chosen at bcosify.py:33-43."""
    clip_kd = model_config["bcosify_args"].get("clip_kd", None)
    mean_zero = model_config.get("bfy_mean_zero", False)
    linearprobe = model_config["bcosify_args"].get("linearprobe_clip", False)
    if clip_kd and mean_zero:
        return CLIP_MEAN_ZERO, CLIP_STD_ADDINVERSE
    if (clip_kd or linearprobe) and not mean_zero:
        return CLIP_MEAN_ADDINVERSE, CLIP_STD_ADDINVERSE
    return IMAGENET_MEAN_ADDINVERSE, IMAGENET_STD_ADDINVERSE


class BcosifyNetwork(BcosUtilMixin, nn.Module):
    def __init__(self, model, model_config, add_channels=True, logit_layer=False):
        super().__init__()
        self.model = model
        self.model_config = model_config
        self.logit_layer = None
        if logit_layer:
            self.logit_layer = LogitLayer(logit_temperature=None, logit_bias=-math.log(1000 - 1))
        self.clip_kd = model_config["bcosify_args"].get("clip_kd", None)
        self.bfy_mean_zero = model_config.get("bfy_mean_zero", False)
        self.linearprobe_clip = model_config["bcosify_args"].get("linearprobe_clip", False)
        mean, std = select_normalization(model_config)
        self.bcosifynormalize = Normalize6(mean=mean, std=std)
        if add_channels:
            BcosifyNetwork.add_channels(self.model)
        BcosifyNetwork.bcosify(self.model, self.model_config)

    def forward(self, x):
        engine = getattr(self, "_bcos_engine", None)
        if engine is not None and not torch.is_grad_enabled() and not self.training:
            return engine.forward(x)          # eval + no_grad: the fused plan (it re-reads parameters that changed)
        if engine is not None and self.training and torch.is_grad_enabled():
            # train() + autograd: the whole network as ONE autograd node over the engine's layer list (bcos_hip/train_plan.py);
            # networks outside that plan's scope (attention-pool heads, grouped / MaxOut layers, learnable B) return None here
            from bcos_hip import train_plan
            out = train_plan.train_forward(engine, x)
            if out is not None:
                return out
        out = self.model(self.bcosifynormalize(x))
        return self.logit_layer(out) if self.logit_layer else out

    @classmethod
    def add_channels(cls, model):
        """Every 3-channel conv gets W <- cat(W, -W) / 2 and in_channels = 6 (reference :55-72)."""
        found = False
        for module in model.modules():
            if isinstance(module, nn.Conv2d) and module.in_channels == 3:
                if found:
                    warnings.warn("Found multiple layers with 3 input channels. "
                                  "Bcosification might thus not work as intended.")
                found = True
                module.in_channels = 6
                module.weight.data = torch.cat((module.weight.data, -module.weight.data), dim=1) / 2
        if not found:
            warnings.warn("No conv layer with 3 input channels was found. However, 'add_channels' was set to True."
                          "Bcosification might thus not work as intended.")

    @classmethod
    def bcosify(cls, model, model_config):
        args = model_config.get("bcosify_args", None) or {}
        clip_kd = args.get("clip_kd", False)
        norm_layer = args.get("norm_layer", "BnUncV2")
        gap = args.get("gap", True)
        act_layer = args.get("act_layer", True)
        last_layer_name = model_config.get("last_layer_name", "NoLastLayerName")
        for name, child in list(model.named_children()):
            if len(list(child.children())) > 0:
                if clip_kd and name == "attnpool" and _is_clip_attnpool(child):
                    from bcos.modules.bcosattnpool import BcosAttentionPool2d
                    setattr(model, name, BcosAttentionPool2d.from_standard_module(model, child, model_config))
                    cls.bcosify(model.attnpool, model_config)   # c_proj (and v_proj keys) inside the new pool
                else:
                    cls.bcosify(child, model_config)
            if isinstance(child, nn.Conv2d):
                setattr(model, name, BcosifyConv2d.from_standard_module(child, model_config))
            elif isinstance(child, nn.Linear) and (name != last_layer_name or clip_kd or not gap):
                if name not in ("k_proj", "v_proj", "q_proj"):
                    setattr(model, name, BcosifyLinear.from_standard_module(child, model_config))
            elif isinstance(child, nn.Linear) and name == last_layer_name and gap:
                setattr(model, name, BcosifyConv2d.from_standard_module_linear(child, model_config))
            elif isinstance(child, nn.Sequential):
                setattr(model, name, BcosSequential.from_standard_module(child))
            elif isinstance(child, nn.BatchNorm2d) and norm_layer in ("BnUnc2d", "BnUncV2"):
                setattr(model, name, BatchNormUncentered2d.from_standard_module(child, model_config))
            if isinstance(child, nn.ReLU) and not act_layer:
                setattr(model, name, nn.Identity())


def _is_clip_attnpool(module) -> bool:
    try:
        from CLIP.clip.model import AttentionPool2d
    except Exception:
        return False
    return isinstance(module, AttentionPool2d)
