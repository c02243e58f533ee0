"""B-cosification of SimpleViT models -- the conversion surface of the reference's `bcosify_vit.py:45-154`, producing
HIP-backed layers:  nn.Linear -> BcosifyLinear (except `to_qkv`), nn.LayerNorm -> DetachableLayerNorm,
nn.GELU -> MyGELU (gate detached in explanation mode), nn.Conv2d -> BcosifyConv2d, after widening the patch embedding to
the 6-channel (r,g,b,1-r,1-g,1-b) encoding with the per-pixel interleave required by the "(p1 p2 c)" patch flattening.
"""
import math
import warnings

import torch
import torch.nn as nn
from torch.autograd import Function

from bcos.common import BcosUtilMixin
from bcos.modules import LogitLayer, norms
from bcos.modules import _hipfn
from bcos.modules.bcosifyconv2d import BcosifyConv2d
from bcos.modules.bcosifylinear import BcosifyLinear
from bcos.modules.common import DetachableModule
from bcos_hip import ops
from bcosify import (CLIP_MEAN_ADDINVERSE, CLIP_MEAN_ZERO, CLIP_STD_ADDINVERSE, IMAGENET_MEAN_ADDINVERSE,  # noqa: F401
                     IMAGENET_STD_ADDINVERSE, Normalize6, select_normalization)


class _GeluFn(Function):
    @staticmethod
    def forward(ctx, x, detach):
        _hipfn.require_hip(x, "MyGELU")
        # a channels_last [N,C,H,W] activation (conv stems of the ViT-C models) is processed as its dense NHWC image
        ctx.cl = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
        xc = x.permute(0, 2, 3, 1) if ctx.cl else (x if x.is_contiguous() else x.contiguous())
        y, gate = ops.gelu_gate(xc, want_gate=ctx.needs_input_grad[0] and detach)
        ctx.detach_mode = detach
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(gate if detach else xc)
        return y.permute(0, 3, 1, 2) if ctx.cl else y

    @staticmethod
    def backward(ctx, gy):
        if not ctx.detach_mode:         # training mode: the gate is differentiated too, d/dx [x Phi(x)] = Phi(x) + x phi(x)
            (xc,) = ctx.saved_tensors
            if ctx.cl:
                return ops.gelu_bwd(gy.permute(0, 2, 3, 1).contiguous(), xc).permute(0, 3, 1, 2), None
            return ops.gelu_bwd(gy if gy.is_contiguous() else gy.contiguous(), xc), None
        (gate,) = ctx.saved_tensors
        if ctx.cl:
            return ops.mul(gy.permute(0, 2, 3, 1).contiguous(), gate).permute(0, 3, 1, 2), None
        return ops.mul(gy if gy.is_contiguous() else gy.contiguous(), gate), None


class MyGELU(DetachableModule):
    """x * Phi(x) with the gate Phi(x) = 0.5 (1 + erf(x / sqrt 2)) held constant in explanation mode
    (reference bcosify_vit.py:27-32).  One streaming HIP kernel; fused into the preceding B-cos linear by the engine."""

    def forward(self, x):
        return _GeluFn.apply(x, self.detach)


class BcosifyNormLayer(BcosUtilMixin, nn.Module):
    """Normalisation-only wrapper (reference :36-43)."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.bcosifynormalize = Normalize6(mean=IMAGENET_MEAN_ADDINVERSE, std=IMAGENET_STD_ADDINVERSE)

    def forward(self, x):
        return self.model(self.bcosifynormalize(x))


class BcosifyNetwork(BcosUtilMixin, nn.Module):
    def __init__(self, model, model_config, add_channels=True, logit_layer=False):
        super().__init__()
        self.model = model
        self.model_config = model_config
        self.logit_layer = None
        if logit_layer:
            self.logit_bias = model_config.get("logit_bias", -math.log(1000 - 1))
            self.logit_temperature = model_config.get("logit_temperature", None)
            self.logit_layer = LogitLayer(logit_temperature=self.logit_temperature, logit_bias=self.logit_bias)
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
            # train() + autograd: the whole network as ONE autograd node over the engine's block list (bcos_hip/vit_train_plan.py);
            # conv-stem models, MaxOut / unit-norm / learnable-B layers and modules in explanation mode return None here
            from bcos_hip import vit_train_plan
            out = vit_train_plan.train_forward(engine, x)
            if out is not None:
                return out
        out = self.model(self.bcosifynormalize(x))
        return self.logit_layer(out) if self.logit_layer else out

    @classmethod
    def add_channels(cls, model):
        """ViT-C: the first stem conv gets W <- cat(W, -W)/2.  Plain ViT: the patch-embedding linear [out, p*p*3] is
        widened to [out, p*p*6] with every pixel's (r,g,b) weights followed by their negatives, both halved, matching
        the "(p1 p2 c)" flattening of 6-channel pixels (reference :94-121)."""
        for name, module in model.named_modules():
            if name == "to_patch_embedding.conv_stem.0":
                module.in_channels = 6
                module.weight.data = torch.cat((module.weight.data, -module.weight.data), dim=1) / 2
                return
            if name == "to_patch_embedding.linear":
                module.in_features *= 2
                w = module.weight.data.view(module.out_features, -1, 3) / 2
                module.weight.data = torch.cat([w, -w], dim=2).reshape(module.out_features, module.in_features)
                return
        warnings.warn("No linear layer was found. Bcosification might thus not work as intended.")

    @classmethod
    def bcosify(cls, model, model_config):
        act_layer = model_config.get("act_layer", True)
        for name, child in list(model.named_children()):
            if len(list(child.children())) > 0:
                cls.bcosify(child, model_config)
            if isinstance(child, nn.Conv2d):
                setattr(model, name, BcosifyConv2d.from_standard_module(child, model_config))
            elif isinstance(child, nn.Linear):
                if name != "to_qkv":
                    setattr(model, name, BcosifyLinear.from_standard_module(child, model_config))
            elif isinstance(child, nn.GELU):
                setattr(model, name, MyGELU() if act_layer else nn.Identity())
            elif isinstance(child, nn.LayerNorm):
                setattr(model, name, norms.DetachableLayerNorm.from_standard_module(child, model_config))
            elif isinstance(child, nn.GroupNorm):
                setattr(model, name, norms.DetachableGroupNorm2d.from_standard_module(child, model_config))
