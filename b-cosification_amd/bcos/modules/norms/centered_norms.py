"""Detachable (centred) norms for B-cos transformers on MI355X.

Reference: bcos/modules/norms/centered_norms.py -- `DetachableLayerNorm` :187-245 is the one the B-cosified ViTs use
(SURVEY.md a10).  In explanation mode its variance is a constant while the mean stays differentiable, so
    y = w (x - mean(x)) / std (+ b),    d y / d x  applied to g:   h - mean(h),  h = g w / std.
Forward and that input gradient are one-wavefront-per-row HIP kernels (bcos_layernorm_fwd / _bwd_detached).
The 2-D group-norm variants (`DetachableGroupNorm2d` :93-160 and its instance / layer-norm forms: the conv stems of the
ViT-C models) follow the same rule per (image, group) on NHWC data (bcos_groupnorm_fwd / _bwd_detached).
"""
import torch
import torch.nn as nn
from torch import Tensor
from torch.autograd import Function

from bcos_hip import ops

from bcos.modules import _hipfn
from bcos.modules.common import DetachableModule

__all__ = ["DetachableLayerNorm", "DetachableGroupNorm2d", "DetachableGNInstanceNorm2d", "DetachableGNLayerNorm2d"]


class _LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, detach):
        _hipfn.require_hip(x, "DetachableLayerNorm")
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w = weight.detach().contiguous() if weight is not None else None
        b = bias.detach().contiguous() if bias is not None else None
        need_x = ctx.needs_input_grad[0]
        need_wb = (weight is not None and ctx.needs_input_grad[1]) or (bias is not None and ctx.needs_input_grad[2])
        y, rstd = ops.layernorm_fwd(x2, w, b, eps, want_rstd=need_x or need_wb)
        ctx.detach_mode = detach
        ctx.w = w
        ctx.has_bias = bias is not None
        if rstd is not None:
            ctx.save_for_backward(rstd, *([x2] if not detach else []))
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        rstd = ctx.saved_tensors[0]
        g2 = gy.reshape(-1, gy.shape[-1])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        need_w = ctx.w is not None and ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        gw = gb = None
        if not ctx.detach_mode:
            # training mode (= F.layer_norm's gradient, centered_norms.py:200-202): nothing detached
            x2 = ctx.saved_tensors[1]
            gx, xhat = ops.layernorm_bwd(g2, x2, ctx.w, rstd, want_xhat=need_w)
            if need_w:
                gw = ops.colsum(g2, xhat) if g2.shape[1] % 4 == 0 else (g2 * xhat).sum(0)
            if need_b:
                gb = ops.colsum(g2) if g2.shape[1] % 4 == 0 else g2.sum(0)
            return gx.view(gy.shape), gw, gb, None, None
        # explanation mode: the input gradient only (the affine parameters receive gradients in training mode)
        gx, _ = ops.layernorm_bwd_detached(g2, ctx.w, rstd)
        return gx.view(gy.shape), None, None, None, None


class DetachableLayerNorm(nn.LayerNorm, DetachableModule):
    """nn.LayerNorm over the last dimension whose variance is detached in explanation mode (reference :187-245)."""

    def __init__(self, *args, **kwargs):
        DetachableModule.__init__(self)
        super().__init__(*args, **kwargs)

    def forward(self, input: Tensor) -> Tensor:
        if len(self.normalized_shape) != 1:
            raise NotImplementedError("the HIP LayerNorm normalises over the last dimension only")
        return _LayerNormFn.apply(input, self.weight, self.bias, self.eps, self.detach)

    @classmethod
    def from_standard_module(cls, standard_module: nn.LayerNorm, model_config: dict):
        new = cls(normalized_shape=standard_module.normalized_shape, eps=standard_module.eps,
                  elementwise_affine=standard_module.elementwise_affine)
        if model_config.get("weights", None) is not None:
            new.weight.data = standard_module.weight.data
            if standard_module.bias is not None:
                new.bias.data = standard_module.bias.data
        return new


class _GroupNormFn(Function):
    """GroupNorm on the NHWC image of a logical [N,C,H,W] tensor: one workgroup per (image, group)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, detach):
        _hipfn.require_hip(x, "DetachableGroupNorm2d")
        xh = _hipfn.to_nhwc(x)
        w = weight.detach().contiguous() if weight is not None else None
        b = bias.detach().contiguous() if bias is not None else None
        need = ctx.needs_input_grad[0] or (weight is not None and ctx.needs_input_grad[1]) or (bias is not None and ctx.needs_input_grad[2])
        y, rstd = ops.groupnorm_fwd(xh, groups, w, b, eps, want_rstd=need)
        ctx.detach_mode, ctx.groups, ctx.w, ctx.has_bias = detach, groups, w, bias is not None
        if rstd is not None:
            ctx.save_for_backward(rstd, *([xh] if not detach else []))
        return _hipfn.from_nhwc(y)

    @staticmethod
    def backward(ctx, gy):
        rstd = ctx.saved_tensors[0]
        gh = _hipfn.to_nhwc(gy)
        if ctx.detach_mode:         # explanation mode: variance constant, the input gradient only
            return _hipfn.from_nhwc(ops.groupnorm_bwd_detached(gh, ctx.groups, ctx.w, rstd)), None, None, None, None, None
        xh = ctx.saved_tensors[1]   # training mode (= F.group_norm's gradient, reference :109-113)
        need_w = ctx.w is not None and ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        gx, xhat = ops.groupnorm_bwd(gh, xh, ctx.groups, ctx.w, rstd, want_xhat=need_w)
        Cc = gh.shape[-1]
        g2 = gh.reshape(-1, Cc)
        gw = gb = None
        if need_w:
            gw = ops.colsum(g2, xhat.reshape(-1, Cc)) if Cc % 4 == 0 else (g2 * xhat.reshape(-1, Cc)).sum(0)
        if need_b:
            gb = ops.colsum(g2) if Cc % 4 == 0 else g2.sum(0)
        return _hipfn.from_nhwc(gx), gw, gb, None, None, None


class DetachableGroupNorm2d(nn.GroupNorm, DetachableModule):
    """nn.GroupNorm over [N,C,H,W] whose variance is detached in explanation mode (reference :93-160; the norm of the
    conv stems of the ViT-C models).  Forward and the explanation-mode input gradient are HIP kernels
    (bcos_groupnorm_fwd / _bwd_detached)."""

    def __init__(self, *args, **kwargs):
        DetachableModule.__init__(self)
        super().__init__(*args, **kwargs)

    def forward(self, input: Tensor) -> Tensor:
        assert input.dim() == 4, f"Expected 4D input got {input.dim()}D instead!"
        assert input.shape[1] % self.num_groups == 0, (
            "Number of channels in input should be divisible by num_groups, "
            f"but got input of shape {input.shape} and num_groups={self.num_groups}")
        return _GroupNormFn.apply(input, self.weight, self.bias, self.num_groups, self.eps, self.detach)

    @classmethod
    def from_standard_module(cls, mod: nn.GroupNorm, model_config: dict):
        new = cls(num_groups=mod.num_groups, num_channels=mod.num_channels, eps=mod.eps, affine=mod.affine)
        if model_config.get("weights", None) is not None and mod.affine:
            new.weight.data = mod.weight.data
            if mod.bias is not None:
                new.bias.data = mod.bias.data
        return new


class DetachableGNInstanceNorm2d(DetachableGroupNorm2d):
    def __init__(self, num_channels: int, *args, **kwargs):
        super().__init__(num_channels, num_channels, *args, **kwargs)


class DetachableGNLayerNorm2d(DetachableGroupNorm2d):
    """A CNN detachable layer norm: one group (reference :175-186)."""

    def __init__(self, num_channels: int, *args, **kwargs):
        super().__init__(1, num_channels, *args, **kwargs)
