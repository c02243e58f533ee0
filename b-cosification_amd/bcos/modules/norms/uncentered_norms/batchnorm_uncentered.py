"""BatchNormUncentered2d (eval path) on MI355X.

Reference: bcos/modules/norms/uncentered_norms/batchnorm_uncentered.py -- functional :21-60, class :63-115,
`from_standard_module` :117-141.  In eval mode the layer is a per-channel affine map
    y = x / sqrt(running_var + eps) * weight (+ bias)
which is linear in x, so it needs no special handling in explanation mode.  Standalone it runs as one
streaming HIP kernel (bcos_channel_affine); inside the fused engine (bcos_hip/engine.py) it disappears
into the epilogue of the preceding B-cos convolution.  Training-mode batch statistics are SURVEY.md
section 8(f) N4 (not built).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from bcos_hip import ops

from bcos.modules import _hipfn
from bcos.modules.common import DetachableModule

__all__ = ["BatchNormUncentered2d", "batch_norm_uncentered_2d"]


class _ChannelAffineFn(Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        _hipfn.require_hip(x, "BatchNormUncentered2d")
        xh = _hipfn.to_nhwc(x)
        c = xh.shape[-1]
        if c % 4 != 0:
            raise _hipfn.BcosHipError(f"BatchNormUncentered2d HIP kernel needs C % 4 == 0 (got {c})")
        ctx.save_for_backward(scale)
        y_cl, y = _hipfn.empty_cl(x.shape[0], c, x.shape[2], x.shape[3], x.device)
        ops.channel_affine(xh, scale, shift, out=y)
        return y_cl

    @staticmethod
    def backward(ctx, gy):
        (scale,) = ctx.saved_tensors
        g_cl, g = _hipfn.empty_cl(gy.shape[0], gy.shape[1], gy.shape[2], gy.shape[3], gy.device)
        ops.channel_affine(_hipfn.to_nhwc(gy), scale, None, out=g)
        return g_cl, None, None


def batch_norm_uncentered_2d(input, running_var, weight=None, bias=None, training=False, momentum=0.1,
                             eps=1e-5, detach=False):
    """Functional form (reference :21-60); eval only."""
    assert input.dim() == 4, "input should be a 4d tensor!"
    if training:
        raise NotImplementedError("BatchNormUncentered2d with batch statistics (training mode) is not built "
                                  "for MI355X yet: call model.eval() (SURVEY.md section 8(f) N4)")
    assert running_var is not None, "running_var must be defined in eval mode"
    scale = 1.0 / (running_var.detach() + eps).sqrt()
    if weight is not None:
        scale = weight.detach() * scale
    shift = bias.detach().contiguous() if bias is not None else None
    return _ChannelAffineFn.apply(input, scale.contiguous(), shift)


class BatchNormUncentered2d(nn.BatchNorm2d, DetachableModule):
    def __init__(self, *args, **kwargs):
        self.bias = kwargs.pop("bias", None)
        DetachableModule.__init__(self)
        super().__init__(*args, **kwargs)

    def forward(self, input):
        use_batch_stats = self.training or (self.running_mean is None and self.running_var is None)
        return batch_norm_uncentered_2d(input, self.running_var, self.weight, self.bias, training=use_batch_stats,
                                        momentum=0.0 if self.momentum is None else self.momentum, eps=self.eps,
                                        detach=self.detach)

    def channel_scale_shift(self):
        """(scale[C], shift[C] or None) of the eval-mode affine map, for fusion into a conv epilogue."""
        scale = 1.0 / (self.running_var.detach() + self.eps).sqrt()
        if self.weight is not None:
            scale = self.weight.detach() * scale
        shift = self.bias.detach().contiguous() if isinstance(self.bias, torch.Tensor) else None
        return scale.contiguous(), shift

    @classmethod
    def from_standard_module(cls, mod, model_config):
        """nn.BatchNorm2d -> uncentered BN; with norm_layer == 'BnUncV2' the running mean is folded into the
        bias: bias <- bias - running_mean / std * weight (reference :131-134)."""
        new = cls(num_features=mod.num_features, eps=mod.eps, momentum=mod.momentum, affine=mod.affine,
                  track_running_stats=mod.track_running_stats, bias=mod.bias is not None)
        new.weight.data = mod.weight.data
        norm_layer = model_config["bcosify_args"].get("norm_layer", "BnUncV2")
        if mod.bias is not None and norm_layer == "BnUncV2":
            std = (mod.running_var.data + mod.eps).sqrt()
            new.bias.data = mod.bias.data - (mod.running_mean.data / std) * mod.weight.data
        else:
            new.bias.data = mod.bias.data
        if mod.running_var is not None:
            new.running_var.data = mod.running_var.data
        if mod.running_mean is not None:
            new.running_mean.data = mod.running_mean.data   # unused afterwards, kept for the state-dict layout
        return new
