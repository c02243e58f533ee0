"""BatchNormUncentered2d on MI355X.

Reference: bcos/modules/norms/uncentered_norms/batchnorm_uncentered.py -- functional :21-60, class :63-115,
`from_standard_module` :117-141.  In eval mode the layer is a per-channel affine map
    y = x / sqrt(running_var + eps) * weight (+ bias)
which is linear in x, so it needs no special handling in explanation mode.  Standalone it runs as one
streaming HIP kernel (bcos_channel_affine); inside the fused engine (bcos_hip/engine.py) it disappears
into the epilogue of the preceding B-cos convolution.
Training mode (reference :36-44): var = x.var((0,2,3), unbiased=False) of the batch -- the
centred variance, although the normalisation itself does not subtract the mean --, running_var <- (1-m) running_var + m var,
y = x / sqrt(var + eps) * weight + bias.  Statistics are two column-reduction launches (mean, then centred squares), the
backward  gx = gy w/std - (x - mean) w (sum gy x) / (M (var+eps)^1.5)  one more reduction and one streaming kernel
(csrc/bcos_train.hip); with `detach` (explanation mode) the variance is a constant and the second term vanishes.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from bcos_hip import ops

from bcos.modules import _hipfn
from bcos.modules.common import DetachableModule

__all__ = ["BatchNormUncentered2d", "batch_norm_uncentered_2d"]


class _ChannelAffineFn(Function):
    """eval mode: y = x * scale + shift, scale = weight / sqrt(running_var + eps).  `weight` / `bias` are passed along so that they
    receive their gradients (sum gy x / std, sum gy) when they are trained with the statistics frozen, like the reference's eval
    branch (:46-60), whose weight and bias stay in the autograd graph."""

    @staticmethod
    def forward(ctx, x, scale, shift, weight, bias, rstd):
        _hipfn.require_hip(x, "BatchNormUncentered2d")
        xh = _hipfn.to_nhwc(x)
        c = xh.shape[-1]
        if c % 4 != 0:
            raise _hipfn.BcosHipError(f"BatchNormUncentered2d HIP kernel needs C % 4 == 0 (got {c})")
        need_w = weight is not None and ctx.needs_input_grad[3]
        ctx.save_for_backward(scale, rstd, *((xh,) if need_w else ()))
        y_cl, y = _hipfn.empty_cl(x.shape[0], c, x.shape[2], x.shape[3], x.device)
        ops.channel_affine(xh, scale, shift, out=y)
        return y_cl

    @staticmethod
    def backward(ctx, gy):
        scale, rstd = ctx.saved_tensors[:2]
        gh = _hipfn.to_nhwc(gy)
        c = gh.shape[-1]
        g_cl = gw = gb = None
        if ctx.needs_input_grad[0]:
            g_cl, g = _hipfn.empty_cl(gy.shape[0], gy.shape[1], gy.shape[2], gy.shape[3], gy.device)
            ops.channel_affine(gh, scale, None, out=g)
        if ctx.needs_input_grad[3]:
            gw = ops.colsum(gh.reshape(-1, c), ctx.saved_tensors[2].view(-1, c)) * rstd
        if ctx.needs_input_grad[4]:
            gb = ops.colsum(gh.reshape(-1, c))
        return g_cl, None, None, gw, gb, None


class _BatchStatsFn(Function):
    """y = x / sqrt(var_batch(x) + eps) * weight + bias with gradients to x, weight and bias."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, detach, stats_out):
        _hipfn.require_hip(x, "BatchNormUncentered2d")
        xh = _hipfn.to_nhwc(x)
        c = xh.shape[-1]
        if c % 4 != 0:
            raise _hipfn.BcosHipError(f"BatchNormUncentered2d HIP kernels need C % 4 == 0 (got {c})")
        x2 = xh.view(-1, c)
        m = x2.shape[0]
        mean = ops.colsum(x2) / m
        var = ops.colsum(x2, x2, mean, mean) / m                 # centred second moment: x.var(unbiased=False)
        stats_out.append(var)
        rstd = torch.rsqrt(var + eps)
        g = rstd if weight is None else weight.detach() * rstd
        y_cl, y = _hipfn.empty_cl(x.shape[0], c, x.shape[2], x.shape[3], x.device)
        ops.channel_affine(xh, g.contiguous(), bias.detach().contiguous() if bias is not None else None, out=y)
        ctx.save_for_backward(xh, mean, rstd, g, weight if weight is not None else rstd)
        ctx.flags = (bool(detach), weight is not None, bias is not None, float(eps))
        return y_cl

    @staticmethod
    def backward(ctx, gy):
        xh, mean, rstd, g, weight = ctx.saved_tensors
        detach, has_w, has_b, eps = ctx.flags
        c = xh.shape[-1]
        gh = _hipfn.to_nhwc(gy)
        g2, x2 = gh.view(-1, c), xh.view(-1, c)
        m = x2.shape[0]
        gx_cl = gw = gb = None
        sgx = ops.colsum(g2, x2) if (has_w and ctx.needs_input_grad[1]) or (ctx.needs_input_grad[0] and not detach) else None
        if ctx.needs_input_grad[0]:
            gx_cl, gx = _hipfn.empty_cl(gy.shape[0], c, gy.shape[2], gy.shape[3], gy.device)
            if detach:                                                # variance held constant: a per-channel scale
                ops.channel_affine(gh, g.contiguous(), None, out=gx)
            else:                                                     # + dL/dvar * 2 (x - mean) / M
                coef = -(g * sgx) * rstd * rstd / m
                ops.channel_axpby(gh, g.contiguous(), xh, mean.contiguous(), coef.contiguous(), out=gx)
        if has_w and ctx.needs_input_grad[1]:
            gw = sgx * rstd                                           # sum gy * x / std
        if has_b and ctx.needs_input_grad[2]:
            gb = ops.colsum(g2)
        return gx_cl, gw, gb, None, None, None


def batch_norm_uncentered_2d(input, running_var, weight=None, bias=None, training=False, momentum=0.1,
                             eps=1e-5, detach=False):
    """Functional form (reference :21-60)."""
    assert input.dim() == 4, "input should be a 4d tensor!"
    if training:
        stats = []
        out = _BatchStatsFn.apply(input, weight, bias, eps, detach, stats)
        if running_var is not None:
            with torch.no_grad():
                running_var.copy_((1 - momentum) * running_var + momentum * stats[0])
        return out
    assert running_var is not None, "running_var must be defined in eval mode"
    rstd = 1.0 / (running_var.detach() + eps).sqrt()
    scale = rstd if weight is None else weight.detach() * rstd
    shift = bias.detach().contiguous() if bias is not None else None
    return _ChannelAffineFn.apply(input, scale.contiguous(), shift, weight, bias if isinstance(bias, torch.Tensor) else None, rstd)


class BatchNormUncentered2d(nn.BatchNorm2d, DetachableModule):
    def __init__(self, *args, **kwargs):
        self.bias = kwargs.pop("bias", None)
        DetachableModule.__init__(self)
        super().__init__(*args, **kwargs)

    def forward(self, input):
        momentum = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:                                  # cumulative moving average (reference :84-88)
                momentum = 1.0 / float(self.num_batches_tracked)
        use_batch_stats = self.training or (self.running_mean is None and self.running_var is None)
        running_var = self.running_var if (not self.training or self.track_running_stats) else None
        return batch_norm_uncentered_2d(input, running_var, self.weight, self.bias, training=use_batch_stats,
                                        momentum=momentum, eps=self.eps, detach=self.detach)

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
