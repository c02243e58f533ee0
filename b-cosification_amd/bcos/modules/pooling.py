"""HIP-backed average pooling for the module path.

The B-cosification recipe swaps the ResNet stem's MaxPool for nn.AvgPool2d(3, 2, 1)
(bcos/experiments/ImageNet/bcosification/experiment_parameters.py:99), which puts an average pool on the
explanation path (a21 in SURVEY.md section 8).  `AvgPool2d` below is an nn.AvgPool2d subclass (so isinstance
checks and reprs are unchanged) whose forward AND input-gradient run as NHWC streaming HIP kernels
(bcos_avgpool2d_fwd / _bwd).  Besides keeping activations channels-last between the B-cos layers, this avoids
PyTorch-ROCm 2.10's avg_pool2d backward for channels_last inputs, which returns wrong gradients on gfx950
(measured: relL2 0.99 against the CPU result, scripts/dbg_pool.py).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from bcos_hip import ops

from . import _hipfn

__all__ = ["AvgPool2d", "use_hip_pools"]


class _AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        xh = _hipfn.to_nhwc(x)
        n, h, w, c = xh.shape
        oh, ow = ops.conv_out_size(h, k, s, p), ops.conv_out_size(w, k, s, p)
        y_cl, y = _hipfn.empty_cl(n, c, oh, ow, x.device)
        ops.avgpool2d_fwd(xh, k, s, p, out=y)
        ctx.geom = (h, w, k, s, p)
        return y_cl

    @staticmethod
    def backward(ctx, gy):
        h, w, k, s, p = ctx.geom
        g_cl, g = _hipfn.empty_cl(gy.shape[0], gy.shape[1], h, w, gy.device)
        ops.avgpool2d_bwd(_hipfn.to_nhwc(gy), h, w, k, s, p, out=g)
        return g_cl, None, None, None


class AvgPool2d(nn.AvgPool2d):
    def _hip_ok(self, x):
        k, s, p = _hipfn._pair(self.kernel_size), _hipfn._pair(self.stride), _hipfn._pair(self.padding)
        return (x.dim() == 4 and x.shape[1] % 4 == 0 and k[0] == k[1] and s[0] == s[1] and p[0] == p[1]
                and not self.ceil_mode and self.count_include_pad and self.divisor_override is None)

    def forward(self, input):
        _hipfn.require_hip(input, "bcos.modules.pooling.AvgPool2d")
        if not self._hip_ok(input):
            raise _hipfn.BcosHipError(
                "bcos.modules.pooling.AvgPool2d: needs a 4-d tensor with C % 4 == 0 and square, "
                "count_include_pad pooling without ceil_mode / divisor_override")
        return _AvgPoolFn.apply(input, _hipfn._pair(self.kernel_size)[0], _hipfn._pair(self.stride)[0],
                                _hipfn._pair(self.padding)[0])


def use_hip_pools(model: nn.Module) -> nn.Module:
    """Replace every plain nn.AvgPool2d in `model` by the HIP-backed subclass (same hyper-parameters)."""
    for name, child in list(model.named_children()):
        if type(child) is nn.AvgPool2d:
            setattr(model, name, AvgPool2d(child.kernel_size, child.stride, child.padding, child.ceil_mode,
                                           child.count_include_pad, child.divisor_override))
        else:
            use_hip_pools(child)
    return model
