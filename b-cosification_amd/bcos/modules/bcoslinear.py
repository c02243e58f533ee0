"""B-cos linear layer on MI355X.

API-compatible with the reference's bcos/modules/bcoslinear.py (NormedLinear :20-27, BcosLinear :30-142).
The convolutional models do not use this class (they use 1x1 BcosConv2d); the ViTs do.  Both lower to the
same fused HIP GEMM (bcos_hip.ops.linear_fwd), the only difference being where the epsilon sits:
norm = ||x||_2 + 1e-12 here (reference :113) versus sqrt(sum x^2 + 1e-6) for convolutions.
"""
from typing import Union

import torch
import torch.nn as nn
from torch import Tensor

from bcos_hip import ops

from . import _hipfn
from .common import DetachableModule

__all__ = ["NormedLinear", "BcosLinear"]


class NormedLinear(nn.Linear):
    """nn.Linear with rows projected to unit L2 norm on every call (reference :20-27)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._unit_cache = (None, None)
        self._wcache = _hipfn.WeightCache()

    def effective_weight(self, track_grad: bool = True) -> Tensor:
        """`track_grad=False` (explanation mode of the enclosing B-cos layer): always the cached, detached projection."""
        w = self.weight
        _hipfn.require_hip(w, "NormedLinear")
        if track_grad and _hipfn.wants_projection_grad(self, w):    # training step: the projection is part of the graph
            return _hipfn.UnitNormFn.apply(w, None)
        key = (w.data_ptr(), w._version)
        if self._unit_cache[0] != key:
            self._unit_cache = (key, ops.weight_rownorm_scale(w.detach().contiguous(), None))
        return self._unit_cache[1]

    def forward(self, input: Tensor) -> Tensor:
        return _hipfn.plain_linear(input, self.effective_weight(), self.bias, self._wcache, self.weight)


class BcosLinear(DetachableModule):
    """y = |cos(x, w)|^(B-1) * (w_hat . x) over the last dimension (arXiv:2205.10268).

    Constructor arguments as in the reference (:59-68): in_features, out_features, bias (ignored),
    device, dtype, b=2, max_out=1.
    """

    def __init__(
        self,
        in_features: int,
        out_features: int,
        bias: bool = False,
        device=None,
        dtype=None,
        b: Union[int, float] = 2,
        max_out: int = 1,
    ) -> None:
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.bias = False
        self.device = device
        self.dtype = dtype
        self.b = b
        self.max_out = max_out
        self._wcache = _hipfn.WeightCache()
        self.linear = NormedLinear(in_features, out_features * self.max_out, bias=False, device=device, dtype=dtype)

    def _effective_weight_and_bias(self):
        lin = self.linear
        if isinstance(lin, NormedLinear):
            return lin.effective_weight(track_grad=not self.detach), lin.bias
        if isinstance(lin, nn.Linear):
            return lin.weight, lin.bias
        raise TypeError(f"BcosLinear.linear must be a (Normed)Linear, got {type(lin).__name__}")

    def _b_value(self) -> float:
        """self.b as a host float; a tensor / Parameter B (trainer.py:463) is read back once per in-place version instead of
        forcing a device sync in every forward."""
        b = self.b
        if not isinstance(b, torch.Tensor):
            return float(b)
        key = (b.data_ptr(), b._version)
        cached = getattr(self, "_b_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, float(b.detach().item()))
            object.__setattr__(self, "_b_cache", cached)
        return cached[1]

    def forward(self, in_tensor: Tensor) -> Tensor:
        lin = self.linear
        cfg = dict(b=self._b_value(), max_out=self.max_out, detach=self.detach, cache=self._wcache,
                   w_src=lin.weight)
        if isinstance(lin, NormedLinear) and _hipfn.folds_projection(lin, self.detach):
            # training step of a native layer: the unit-norm projection is folded into the contraction (one forward launch)
            return _hipfn.FoldedUnitNormFn.apply(in_tensor, lin.weight, None, cfg, _hipfn.learnable_b(self), _hipfn.BcosLinearFn)
        w, bias = self._effective_weight_and_bias()
        return _hipfn.BcosLinearFn.apply(in_tensor, w, bias, cfg, _hipfn.learnable_b(self))

    def extra_repr(self) -> str:
        s = f"B={self._b_value():g}"
        if self.max_out > 1:
            s += f", max_out={self.max_out}"
        return s + ","
