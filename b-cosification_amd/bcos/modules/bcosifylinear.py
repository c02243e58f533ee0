"""BcosifyLinear: the B-cos transform on top of a plain nn.Linear (no weight normalisation, optional bias).

API-compatible with the reference's bcos/modules/bcosifylinear.py:17-133.
"""
import torch
import torch.nn as nn
from torch import Tensor

from . import _hipfn
from .bcoslinear import BcosLinear

__all__ = ["BcosifyLinear"]


class BcosifyLinear(BcosLinear):
    def __init__(self, *args, clamping: bool = False, b_loss: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.clamping = clamping
        self.b_loss = b_loss
        self.linear = nn.Linear(
            in_features=self.in_features,
            out_features=self.out_features * self.max_out,
            bias=self.bias,
            device=self.device,
            dtype=self.dtype,
        )

    @property
    def weight(self) -> Tensor:
        return self.linear.weight

    def _b_host(self) -> float:
        """self.b as a host float; a tensor / Parameter B (trainer.py:463) is read back once per in-place version instead
        of forcing a device sync in every forward."""
        b = self.b
        if not isinstance(b, torch.Tensor):
            return float(b)
        key = (b.data_ptr(), b._version)
        cached = getattr(self, "_b_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, float(b.detach().item()))
            object.__setattr__(self, "_b_cache", cached)
        return cached[1]

    def _scaling(self):
        """(exponent B handed to the kernel, force the general pow form) following the reference's branches
        (bcosifyconv2d.py:60-65,78-79,91-98 / bcosifylinear.py:52-57,70-71,83-90): `self.b == 1` without b_loss returns the
        plain linear output even when clamping is on; `self.b == 2` without b_loss takes |lin| / norm; everything else is
        (|cos| + 1e-6)^(B_eff - 1) with B_eff = b + 2 (b_loss), clamp(b, 1 + 1e-6) (clamping) or b."""
        b = self._b_host()
        if not self.b_loss:
            if b == 1:
                return 1.0, False
            if b == 2:
                return 2.0, False
        if self.b_loss:
            return b + 2.0, True
        if self.clamping:
            return max(b, 1.0 + 1e-6), True
        return b, False

    def _b_value(self) -> float:
        return self._scaling()[0]

    def _b_chain(self) -> float:
        """d B_eff / d b for a learnable exponent: clamp(b, 1 + 1e-6) passes the gradient where b >= the bound (fp32 compare,
        like torch.clamp's backward); b + 2 (b_loss) and the plain exponent pass it unchanged."""
        if self.clamping and not self.b_loss:
            return 1.0 if self._b_host() >= float(torch.tensor(1.0 + 1e-6, dtype=torch.float32)) else 0.0
        return 1.0

    def forward(self, in_tensor: Tensor) -> Tensor:
        b, force_pow = self._scaling()
        lin = self.linear
        cfg = dict(b=b, max_out=self.max_out, detach=self.detach, cache=self._wcache, w_src=lin.weight,
                   force_pow=force_pow, b_chain=self._b_chain())
        return _hipfn.BcosLinearFn.apply(in_tensor, lin.weight, lin.bias, cfg, _hipfn.learnable_b(self))

    @classmethod
    def from_standard_module(cls, mod, model_config):
        """nn.Linear -> BcosifyLinear (reference :109-133)."""
        args = model_config["bcosify_args"]
        new = cls(mod.in_features, mod.out_features, bias=mod.bias is not None, device=mod.weight.device,
                  dtype=mod.weight.dtype, max_out=1, clamping=args.get("clamping", False),
                  b_loss=args.get("learn_b", False), b=model_config["bcos_args"].get("b", 1))
        if model_config.get("weights", None) is not None:
            new.linear.weight.data = mod.weight.data
            if mod.bias is not None:
                new.linear.bias = nn.Parameter(mod.bias.data)
        return new
