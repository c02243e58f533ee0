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

    def _b_value(self) -> float:
        b = self.b
        if self.clamping:
            b = b.clamp(1 + 1e-6)
        if self.b_loss:
            b = self.b + 2
        return float(b.detach().item()) if isinstance(b, torch.Tensor) else float(b)

    def forward(self, in_tensor: Tensor) -> Tensor:
        b = self._b_value()
        if not self.b_loss and not self.clamping:
            plain_b = self.b.detach().item() if isinstance(self.b, torch.Tensor) else self.b
            if plain_b == 1:
                b = 1.0
        lin = self.linear
        cfg = dict(b=b, max_out=self.max_out, detach=self.detach, cache=self._wcache, w_src=lin.weight,
                   force_pow=bool(self.b_loss))
        return _hipfn.BcosLinearFn.apply(in_tensor, lin.weight, lin.bias, cfg)

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
