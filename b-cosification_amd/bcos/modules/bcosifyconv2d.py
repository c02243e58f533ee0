"""BcosifyConv2d: the B-cos transform on top of a *plain* nn.Conv2d (no weight normalisation, optional bias)
-- what `bcosify.py` turns every nn.Conv2d of a pretrained network into.

API-compatible with the reference's bcos/modules/bcosifyconv2d.py:7-182 (constructor kwargs `clamping`,
`b_loss`, the `weight` property, `from_standard_module`, `from_standard_module_linear`, and the
`model_config` keys they read).
"""
import torch
import torch.nn as nn
from torch import Tensor

from . import _hipfn
from .bcosconv2d import BcosConv2d

__all__ = ["BcosifyConv2d"]


def _read_config(model_config):
    args = model_config["bcosify_args"]
    return dict(clamping=args.get("clamping", False), b_loss=args.get("learn_b", False),
                b=model_config["bcos_args"].get("b", 1))   # b defaults to 1 (= identity) like the reference (:127)


class BcosifyConv2d(BcosConv2d):
    def __init__(self, *args, clamping: bool = False, b_loss: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.clamping = clamping
        self.b_loss = b_loss
        # swap the unit-norm convolution for a standard one; `self.bias` is None unless
        # from_standard_module attaches one afterwards (reference :18-31)
        self.linear = nn.Conv2d(
            in_channels=self.in_channels,
            out_channels=self.out_channels * self.max_out,
            kernel_size=self.kernel_size,
            stride=self.stride,
            padding=self.padding,
            dilation=self.dilation,
            groups=self.groups,
            bias=self.bias,
            padding_mode=self.padding_mode,
            device=self.device,
            dtype=self.dtype,
        )

    @property
    def weight(self) -> Tensor:
        # CLIP's ModifiedResNet reads conv1.weight.dtype (CLIP/clip/model.py:146)
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
        return self.forward_impl(in_tensor)

    def forward_impl(self, in_tensor: Tensor) -> Tensor:
        lin = self.linear
        if lin.padding_mode != "zeros":
            raise NotImplementedError("only zero padding is implemented by the HIP kernels")
        b, force_pow = self._scaling()
        cfg = dict(stride=tuple(lin.stride), padding=tuple(lin.padding), dilation=tuple(lin.dilation),
                   groups=lin.groups, b=b, max_out=self.max_out, detach=self.detach, cache=self._wcache,
                   w_src=lin.weight, force_pow=force_pow, b_chain=self._b_chain())
        return _hipfn.BcosConv2dFn.apply(in_tensor, lin.weight, lin.bias, cfg, _hipfn.learnable_b(self))

    @classmethod
    def _from(cls, model_config, weight, bias, **geometry):
        cfgd = _read_config(model_config)
        new = cls(bias=bias is not None, **geometry, **cfgd)
        if model_config.get("weights", None) is not None:
            new.linear.weight.data = weight.data.view_as(new.linear.weight.data)
            if bias is not None:
                new.linear.bias = nn.Parameter(bias.data)
        return new

    @classmethod
    def from_standard_module(cls, mod, model_config):
        """nn.Conv2d -> BcosifyConv2d, copying weight (and bias) when model_config['weights'] is set."""
        return cls._from(model_config, mod.weight, mod.bias, in_channels=mod.in_channels,
                         out_channels=mod.out_channels, kernel_size=mod.kernel_size, stride=mod.stride,
                         padding=mod.padding, dilation=mod.dilation, groups=mod.groups,
                         padding_mode=mod.padding_mode)

    @classmethod
    def from_standard_module_linear(cls, mod, model_config):
        """nn.Linear classifier -> 1x1 BcosifyConv2d (applied before global average pooling)."""
        return cls._from(model_config, mod.weight, mod.bias, in_channels=mod.in_features,
                         out_channels=mod.out_features, kernel_size=1, stride=1, padding=0, dilation=1,
                         groups=1, padding_mode="zeros")
