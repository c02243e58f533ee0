"""B-cos 2-D convolution on MI355X.

API-compatible with the reference's bcos/modules/bcosconv2d.py (NormedConv2d :17-41, BcosConv2d :43-262,
BcosConv2dWithScale :265-326): same constructor signatures, attributes (`linear`, `b`, `max_out`, `detach`,
...) and state-dict keys (`linear.weight`), but `forward` is a single fused HIP launch
(bcos_hip.ops.tapconv) instead of a chain of ATen ops.
"""
import math
import warnings
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from bcos_hip import ops

from . import _hipfn
from .common import DetachableModule

__all__ = ["NormedConv2d", "BcosConv2d", "BcosConv2dWithScale"]


class NormedConv2d(nn.Conv2d):
    """nn.Conv2d whose filters are projected to unit L2 norm on every call (reference :17-41).

    The projection w / ||w|| (times the optional per-filter `scale`) runs as one wavefront-per-filter HIP
    kernel (bcos_weight_rownorm_scale) and is cached until the parameter changes.
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.scale = None
        self.use_weight_norm = True
        self._unit_cache = (None, None)
        self._wcache = _hipfn.WeightCache()

    def effective_weight(self, track_grad: bool = True) -> Tensor:
        """`track_grad=False` (explanation mode of the enclosing B-cos layer): always the cached, detached projection."""
        w = self.weight
        if not self.use_weight_norm:
            return w
        _hipfn.require_hip(w, "NormedConv2d")
        if track_grad and _hipfn.wants_projection_grad(self, w, self.scale):       # training step: the projection is part of the graph
            return _hipfn.UnitNormFn.apply(w, self.scale)
        gain = self.scale.detach().reshape(-1).contiguous() if self.scale is not None else None
        key = (w.data_ptr(), w._version, None if gain is None else (gain.data_ptr(), self.scale._version))
        if self._unit_cache[0] != key:
            unit = ops.weight_rownorm_scale(w.detach().contiguous().view(w.shape[0], -1), gain).view_as(w)
            self._unit_cache = (key, unit)
        return self._unit_cache[1]

    def forward(self, in_tensor: Tensor) -> Tensor:
        if self.padding_mode != "zeros":
            raise NotImplementedError("only zero padding is implemented by the HIP kernels")
        return _hipfn.plain_conv2d(in_tensor, self.effective_weight(), self.bias, tuple(self.stride),
                                   tuple(self.padding), tuple(self.dilation), self.groups, self._wcache, self.weight)

    def set_scale(self, weight: Tensor, trainable=False):
        self.scale = nn.Parameter(weight.norm(p=2, dim=(1, 2, 3), keepdim=True), requires_grad=trainable)

    def toggle_weight_norm(self, use_weight_norm):
        self.use_weight_norm = use_weight_norm


class BcosConv2d(DetachableModule):
    """y = |cos(x_patch, w)|^(B-1) * (w_hat . x_patch) with unit-norm filters w_hat (arXiv:2205.10268).

    Parameters are those of the reference class (bcosconv2d.py:84-101): in_channels, out_channels,
    kernel_size=1, stride=1, padding=0, dilation=1, groups=1, padding_mode="zeros", device, dtype,
    bias (ignored: B-cos layers have none), b=2, max_out=1; extra keyword arguments are swallowed like
    the reference does.
    """

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size: Union[int, Tuple[int, ...]] = 1,
        stride: Union[int, Tuple[int, ...]] = 1,
        padding: Union[int, Tuple[int, ...]] = 0,
        dilation: Union[int, Tuple[int, ...]] = 1,
        groups: int = 1,
        padding_mode: str = "zeros",
        device=None,
        dtype=None,
        bias: bool = False,
        b: Union[int, float] = 2,
        max_out: int = 1,
        **kwargs,
    ):
        assert max_out > 0, f"max_out should be greater than 0, was {max_out}"
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.dilation = _hipfn._pair(dilation)
        self.groups = groups
        self.padding_mode = padding_mode
        self.device = device
        self.dtype = dtype
        self.bias = None
        self.b = b
        self.max_out = max_out
        if any(d > 1 for d in self.dilation):
            # the reference switches to a ones-kernel convolution here and warns; the fused kernel handles
            # dilated taps natively, the warning is kept for behavioural parity
            warnings.warn("dilation > 1 is much slower!")
        self._wcache = _hipfn.WeightCache()
        self.linear = NormedConv2d(
            in_channels=in_channels,
            out_channels=out_channels * max_out,
            kernel_size=kernel_size,
            stride=stride,
            padding=padding,
            dilation=dilation,
            groups=groups,
            bias=False,
            padding_mode=padding_mode,
            device=device,
            dtype=dtype,
        )

    # -- what the fused kernel needs from `self.linear` ------------------------------------------------
    def _effective_weight_and_bias(self):
        lin = self.linear
        if isinstance(lin, NormedConv2d):
            return lin.effective_weight(track_grad=not self.detach), lin.bias
        if isinstance(lin, nn.Conv2d):
            return lin.weight, lin.bias
        raise TypeError(f"BcosConv2d.linear must be a (Normed)Conv2d, got {type(lin).__name__}")

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
        return self.forward_impl(in_tensor)

    def forward_impl(self, in_tensor: Tensor) -> Tensor:
        lin = self.linear
        if lin.padding_mode != "zeros":
            raise NotImplementedError("only zero padding is implemented by the HIP kernels")
        cfg = dict(stride=tuple(lin.stride), padding=tuple(lin.padding), dilation=tuple(lin.dilation),
                   groups=lin.groups, b=self._b_value(), max_out=self.max_out, detach=self.detach,
                   cache=self._wcache, w_src=lin.weight)
        if isinstance(lin, NormedConv2d) and lin.use_weight_norm and _hipfn.folds_projection(lin, self.detach):
            # training step of a native layer: the unit-norm projection is folded into the contraction (one forward launch)
            return _hipfn.FoldedUnitNormFn.apply(in_tensor, lin.weight, lin.scale, cfg, _hipfn.learnable_b(self), _hipfn.BcosConv2dFn)
        w, bias = self._effective_weight_and_bias()
        return _hipfn.BcosConv2dFn.apply(in_tensor, w, bias, cfg, _hipfn.learnable_b(self))

    def calc_patch_norms(self, in_tensor: Tensor) -> Tensor:
        """sqrt(sum over each patch of x^2 + 1e-6), [N,1|G,Ho,Wo]  (reference :196-231): the same kernel run
        with a zero filter bank, writing only the norms."""
        lin = self.linear
        _hipfn.require_hip(in_tensor, "calc_patch_norms")
        x = _hipfn.to_nhwc(in_tensor)
        N, H, W, Cin = x.shape
        kh, kw = lin.kernel_size
        G = lin.groups
        cin_g = Cin // G
        if G == 1:
            x = _hipfn._pad_last(x)
        cpad = x.shape[3] if G == 1 else cin_g
        zero_w = torch.zeros((4, kh, kw, cpad), device=x.device, dtype=torch.float32)
        geom = ops.fwd_geom(N, H, W, cpad, 4, kh, kw, lin.stride[0], lin.stride[1], lin.padding[0], lin.padding[1],
                            lin.dilation[0], lin.dilation[1])
        norm = torch.empty((N, geom["P"], geom["Q"], G), device=x.device, dtype=torch.float32)
        dummy = torch.empty((N, geom["P"], geom["Q"], 4), device=x.device, dtype=torch.float32)
        if G > 1:
            geom.update(groups=G, a_pitch=Cin, out_pitch=4 * G, norm_pitch=G)
            dummy = torch.empty((N, geom["P"], geom["Q"], 4 * G), device=x.device, dtype=torch.float32)
            zero_w = torch.zeros((4 * G, kh, kw, cpad), device=x.device, dtype=torch.float32)
        ops.tapconv(x, zero_w, geom, out=dummy, norm_out=norm, bcos_mode=_hipfn.BCOS_CONV_EPS, flags=_hipfn.BCOS_EPI_NORM_ONLY,
                    track_absmax=False)
        norm = norm.permute(0, 3, 1, 2)
        if G > 1:
            norm = torch.repeat_interleave(norm, repeats=self.out_channels // G, dim=1)
        return norm

    _calc_patch_norms_slow = calc_patch_norms

    def extra_repr(self) -> str:
        s = f"B={self._b_value():g}"
        if self.max_out > 1:
            s += f", max_out={self.max_out}"
        return s + ","


class BcosConv2dWithScale(BcosConv2d):
    """Deprecated v1 variant: output divided by a fixed scale (reference :265-326)."""

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size: Union[int, Tuple[int, ...]] = 1,
        stride: Union[int, Tuple[int, ...]] = 1,
        padding: Union[int, Tuple[int, ...]] = 0,
        dilation: Union[int, Tuple[int, ...]] = 1,
        groups: int = 1,
        padding_mode: str = "zeros",
        device=None,
        dtype=None,
        b: Union[int, float] = 2,
        max_out: int = 1,
        scale: Optional[float] = None,
        scale_factor: Union[int, float] = 100.0,
        **kwargs,
    ):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                         device, dtype, b=b, max_out=max_out, **kwargs)
        if scale is None:
            ks = kernel_size if not isinstance(kernel_size, tuple) else math.sqrt(kernel_size[0] * kernel_size[1])
            self.scale = (ks * math.sqrt(self.in_channels)) / scale_factor
        else:
            assert scale != 1.0, "For scale=1.0, use the normal BcosConv2d instead!"
            self.scale = scale
        warnings.warn("BcosConv2dWithScale is deprecated and will be removed in a future version. "
                      "Use BcosConv2d with scale=1.0 instead.", DeprecationWarning)

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.forward_impl(in_tensor) / self.scale

    def extra_repr(self) -> str:
        return f"scale={self.scale:.3f}, " + super().extra_repr()
