"""Explanation helpers of B-cos models -- API-compatible with the reference's bcos/common.py
(BcosUtilMixin :38-344, explanation_mode :347-384, gradient_to_image :387-436).

`explain` keeps the reference semantics (one image, arg-max logit, autograd backward through the module
graph -- every B-cos layer's backward is a HIP input-gradient kernel).  `explain_batch` is the MI355X
addition: whole batches through the fused engine (bcos_hip/engine.py) when the wrapped network is a
recognised topology, else through the same module graph.
"""
import warnings
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

if torch.__version__ < "2.0":
    from torch.autograd.grad_mode import _DecoratorContextManager  # noqa
else:
    from torch.utils._contextlib import _DecoratorContextManager  # noqa

__all__ = ["BcosUtilMixin", "explanation_mode", "gradient_to_image", "plot_contribution_map"]

TensorLike = Union[Tensor, np.ndarray]


class explanation_mode(_DecoratorContextManager):
    """Context manager / decorator: sets `.detach` on every module that has `set_explanation_mode`
    while active (reference :347-384).  Not thread-safe, like the reference."""

    def __init__(self, model: "nn.Module"):
        self.model = model
        self.expl_modules = None

    def find_expl_modules(self) -> None:
        self.expl_modules = [m for m in self.model.modules() if hasattr(m, "set_explanation_mode")]

    def __enter__(self):
        if self.expl_modules is None:
            self.find_expl_modules()
        for m in self.expl_modules:
            m.set_explanation_mode(True)

    def __exit__(self, exc_type, exc_val, exc_tb):
        for m in self.expl_modules:
            m.set_explanation_mode(False)


class BcosUtilMixin:
    """Mixin adding `explanation_mode()`, `explain()`, `attribute()`, ... to a B-cos nn.Module.
    Inherit from it *before* the nn.Module base class."""

    to_probabilities = torch.sigmoid

    def __init__(self, *args: Any, **kwargs: Any):
        self.__explanation_mode_ctx = explanation_mode(self)  # type: ignore
        super().__init__(*args, **kwargs)

    def explanation_mode(self) -> "explanation_mode":
        return self.__explanation_mode_ctx

    # -- single image, reference semantics (bcos/common.py:92-188) ---------------------------------------
    def explain(self, in_tensor, idx=None, **grad2img_kwargs) -> "Dict[str, Any]":
        if in_tensor.ndim == 3:
            raise ValueError("Expected 4-dimensional input tensor")
        if in_tensor.shape[0] != 1:
            raise ValueError("Expected batch size of 1")
        if not in_tensor.requires_grad:
            warnings.warn("Input tensor did not require grad! Has been set automatically to True!")
            in_tensor.requires_grad = True
        if self.training:  # noqa
            warnings.warn("Model is in training mode! This might lead to unexpected results! Use model.eval()!")

        result = dict()
        with torch.enable_grad(), self.explanation_mode():
            out = self(in_tensor)  # noqa
            pred_out = out.max(1)
            result["prediction"] = pred_out.indices.item()
            if idx is None:
                logit = pred_out.values
                result["explained_class_idx"] = pred_out.indices.item()
            else:
                logit = out[0, idx]
                result["explained_class_idx"] = idx
            logit.backward(inputs=[in_tensor])

        result["dynamic_linear_weights"] = in_tensor.grad
        result["contribution_map"] = (in_tensor * in_tensor.grad).sum(1)
        result["explanation"] = gradient_to_image(in_tensor[0], in_tensor.grad[0], **grad2img_kwargs)
        return result

    # -- batched explanation (MI355X addition) -----------------------------------------------------------
    def explain_batch(self, images: Tensor, targets: Optional[Tensor] = None, render: bool = False,
                      smooth: int = 15, alpha_percentile: float = 99.5) -> "Dict[str, Tensor]":
        """Forward + explanation for a whole batch.

        Returns logits [N,K], prediction [N], explained_class_idx [N], dynamic_linear_weights [N,C,H,W]
        and contribution_map [N,H,W]; row n equals `explain(images[n:n+1], idx=targets[n])`.
        Uses the fused engine when one is attached (`bcos_hip.engine.attach`), else autograd over the modules.
        `render=True` adds "explanation" [N,H,W,4]: the RGBA images of `gradient_to_image` for the whole batch, rendered
        on the device (bcos_render_explanations).
        """
        out = self._explain_batch(images, targets)
        if render:
            from bcos_hip import ops
            out["explanation"] = ops.render_explanations(images.detach().contiguous(),
                                                         out["dynamic_linear_weights"].contiguous(), smooth=smooth,
                                                         alpha_percentile=alpha_percentile)
        return out

    def _explain_batch(self, images: Tensor, targets: Optional[Tensor] = None) -> "Dict[str, Tensor]":
        engine = getattr(self, "_bcos_engine", None)
        if engine is not None and getattr(engine, "supports_explain", True) and not self.training:
            return engine.explain(images, targets)
        x = images.detach().clone().requires_grad_(True)
        with torch.enable_grad(), self.explanation_mode():
            logits = self(x)  # noqa
            pred = logits.max(1)
            idx = pred.indices if targets is None else targets.to(logits.device)
            (grad,) = torch.autograd.grad(logits.gather(1, idx.view(-1, 1)).sum(), x)
        from bcos_hip import ops
        return dict(logits=logits.detach(), prediction=pred.indices, explained_class_idx=idx,
                    dynamic_linear_weights=grad,
                    contribution_map=ops.contrib_map(x.detach().contiguous(), grad.contiguous()))

    # -- Input x Gradient attribution (reference :280-344, captum.InputXGradient semantics) ------------------
    def attribute(self, image: Union[Tensor, Tuple[Tensor]], target: Union[int, Tuple[int], Tensor, List[int]],
                  **kwargs: Any) -> Tensor:
        _ = kwargs
        from interpretability.explanation_methods.explainers.captum import IxG

        with self.explanation_mode():
            return IxG(self).attribute(image, target)

    def attribute_selection(self, image: Tensor, targets: Union[Tuple[int], Tensor, List[int]], **kwargs: Any) -> Tensor:
        _ = kwargs
        return torch.cat([self.attribute(image, t) for t in targets], dim=0)

    @staticmethod
    def gradient_to_image(image: "Tensor", linear_mapping: "Tensor", smooth: int = 15,
                          alpha_percentile: float = 99.5) -> "np.ndarray":
        return gradient_to_image(image, linear_mapping, smooth=smooth, alpha_percentile=alpha_percentile)

    @staticmethod
    def plot_contribution_map(contribution_map, ax=None, vrange=None, vshift=0, hide_ticks=True, cmap="bwr",
                              percentile=99.5):
        return plot_contribution_map(contribution_map, ax, vrange, vshift, hide_ticks, cmap, percentile)


def gradient_to_image(image, linear_mapping, smooth=15, alpha_percentile=99.5, return_contribs=False):
    """RGBA rendering [H,W,4] of the dynamic linear mapping of one image (reference :387-436): colour = the
    per-pixel weight direction over the (r,g,b,1-r,1-g,1-b) channels, alpha = its L2 norm, zeroed where the
    contribution is negative, box-smoothed and clipped at the `alpha_percentile` quantile.

    Rendered by the batched device kernel (bcos_hip.ops.render_explanations: per-pixel colour/alpha, LDS box filter, exact
    quantile by radix select -- SURVEY.md section 8(f) N1).  There is no CPU formulation in this package: CPU tensors raise
    (the torch restatement of the reference's statements is oracle.bcos_oracle.gradient_to_image, test infrastructure)."""
    from bcos_hip import ops
    from bcos_hip.lib import BcosHipError
    if image.dim() != 3 or image.shape[0] != 6 or tuple(linear_mapping.shape) != tuple(image.shape):
        raise ValueError(f"gradient_to_image: expected image and linear_mapping of shape [6, H, W], got {tuple(image.shape)} / "
                         f"{tuple(linear_mapping.shape)}")
    if smooth and smooth % 2 == 0:
        raise BcosHipError("gradient_to_image: the device box filter needs an odd `smooth` (the reference's default is 15)")
    rgba_t = ops.render_explanations(image.detach()[None].float().contiguous(), linear_mapping.detach()[None].float().contiguous(),
                                     smooth=smooth, alpha_percentile=alpha_percentile)[0]
    rgba = rgba_t.cpu().numpy()
    if return_contribs:
        return rgba, (image * linear_mapping).sum(0, keepdim=True).detach().cpu().numpy()
    return rgba


def plot_contribution_map(contribution_map: TensorLike, ax=None, vrange: Optional[float] = None, vshift: float = 0,
                          hide_ticks: bool = True, cmap: str = "bwr", percentile: float = 99.5):
    """Host-side matplotlib helper (reference :439-516); out of the hot path, kept for API completeness."""
    import matplotlib.pyplot as plt

    if isinstance(contribution_map, torch.Tensor):
        contribution_map = contribution_map.detach().cpu().numpy()
    contribution_map = np.squeeze(contribution_map)
    if vrange is None or vrange == "auto":
        vrange = np.percentile(np.abs(contribution_map.flatten()), percentile)
    if ax is None:
        ax = plt.gca()
    im = ax.imshow(contribution_map, cmap=cmap, vmin=-vrange + vshift, vmax=vrange + vshift)
    if hide_ticks:
        ax.set_xticks([])
        ax.set_yticks([])
    return ax, im
