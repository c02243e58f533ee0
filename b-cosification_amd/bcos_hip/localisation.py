"""Grid pointing game ("localisation") harness on the fused engine -- SURVEY.md section 8(f) row N2.

Reference: `LocalisationAnalyser.analysis` / `make_multi_image` (interpretability/analyses/localisation.py:248-415,
417-466) with `explainer.attribute_selection(multi_img, tgts).sum(1)` (explanation_methods/utils.py:82-99).  The
reference re-runs the network forward once per target; here the multi-image is forwarded ONCE and only the
input-gradient pass is repeated per target (`ResNetEngine.explain_targets`), then smoothing, clamping and the per-cell
shares run as two small device kernels (bcos_box_filter, bcos_localisation_fractions).  Dataset iteration, confidence
sorting, plotting and result files stay with the caller (out of scope).
"""
from typing import Dict, Optional

import torch

from . import ops


def make_multi_image(imgs: torch.Tensor) -> torch.Tensor:
    """[g*g, C, h, w] -> [1, C, g*h, g*w]: image i = a*g + b goes to grid row b, column a
    (localisation.py:434-446: view(-1, g, g, C, h, w).permute(0, 3, 2, 4, 1, 5))."""
    g = int(round(imgs.shape[0] ** 0.5))
    if g * g != imgs.shape[0]:
        raise ValueError(f"make_multi_image: {imgs.shape[0]} images do not form a square grid")
    return _tile(imgs, g)


def _tile(imgs: torch.Tensor, g: int) -> torch.Tensor:
    _, C, h, w = imgs.shape
    return imgs.view(-1, g, g, C, h, w).permute(0, 3, 2, 4, 1, 5).reshape(-1, C, h * g, w * g)


def make_multi_images(imgs: torch.Tensor, g: int) -> torch.Tensor:
    """Batched form: [B*g*g, C, h, w] -> [B, C, g*h, g*w], consecutive groups of g*g images per grid."""
    if imgs.shape[0] % (g * g):
        raise ValueError(f"make_multi_images: {imgs.shape[0]} images are not a multiple of {g}x{g}")
    return _tile(imgs, g)


@torch.no_grad()
def grid_pointing_game(engine, multi_imgs: torch.Tensor, targets: torch.Tensor, single_shape: int, smooth: int = 0,
                       neg: bool = False, attributions: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """multi_imgs [B, C, g*s, g*s] (s = single_shape), targets [B, T] (the class of every grid cell, T = g*g, in the
    image order of make_multi_image) ->
        attributions [B, T, H, W]   contribution map of target t on multi-image b (= attribute_selection(...).sum(1)),
        fractions    [B, T, T]      share of the (smoothed, positive) attribution of target t inside cell c,
        metric       [B, T]         fractions[b, t, t]: the localisation score of localisation.py:402."""
    B = multi_imgs.shape[0]
    tg = torch.as_tensor(targets, device=multi_imgs.device, dtype=torch.int64).view(B, -1)
    T = tg.shape[1]
    if attributions is None:
        attributions = engine.explain_targets(multi_imgs, tg)["contribution_maps"]
    H, W = attributions.shape[-2:]
    att = attributions.reshape(B * T, H, W).contiguous()
    if smooth and smooth > 1:
        att = ops.box_filter(att, smooth)
    frac = ops.localisation_fractions(att, single_shape, single_shape, neg=neg).view(B, T, -1)
    if frac.shape[-1] != T:
        raise ValueError(f"grid_pointing_game: {T} targets but {frac.shape[-1]} grid cells of size {single_shape}")
    metric = torch.diagonal(frac, dim1=1, dim2=2)
    if neg:
        metric = 1 - metric                       # localisation.py:410-411
    return dict(attributions=attributions, fractions=frac, metric=metric)
