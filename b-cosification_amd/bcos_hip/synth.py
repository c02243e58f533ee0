"""Synthetic, *calibrated* B-cosified networks for benchmarks and parity tests.

There is no network access for pretrained weights, and a B-cos net with default random init collapses to
~1e-12 activations after a few layers, which would make every parity check vacuous.
Recipe (deterministic given the seeds and the torch version of the image):
  1. torch.manual_seed(seed); standard topology with default init;
  2. BatchNorm statistics / affine parameters randomised BEFORE conversion so the BnUncV2 fold is exercised;
  3. conversion with the reference's config keys (bcosification/experiment_parameters.py:86-100), MaxPool ->
     AvgPool2d(3,2,1), all biases set to None (bcosification/model.py:45-55);
  4. calibration in execution order on a seed batch: every B-cos conv's weight is scaled by rms(y)^-1/2
     (y is quadratic in W) so that rms(y) = 1, every uncentered BN gets running_var <- var(input).
Step 4 is either executed (`calibrate`, on whatever device / module implementation the net lives on) or
replayed from recorded numbers (`apply_calibration`), which is how the golden fixtures pin identical weights
in the build container (reference modules, CPU) and on the GPU box (HIP modules).
"""
import math
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn


def resnet_model_config(arch: str = "resnet50") -> dict:
    """The `model` section of CONFIGS['resnet_50'] / ['resnet_18'] (bcosification/experiment_parameters.py:41-106)."""
    return dict(
        is_bcos=True,
        name=arch,
        last_layer_name="fc",
        weights="synthetic",     # truthy: the converters copy the weights of the standard modules
        args=dict(num_classes=1000, logit_bias=-math.log(1000 - 1)),
        bcos_args=dict(b=2, max_out=1),
        bcosify_args=dict(fix_b=True, use_bias=False, norm_layer="BnUncV2", manual_optim=False, gap=True,
                          act_layer=True),
        standard_changes={"maxpool": nn.AvgPool2d(kernel_size=3, stride=2, padding=1)},
    )


def randomize_batchnorm(model: nn.Module, gen: torch.Generator):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            m.running_mean.copy_(torch.randn(c, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(c, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(c, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(c, generator=gen) * 0.1)


def standard_resnet(arch: str, seed: int = 0, resnet_cls=None, blocks=None):
    """Step 1-2.  `resnet_cls` / `blocks` let the golden generator pass the reference's own ResNetBcos."""
    if resnet_cls is None:
        from bcos.models.standard_models import BasicBlock, Bottleneck, ResNetBcos
        resnet_cls, blocks = ResNetBcos, dict(basic=BasicBlock, bottleneck=Bottleneck)
    # ("resnet14b": one Bottleneck per stage -- every block has a downsample shortcut; the shallow instance behind the tight training fixture)
    spec = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]),
            "resnet50": ("bottleneck", [3, 4, 6, 3]), "resnet14b": ("bottleneck", [1, 1, 1, 1])}[arch]
    torch.manual_seed(seed)
    net = resnet_cls(blocks[spec[0]], spec[1])
    gen = torch.Generator().manual_seed(seed + 1)
    randomize_batchnorm(net, gen)
    return net


def finish_conversion(model: nn.Module, model_config: dict, hip_pools: bool = True):
    """bcosification/model.py:44-55: apply `standard_changes`, then null every bias.  `hip_pools` swaps plain
    nn.AvgPool2d modules for the HIP-backed subclass (bcos/modules/pooling.py) -- pass False when the modules
    are the reference's own (golden generation on CPU)."""
    for k, v in (model_config.get("standard_changes") or {}).items():
        setattr(model.model, k, v)
    if hip_pools:
        from bcos.modules.pooling import use_hip_pools
        use_hip_pools(model)
    for mod in model.modules():
        if hasattr(mod, "bias") and mod.bias is not None:
            mod.bias = None
    return model


def build_bcosified_resnet(arch: str = "resnet50", seed: int = 0):
    """Steps 1-3 with this package's modules (CPU tensors; move with .to('cuda') afterwards)."""
    from bcosify import BcosifyNetwork
    cfg = resnet_model_config(arch)
    net = BcosifyNetwork(standard_resnet(arch, seed), cfg, add_channels=True, logit_layer=True)
    finish_conversion(net, cfg)
    return net.eval()


def synthetic_images(n: int, seed: int = 123, size: int = 224, smooth: bool = True) -> torch.Tensor:
    """[n,6,size,size] AddInverse-encoded images in [0,1].  `smooth`: bilinearly upsampled 14x14 noise plus a
    per-image colour offset (pure pixel noise makes every image predict the same class)."""
    gen = torch.Generator().manual_seed(seed)
    if smooth:
        low = torch.rand(n, 3, 14, 14, generator=gen)
        x3 = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)
        x3 = (0.7 * x3 + 0.3 * torch.rand(n, 3, 1, 1, generator=gen)).clamp(0, 1)
    else:
        x3 = torch.rand(n, 3, size, size, generator=gen)
    return torch.cat([x3, 1 - x3], dim=1).contiguous()


def structured_images(n: int, seed: int = 77, size: int = 224) -> torch.Tensor:
    """[n,6,size,size] AddInverse-encoded images whose STRUCTURE is what noise fields lack (dynamic range inside an
    image -- sparse activations, long-tailed gradients in the explanation pass).  Image i is of kind i % 4:
    0 a smooth field (as synthetic_images), 1 a few small bright spots on black, 2 a bright disc with a sharp edge on a dark
    ground, 3 half black / half white with a faint fine checker texture.  Seeded; the same tensor on every host."""
    gen = torch.Generator().manual_seed(seed)
    ii = torch.arange(size).view(1, size, 1).float()
    jj = torch.arange(size).view(1, 1, size).float()
    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:
            low = torch.rand(1, 3, 14, 14, generator=gen)
            x3 = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)[0]
            x3 = (0.7 * x3 + 0.3 * torch.rand(3, 1, 1, generator=gen)).clamp(0, 1)
        elif kind == 1:
            x3 = torch.zeros(3, size, size)
            for _ in range(5):
                cy, cx = (torch.rand(2, generator=gen) * (size - 16) + 8).tolist()
                col = torch.rand(3, 1, 1, generator=gen) * 0.5 + 0.5
                spot = torch.exp(-((ii - cy) ** 2 + (jj - cx) ** 2) / (2 * 2.5 ** 2))
                x3 = torch.maximum(x3, col * spot)
        elif kind == 2:
            cy, cx = (torch.rand(2, generator=gen) * size * 0.4 + size * 0.3).tolist()
            disc = (((ii - cy) ** 2 + (jj - cx) ** 2) <= (size * 0.22) ** 2).float()
            col = torch.rand(3, 1, 1, generator=gen) * 0.3 + 0.7
            x3 = 0.02 + disc * (col - 0.02)
        else:
            half = (jj >= size // 2).float().expand(1, size, size)
            checker = (((ii.long() // 2 + jj.long() // 2) % 2).float() - 0.5) * 0.01
            x3 = (half * 0.98 + 0.01 + checker).expand(3, size, size).clamp(0, 1)
        out.append(torch.cat([x3, 1 - x3], dim=0))
    return torch.stack(out).contiguous()


def _is_bcos_conv(m) -> bool:
    """a B-cos conv or linear layer of either implementation (duck-typed: `.linear` + `.b`)"""
    return hasattr(m, "linear") and hasattr(m, "b") and isinstance(getattr(m, "linear", None), (nn.Conv2d, nn.Linear))


def _is_bnu(m) -> bool:
    return type(m).__name__.startswith("BatchNormUncentered2d")


@torch.no_grad()
def calibrate(net: nn.Module, x: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
    """Step 4, executed: returns the record {module name: conv gain (0-d) | bn running_var [C]}."""
    record: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    names = {m: n for n, m in net.named_modules()}
    hooks = []

    # On a HIP device every statistic is summed in a FIXED order by this repo's own kernel (ops.channel_moments_ordered): replicas
    # that calibrate independently must end bit-identical, and torch's multi-block reductions were measured not to be reproducible
    # when several processes time-slice one device -- on a settled tensor, behind a device synchronisation, with nothing of this
    # repo in flight (profiles/r04_var_triage.txt).  CPU tensors (the fixture generator, the oracle) keep torch.
    def _moments(t):
        if t.is_cuda and t.dim() >= 2 and t.dtype == torch.float32:
            from . import ops
            return ops.channel_moments_ordered(t)
        return None

    def conv_pre(mod, args):
        y = mod(*args)          # re-entrancy guarded below
        mom = _moments(y)
        if mom is not None:
            gain = torch.tensor(max(mom[2], 0.0) ** 0.5, dtype=torch.float32).clamp_min(1e-30).pow(-0.5).to(y.device)
        else:
            gain = y.pow(2).mean().sqrt().clamp_min(1e-30).pow(-0.5)
        mod.linear.weight.mul_(gain)    # in place on the Parameter (bumps its version: kernel-layout caches refresh)
        record[names[mod]] = gain.detach().cpu()

    def bn_pre(mod, args):
        mom = _moments(args[0])
        var = mom[1] if mom is not None else args[0].var(dim=(0, 2, 3), unbiased=False)
        mod.running_var.copy_(var)
        record[names[mod]] = var.detach().cpu()

    busy = set()

    def guard(fn):
        def wrapped(mod, args):
            if mod in busy:
                return None
            busy.add(mod)
            try:
                fn(mod, args)
            finally:
                busy.discard(mod)
            return None
        return wrapped

    for m in net.modules():
        if _is_bcos_conv(m):
            hooks.append(m.register_forward_pre_hook(guard(conv_pre)))
        elif _is_bnu(m):
            hooks.append(m.register_forward_pre_hook(guard(bn_pre)))
    try:
        net(x)
    finally:
        for h in hooks:
            h.remove()
    return record


@torch.no_grad()
def apply_calibration(net: nn.Module, record: Dict[str, torch.Tensor]):
    """Step 4, replayed from a record produced by `calibrate` (possibly with another implementation)."""
    mods = dict(net.named_modules())
    for name, val in record.items():
        m = mods[name]
        val = torch.as_tensor(val)
        if _is_bcos_conv(m):
            m.linear.weight.mul_(val.to(m.linear.weight.device, m.linear.weight.dtype))
        else:
            m.running_var.copy_(val.to(m.running_var.device))
    return net


# ---- SimpleViT (BASELINE.json configs[2]) -------------------------------------------------------------------
def vit_model_config(arch: str = "simple_vit_ti_patch16_224") -> dict:
    """CONFIGS['bcosifyv2_<arch>_..._gapReorder'] model section (vit_bcosification/experiment_parameters.py:129-215)."""
    return dict(
        is_bcos=True,
        name=arch,
        weights="pretrained",
        args=dict(num_classes=1000, channels=6, gap_reorder=True),
        bcos_args=dict(b=2, max_out=1),
        bcosify_args=dict(fix_b=True, use_bias=False),
        logit_layer=True,
        act_layer=True,
        logit_bias=math.log(1 / (1000 - 1)),
    )


def standard_vit(arch: str, seed: int = 0, vit_module=None):
    """The standard (non-B-cos) SimpleViT that the recipe starts from (vit_final/model.py:21-46 with nn.Linear,
    nn.LayerNorm, nn.GELU, 3 input channels).  `vit_module` lets the golden generator pass the reference's module."""
    if vit_module is None:
        from bcos.models import vit as vit_module
    torch.manual_seed(seed)
    return getattr(vit_module, arch)(channels=3, linear_layer=nn.Linear, norm_layer=nn.LayerNorm, act_layer=nn.GELU,
                                     conv2d_layer=nn.Conv2d, norm2d_layer=lambda c: nn.GroupNorm(1, c))


def finish_vit_conversion(model: nn.Module, model_config: dict):
    """vit_bcosification/model.py:19-29: null the biases unless use_bias, set gap_reorder."""
    if not model_config.get("bcosify_args", {}).get("use_bias", False):
        for mod in model.modules():
            if hasattr(mod, "bias") and mod.bias is not None:
                mod.bias = None
    if model_config["args"].get("gap_reorder", False):
        model.model.gap_reorder = True
    return model


def build_bcosified_vit(arch: str = "simple_vit_ti_patch16_224", seed: int = 0):
    from bcosify_vit import BcosifyNetwork
    cfg = vit_model_config(arch)
    net = BcosifyNetwork(standard_vit(arch, seed), cfg, add_channels=True, logit_layer=cfg["logit_layer"])
    return finish_vit_conversion(net, cfg).eval()


# ---- CLIP RN50 image encoder (BASELINE.json configs[3]) -------------------------------------------------------
def clip_model_config() -> dict:
    """CONFIGS['resnet_50_clip_b2_noBias_..._bcosification'] model section
    (clip_bcosification/experiment_parameters.py:41-106)."""
    return dict(
        is_bcos=True,
        name="resnet50clip",
        weights="clip",
        bcos_args=dict(b=2, max_out=1),
        bcosify_args=dict(clip_kd=True, fix_b=True, norm_layer="BnUncV2", use_bias=False),
    )


def standard_clip_rn50(seed: int = 0, clip_module=None):
    """CLIP RN50 vision tower (layers (3,4,6,3), width 64, 32 heads, 1024-d output) with default init and randomised
    BatchNorm statistics; `clip_module` lets the golden generator pass the reference's CLIP/clip/model.py."""
    if clip_module is None:
        from CLIP.clip import model as clip_module
    torch.manual_seed(seed)
    net = clip_module.ModifiedResNet([3, 4, 6, 3], 1024, 32, input_resolution=224, width=64)
    randomize_batchnorm(net, torch.Generator().manual_seed(seed + 1))
    return net


def standard_clip_resnet(layers, output_dim, heads, width, seed: int = 0, clip_module=None, input_resolution: int = 64):
    """A ModifiedResNet (CLIP/clip/model.py:94-154) of any depth / width with default init and randomised BatchNorm statistics -- the
    small instance behind the tight training-step fixture (tests/golden/clip_tiny_train_step.*)."""
    if clip_module is None:
        from CLIP.clip import model as clip_module
    torch.manual_seed(seed)
    net = clip_module.ModifiedResNet(list(layers), output_dim, heads, input_resolution=input_resolution, width=width)
    randomize_batchnorm(net, torch.Generator().manual_seed(seed + 1))
    return net


def build_bcosified_clip_resnet(layers, output_dim, heads, width, seed: int = 0):
    from bcosify import BcosifyNetwork
    net = BcosifyNetwork(standard_clip_resnet(layers, output_dim, heads, width, seed), clip_model_config(), add_channels=True, logit_layer=False)
    return finish_clip_conversion(net).eval()


def finish_clip_conversion(model: nn.Module, hip_pools: bool = True):
    """clip_bcosification/model.py:17-23: null every bias and the attention pool's positional embedding."""
    for mod in model.modules():
        if hasattr(mod, "bias") and mod.bias is not None:
            mod.bias = None
        if hasattr(mod, "positional_embedding") and mod.positional_embedding is not None:
            mod.positional_embedding = None
    if hip_pools:
        from bcos.modules.pooling import use_hip_pools
        use_hip_pools(model)
    return model


def build_bcosified_clip_rn50(seed: int = 0, attn_unpool: bool = False):
    """`attn_unpool`: the head variant that keeps every location (model_config['attn_unpool'], bcosattnpool.py:61-77)."""
    from bcosify import BcosifyNetwork
    cfg = dict(clip_model_config(), attn_unpool=True) if attn_unpool else clip_model_config()
    net = BcosifyNetwork(standard_clip_rn50(seed), cfg, add_channels=True, logit_layer=False)
    return finish_clip_conversion(net).eval()
