"""CLIP "ModifiedResNet" image-encoder topology, built from a stage table.

What has to match the reference's vendored OpenAI file (CLIP/clip/model.py:10-154) is the STATE-DICT CONTRACT, because
B-cosified CLIP checkpoints must load unchanged: module names conv{1,2,3} / bn{1,2,3} / relu{1,2,3} /
avgpool / layer{1..4}.<i>.{conv,bn,relu}{1,2,3} / downsample."-1","0","1" / attnpool.{q,k,v,c}_proj, and the tensor
shapes behind them.  The forward passes below are written against that naming scheme (units are looked up by index),
not transcribed; the plain attention pool never runs in this package -- `bcosify.py` replaces it by
BcosAttentionPool2d (clip_kd configuration) before any forward -- so it only carries its parameters.
"""
from collections import OrderedDict

import torch
from torch import nn

# (unit index, kernel size, stride, output width as a multiple of `planes`) of the three conv/bn/relu units of a block;
# the strided variant keeps every convolution at stride 1 and average-pools instead (anti-aliasing)
_BLOCK_UNITS = ((1, 1, 1), (2, 3, 1), (3, 1, 4))


def _add_unit(module: nn.Module, idx: int, cin: int, cout: int, k: int, stride: int = 1):
    """registers conv<idx> / bn<idx> / relu<idx> on `module`"""
    setattr(module, f"conv{idx}", nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False))
    setattr(module, f"bn{idx}", nn.BatchNorm2d(cout))
    setattr(module, f"relu{idx}", nn.ReLU(inplace=True))


def _run_unit(module: nn.Module, idx: int, x: torch.Tensor, act: bool = True) -> torch.Tensor:
    x = getattr(module, f"bn{idx}")(getattr(module, f"conv{idx}")(x))
    return getattr(module, f"relu{idx}")(x) if act else x


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        width_in = inplanes
        for idx, k, mult in _BLOCK_UNITS:
            _add_unit(self, idx, width_in, planes * mult, k)
            width_in = planes * mult
            if idx == 2:        # the stride of the block, applied as a pool between units 2 and 3
                self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != width_in:
            self.downsample = nn.Sequential(OrderedDict((("-1", nn.AvgPool2d(stride)),
                                                         ("0", nn.Conv2d(inplanes, width_in, 1, stride=1, bias=False)),
                                                         ("1", nn.BatchNorm2d(width_in)))))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shortcut = x if self.downsample is None else self.downsample(x)
        y = _run_unit(self, 2, _run_unit(self, 1, x))
        y = _run_unit(self, 3, self.avgpool(y), act=False)
        return self.relu3(y + shortcut)


class AttentionPool2d(nn.Module):
    """CLIP's QKV attention pool (positional embedding + q/k/v/c projections with biases).  `bcosify.BcosifyNetwork` turns it
    into `bcos.modules.BcosAttentionPool2d` (the hot-path form) when `bcosify_args['clip_kd']` is set; the plain pool stays
    usable for every other CLIP configuration and for the un-converted image encoder (the distillation teacher)
    -- reference CLIP/clip/model.py:58-92, bcosify.py:80 -- as a handful of torch ops: it is outside the B-cos hot path."""

    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int = None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        for name, width in (("k_proj", embed_dim), ("q_proj", embed_dim), ("v_proj", embed_dim), ("c_proj", output_dim or embed_dim)):
            setattr(self, name, nn.Linear(embed_dim, width))
        self.num_heads = num_heads

    def forward(self, x):
        n, c, h, w = x.shape
        heads, dh = self.num_heads, c // self.num_heads
        tok = x.reshape(n, c, h * w).transpose(1, 2)                                   # [N, HW, C]
        tok = torch.cat((tok.mean(1, keepdim=True), tok), 1) + self.positional_embedding.to(x.dtype)
        q = self.q_proj(tok[:, :1]).reshape(n, 1, heads, dh) * dh ** -0.5             # the mean token is the only query
        k = self.k_proj(tok).reshape(n, -1, heads, dh)
        v = self.v_proj(tok).reshape(n, -1, heads, dh)
        att = torch.einsum("nqhd,nkhd->nhqk", q, k).softmax(-1)
        return self.c_proj(torch.einsum("nhqk,nkhd->nqhd", att, v).reshape(n, c))


class ModifiedResNet(nn.Module):
    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.output_dim = output_dim
        self.input_resolution = input_resolution
        # stem: three 3x3 units (the first one strided) and a 2x2 average pool
        for idx, (cin, cout, stride) in enumerate(((3, width // 2, 2), (width // 2, width // 2, 1), (width // 2, width, 1)), start=1):
            _add_unit(self, idx, cin, cout, 3, stride)
        self.avgpool = nn.AvgPool2d(2)
        inplanes = width
        for stage, (planes, depth) in enumerate(zip((width, width * 2, width * 4, width * 8), layers), start=1):
            blocks = []
            for b in range(depth):
                blocks.append(Bottleneck(inplanes, planes, stride=2 if (b == 0 and stage > 1) else 1))
                inplanes = planes * Bottleneck.expansion
            setattr(self, f"layer{stage}", nn.Sequential(*blocks))
        self.attnpool = AttentionPool2d(input_resolution // 32, inplanes, heads, output_dim)

    def forward(self, x):
        x = x.type(self.conv1.weight.dtype)
        for idx in (1, 2, 3):
            x = _run_unit(self, idx, x)
        x = self.avgpool(x)
        for stage in (1, 2, 3, 4):
            x = getattr(self, f"layer{stage}")(x)
        return self.attnpool(x)
