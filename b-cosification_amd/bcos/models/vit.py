"""SimpleViT topology (Beyer et al., "Better plain ViT baselines for ImageNet-1k") with the reference's module and
state-dict names (bcos/models/vit.py:230-339): `to_patch_embedding.linear`, `transformer.encoder_N.attn.{norm,to_qkv,
to_out}`, `transformer.encoder_N.ff.net.{norm,linear1,act,linear2}`, `linear_head.{norm,linear}`; sin-cos positional
embedding (:64-86); optional classifier-before-mean ordering `gap_reorder` (:331-338).

`Attention` (:118-158) is the one piece that needs MI355X code of its own: q and k are detached in explanation mode, so
the softmax matrix is a constant and only v carries gradient; forward and that gradient are LDS-resident HIP kernels
(bcos_attention_fwd / _bwd_v), the plain `to_qkv` projection runs on the same fp32-MFMA GEMM as the B-cos layers.
"""
from collections import OrderedDict
from typing import Any, Callable, List, Tuple, Union

import torch
from torch import Tensor, nn
from torch.autograd import Function

from bcos.modules import _hipfn
from bcos.modules.common import DetachableModule
from bcos_hip import ops

__all__ = ["SimpleViT", "Attention", "PosEmbSinCos2d", "simple_vit_ti_patch16_224", "simple_vit_s_patch16_224",
           "simple_vit_b_patch16_224", "simple_vit_l_patch16_224", "vitc_ti_patch1_14", "vitc_s_patch1_14",
           "vitc_b_patch1_14", "vitc_l_patch1_14", "make_conv_stem"]


def pair(t: Any) -> Tuple[Any, Any]:
    return t if isinstance(t, tuple) else (t, t)


class PatchRearrange(nn.Module):
    """'b c (h p1) (w p2) -> b h w (p1 p2 c)' (einops Rearrange in the reference, :290-294)."""

    def __init__(self, p1: int, p2: int):
        super().__init__()
        self.p1, self.p2 = p1, p2

    def forward(self, x: Tensor) -> Tensor:
        b, c, hh, ww = x.shape
        h, w = hh // self.p1, ww // self.p2
        x = x.reshape(b, c, h, self.p1, w, self.p2).permute(0, 2, 4, 3, 5, 1)
        return x.reshape(b, h, w, self.p1 * self.p2 * c)


class PosEmbSinCos2d(nn.Module):
    """Fixed 2-d sine / cosine position table of SimpleViT (reference bcos/models/vit.py:64-86): for patch (i, j) of an h x w
    grid and frequencies w_k = T^(-k / (dim/4 - 1)), k < dim/4, the row [sin(j w), cos(j w), sin(i w), cos(i w)].
    The table depends on (h, w, dim, T) only: it is built once per geometry / device and kept, so a forward pass launches
    nothing for it (the reference rebuilds it -- meshgrid, pow, sin, cos, cat -- on every call)."""

    def __init__(self, temperature: Union[int, float] = 10_000):
        super().__init__()
        self.temperature = temperature
        self._tables = {}

    def forward(self, patches: Tensor) -> Tensor:
        h, w, dim = (int(v) for v in patches.shape[-3:])
        key = (h, w, dim, float(self.temperature), str(patches.device), patches.dtype)
        table = self._tables.get(key)
        if table is None:
            if dim % 4:
                raise AssertionError("feature dimension must be multiple of 4 for sincos emb")
            quarter = dim // 4
            freq = 1.0 / (self.temperature ** (torch.arange(quarter) / (quarter - 1)))      # w_k, evaluated as the reference does
            col = torch.arange(w)[:, None] * freq[None, :]                              # [w, dim/4]: j w_k
            row = torch.arange(h)[:, None] * freq[None, :]                              # [h, dim/4]: i w_k
            table = torch.empty((h, w, 4, quarter), dtype=torch.float32)
            table[:, :, 0], table[:, :, 1] = col.sin()[None], col.cos()[None]
            table[:, :, 2], table[:, :, 3] = row.sin()[:, None], row.cos()[:, None]
            table = table.reshape(h * w, dim).to(device=patches.device, dtype=patches.dtype)
            self._tables[key] = table
        return table


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, linear_layer=None, norm_layer=None, act_layer=None):
        assert linear_layer is not None and norm_layer is not None and act_layer is not None
        super().__init__()
        self.net = nn.Sequential(OrderedDict(norm=norm_layer(dim), linear1=linear_layer(dim, hidden_dim), act=act_layer(),
                                             linear2=linear_layer(hidden_dim, dim)))

    def forward(self, x: Tensor) -> Tensor:
        return self.net(x)


class _AttentionCoreFn(Function):
    """softmax(q k^T scale) v on packed qkv [B,T,3*inner]; explanation-mode backward feeds v only."""

    @staticmethod
    def forward(ctx, qkv, heads, scale, detach):
        _hipfn.require_hip(qkv, "Attention")
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        need = ctx.needs_input_grad[0]
        out, stats = ops.attention_fwd(qkv, heads, scale, want_stats=need)
        ctx.cfg = (heads, scale, detach)
        if need:
            ctx.save_for_backward(qkv, stats, *([] if detach else [out]))
        return out

    @staticmethod
    def backward(ctx, gout):
        heads, scale, detach = ctx.cfg
        gout = gout if gout.is_contiguous() else gout.contiguous()
        if not detach:          # training mode: q, k and v all receive gradients (csrc/bcos_vit.hip: attention_bwd_full_kernel)
            qkv, stats, out = ctx.saved_tensors
            return ops.attention_bwd(qkv, stats, out, gout, heads, scale), None, None, None
        qkv, stats = ctx.saved_tensors
        gv = ops.attention_bwd_v(qkv, stats, gout if gout.is_contiguous() else gout.contiguous(), heads, scale)
        gqkv = torch.zeros_like(qkv)
        inner = gv.shape[-1]
        gqkv[..., 2 * inner:] = gv
        return gqkv, None, None, None


class Attention(DetachableModule):
    def __init__(self, dim, heads=8, dim_head=64, linear_layer=None, norm_layer=None):
        assert linear_layer is not None and norm_layer is not None
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head**-0.5
        self.norm = norm_layer(dim)
        self.attend = nn.Softmax(dim=-1)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = linear_layer(inner_dim, dim, bias=False)
        self._qkv_cache = _hipfn.WeightCache()

    def forward(self, x: Tensor) -> Tensor:
        x = self.norm(x)
        qkv = _hipfn.plain_linear(x, self.to_qkv.weight, self.to_qkv.bias, self._qkv_cache, self.to_qkv.weight)
        out = _AttentionCoreFn.apply(qkv, self.heads, self.scale, self.detach)
        return self.to_out(out)


class Encoder(nn.Module):
    def __init__(self, dim, heads, dim_head, mlp_dim, linear_layer=None, norm_layer=None, act_layer=None):
        super().__init__()
        self.attn = Attention(dim, heads=heads, dim_head=dim_head, linear_layer=linear_layer, norm_layer=norm_layer)
        self.ff = FeedForward(dim, mlp_dim, linear_layer=linear_layer, norm_layer=norm_layer, act_layer=act_layer)

    def forward(self, x: Tensor) -> Tensor:
        x = self.attn(x) + x
        x = self.ff(x) + x
        return x


class Transformer(nn.Sequential):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, linear_layer=None, norm_layer=None, act_layer=None):
        layers = OrderedDict()
        for i in range(depth):
            layers[f"encoder_{i}"] = Encoder(dim=dim, heads=heads, dim_head=dim_head, mlp_dim=mlp_dim,
                                             linear_layer=linear_layer, norm_layer=norm_layer, act_layer=act_layer)
        super().__init__(layers)


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, channels: int = 6,
                 linear_layer: Callable[..., nn.Module] = None, norm_layer: Callable[..., nn.Module] = None,
                 act_layer: Callable[..., nn.Module] = None, norm2d_layer=None, conv2d_layer=None,
                 conv_stem: List[int] = None, gap_reorder: bool = False, **kwargs):
        super().__init__()
        _ = kwargs
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(patch_size)
        assert linear_layer is not None and norm_layer is not None and act_layer is not None
        if conv_stem:
            assert conv2d_layer is not None, "Provide a conv2d layer class when using conv_stem!"
            assert norm2d_layer is not None, "Provide a norm2d layer class when using conv_stem!"
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            "Image dimensions must be divisible by the patch size."
        self.image_size, self.patch_size = (image_height, image_width), (patch_height, patch_width)
        self.num_patches = (image_height // patch_height) * (image_width // patch_width)
        self.patch_dim = (channels if not conv_stem else conv_stem[-1]) * patch_height * patch_width
        stem = OrderedDict(conv_stem=make_conv_stem(channels, conv_stem, conv2d_layer, norm2d_layer, act_layer)) if conv_stem \
            else OrderedDict()
        self.to_patch_embedding = nn.Sequential(OrderedDict(**stem, rearrage=PatchRearrange(patch_height, patch_width),
                                                            linear=linear_layer(self.patch_dim, dim)))
        self.positional_embedding = PosEmbSinCos2d()
        dim_head = dim // heads
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, linear_layer=linear_layer,
                                       norm_layer=norm_layer, act_layer=act_layer)
        self.to_latent = nn.Identity()
        self.linear_head = nn.Sequential(OrderedDict(norm=norm_layer(dim), linear=linear_layer(dim, num_classes)))
        self.gap_reorder = gap_reorder

    def forward(self, img):
        x = self.to_patch_embedding(img)
        pe = self.positional_embedding(x)
        x = x.flatten(1, -2) + pe
        x = self.transformer(x)
        if self.gap_reorder:
            return self.linear_head(self.to_latent(x)).mean(dim=1)
        return self.linear_head(self.to_latent(x.mean(dim=1)))


def make_conv_stem(in_channels: int, out_channels: List[int], conv2d_layer=None, norm2d_layer=None, act_layer=None):
    """conv 3x3 (stride 2 whenever the width grows) + 2-D norm + activation per entry ("Early convolutions help
    transformers see better"; reference :342-365)."""
    layers = []
    for outc in out_channels:
        layers += [conv2d_layer(in_channels, outc, kernel_size=3, stride=(2 if outc > in_channels else 1), padding=1),
                   norm2d_layer(outc), act_layer()]
        in_channels = outc
    return nn.Sequential(*layers)


def _vitc(dim, depth, heads, mlp_dim, conv_stem, **kwargs):
    """conv-stem ViTs: the stem takes 224 x 224 images to a 14 x 14 map that is tokenised with patch size 1; depth is one
    block less than the plain model's (reference :368-426)."""
    kwargs.setdefault("num_classes", 1_000)
    return SimpleViT(image_size=14, patch_size=1, dim=dim, depth=depth, heads=heads, mlp_dim=mlp_dim, conv_stem=conv_stem, **kwargs)


def vitc_ti_patch1_14(**kwargs):
    return _vitc(192, 11, 3, 768, [24, 48, 96, 192], **kwargs)


def vitc_s_patch1_14(**kwargs):
    return _vitc(384, 11, 6, 1536, [48, 96, 192, 384], **kwargs)


def vitc_b_patch1_14(**kwargs):
    return _vitc(768, 11, 12, 3072, [64, 128, 128, 256, 256, 512], **kwargs)


def vitc_l_patch1_14(**kwargs):
    return _vitc(1024, 13, 16, 4096, [64, 128, 128, 256, 256, 512], **kwargs)


def _simple_vit(dim, depth, heads, mlp_dim, **kwargs):
    kwargs.setdefault("num_classes", 1_000)
    return SimpleViT(image_size=224, patch_size=16, dim=dim, depth=depth, heads=heads, mlp_dim=mlp_dim, **kwargs)


def simple_vit_ti_patch16_224(**kwargs):
    return _simple_vit(192, 12, 3, 768, **kwargs)


def simple_vit_s_patch16_224(**kwargs):
    return _simple_vit(384, 12, 6, 1536, **kwargs)


def simple_vit_b_patch16_224(**kwargs):
    return _simple_vit(768, 12, 12, 3072, **kwargs)


def simple_vit_l_patch16_224(**kwargs):
    return _simple_vit(1024, 14, 16, 4096, **kwargs)
