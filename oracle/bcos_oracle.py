"""CPU oracle for the B-cos forward / explanation hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-fp32-CPU restatement of the reference's
algorithm (shrebox/B-cosification, paths relative to the reference root).  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it -- as the checker or as the
timed CPU baseline, never as part of the product path (b-cosification_amd/ fails loudly without its
HIP library instead of falling back to this).

Parity pinning: the reference ships NO tests / golden vectors for this path, so
this restatement is pinned against outputs of the reference itself, imported in the build
container by oracle/refimport.py; tests/golden/make_golden.py records those outputs (and the
oracle-vs-reference differences it measured) as fixtures under tests/golden/.

Everything is functional and works on state dicts with the reference's key layout
(SURVEY.md section 8 T3), so the same state dict drives the oracle and the HIP product.
"""
import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# bcosify.py:15-20
IMAGENET_MEAN_ADDINVERSE = (0.485, 0.456, 0.406, 0.515, 0.544, 0.594)
IMAGENET_STD_ADDINVERSE = (0.229, 0.224, 0.225, 0.229, 0.224, 0.225)
CLIP_MEAN_ADDINVERSE = (0.48145466, 0.4578275, 0.40821073, 0.51854534, 0.5421725, 0.59178927)
CLIP_MEAN_ZERO = (0.0,) * 6
CLIP_STD_ADDINVERSE = (0.26862954, 0.26130258, 0.27577711, 0.26862954, 0.26130258, 0.27577711)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# ----------------------------------------------------------------------------------------------
# layer level
# ----------------------------------------------------------------------------------------------
def add_inverse(x: torch.Tensor) -> torch.Tensor:
    """bcos/data/transforms.py:54-55"""
    return torch.cat([x, 1 - x], dim=-3)


def normalize6(x: torch.Tensor, mean: Sequence[float], std: Sequence[float]) -> torch.Tensor:
    """torchvision.transforms.Normalize as used at bcosify.py:38-43,51-53 (clone, sub_, div_)."""
    m = torch.tensor(mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
    s = torch.tensor(std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
    return (x - m) / s


def unit_norm_weight(w: torch.Tensor) -> torch.Tensor:
    """NormedConv2d / NormedLinear: bcosconv2d.py:28-29, bcoslinear.py:26."""
    dims = tuple(range(1, w.dim()))
    return w / torch.linalg.vector_norm(w, dim=dims, keepdim=True)


def patch_norm(x, kernel_size, stride, padding, groups=1, out_channels=None):
    """BcosConv2d.calc_patch_norms, bcosconv2d.py:196-231 (dilation 1)."""
    sq = x * x
    if groups == 1:
        sq = sq.sum(1, keepdim=True)
    else:
        sq = sq.unflatten(1, (groups, x.shape[1] // groups)).sum(2)
    n = (F.avg_pool2d(sq, kernel_size, padding=padding, stride=stride, divisor_override=1) + 1e-6).sqrt()
    if groups > 1:
        n = torch.repeat_interleave(n, repeats=out_channels // groups, dim=1)
    return n


def patch_norm_slow(x, weight_shape, stride, padding, dilation, groups):
    """BcosConv2d._calc_patch_norms_slow, bcosconv2d.py:233-250 (any dilation)."""
    ones = torch.ones(weight_shape, dtype=x.dtype, device=x.device)
    return (F.conv2d(x * x, ones, None, stride, padding, dilation, groups) + 1e-6).sqrt()


def _scale_and_apply(out, norm, b, detach):
    """bcosconv2d.py:176-194 / bcoslinear.py:115-130."""
    src = out.detach() if detach else out
    if detach:
        norm = norm.detach()
    if b == 2:
        s = src.abs() / norm
    else:
        s = ((src / norm).abs() + 1e-6).pow(b - 1)
    return s * out, s


def bcos_conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, b=2, max_out=1,
                detach=False, normalize_weight=False, weight_gain=None, return_scale=False):
    """BcosConv2d.forward_impl (bcosconv2d.py:153-194) / BcosifyConv2d.forward_impl (bcosifyconv2d.py:50-102).

    normalize_weight=True  -> native BcosConv2d (NormedConv2d unit-norm weights, optional `scale`, bcosconv2d.py:26-35)
    normalize_weight=False -> BcosifyConv2d (plain nn.Conv2d with optional bias)
    """
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    w = weight
    if normalize_weight:
        w = unit_norm_weight(w)
        if weight_gain is not None:
            w = weight_gain * w
    out = F.conv2d(x, w, bias, stride, padding, dilation, groups)
    if max_out > 1:
        out = out.unflatten(1, (out.shape[1] // max_out, max_out)).max(dim=2).values
    if b == 1:
        return (out, None) if return_scale else out
    kh, kw = weight.shape[2], weight.shape[3]
    if any(d > 1 for d in dilation):
        norm = patch_norm_slow(x, weight.shape, stride, padding, dilation, groups)
        if max_out > 1:  # ones-kernel conv gives Cout*max_out identical maps per group
            norm = norm[:, ::max_out]
    else:
        norm = patch_norm(x, (kh, kw), stride, padding, groups, out.shape[1])
    y, s = _scale_and_apply(out, norm, b, detach)
    return (y, s) if return_scale else y


def bcos_linear(x, weight, bias=None, b=2, max_out=1, detach=False, normalize_weight=False, return_scale=False):
    """BcosLinear.forward (bcoslinear.py:88-130) / BcosifyLinear.forward (bcosifylinear.py:42-95)."""
    w = unit_norm_weight(weight) if normalize_weight else weight
    out = F.linear(x, w, bias)
    if max_out > 1:
        out = out.unflatten(-1, (out.shape[-1] // max_out, max_out)).max(dim=-1).values
    if b == 1:
        return (out, None) if return_scale else out
    norm = torch.linalg.vector_norm(x, dim=-1, keepdim=True) + 1e-12
    y, s = _scale_and_apply(out, norm, b, detach)
    return (y, s) if return_scale else y


def bn_uncentered_eval(x, running_var, weight=None, bias=None, eps=1e-5):
    """batch_norm_uncentered_2d, eval branch: batchnorm_uncentered.py:46-60."""
    std = (running_var + eps).sqrt()[None, :, None, None]
    r = x / std
    if weight is not None:
        r = weight[None, :, None, None] * r
    if bias is not None:
        r = r + bias[None, :, None, None]
    return r.type(x.dtype)


def bn_uncentered_fold(bn_weight, bn_bias, running_mean, running_var, eps, norm_layer="BnUncV2"):
    """BatchNormUncentered2d.from_standard_module: batchnorm_uncentered.py:117-141 (returns weight, bias)."""
    if bn_bias is not None and norm_layer == "BnUncV2":
        std = (running_var + eps).sqrt()
        return bn_weight, bn_bias - (running_mean / std) * bn_weight
    return bn_weight, bn_bias


def layer_norm_detachable(x, normalized_shape, weight, bias, eps=1e-5, detach=False):
    """DetachableLayerNorm.forward: centered_norms.py:197-224."""
    if not detach:
        return F.layer_norm(x, normalized_shape, weight, bias, eps)
    d = len(normalized_shape)
    var, mean = torch.var_mean(x, dim=tuple(range(-d, 0)), unbiased=False, keepdim=True)
    std = (var.detach() + eps).sqrt()
    y = (x - mean) / std
    if weight is not None:
        y = weight * y
    if bias is not None:
        y = y + bias
    return y


def gelu_detachable(x, detach=False):
    """MyGELU.forward: bcosify_vit.py:27-32."""
    gate = 0.5 * (1 + torch.erf(x / np.sqrt(2)))
    if detach:
        gate = gate.detach()
    return gate * x


def logit_layer(x, temperature=None, bias=None):
    """LogitLayer.forward: bcos/modules/logitlayer.py:22-27."""
    if temperature is not None:
        x = x / temperature
    if bias is not None:
        x = x + bias
    return x


# ----------------------------------------------------------------------------------------------
# B-cosified ResNet (torchvision topology, classifier before GAP: bcos/models/standard_models.py:36-54,
# conversion bcosify.py:74-114, factory bcos/experiments/ImageNet/bcosification/model.py:15-57)
# ----------------------------------------------------------------------------------------------
RESNET_SPECS = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
}


def _bnu(sd, prefix, x):
    return bn_uncentered_eval(x, sd[prefix + ".running_var"], sd.get(prefix + ".weight"), sd.get(prefix + ".bias"),
                              eps=1e-5)


def _bconv(sd, prefix, x, stride, padding, b, detach):
    return bcos_conv2d(x, sd[prefix + ".linear.weight"], sd.get(prefix + ".linear.bias"), stride=stride,
                       padding=padding, b=b, detach=detach)


def resnet_features(sd: Dict[str, torch.Tensor], xn: torch.Tensor, arch: str, b=2, detach=False, prefix="model.",
                    stem_pool="avg", taps: Optional[dict] = None, gate_log: Optional[list] = None):
    """Body of ResNetBcos._forward_impl up to and including `fc` (a 1x1 B-cos conv), on the normalised input.
    `gate_log`, if given, receives the pre-activation of every ReLU in execution order."""
    kind, blocks = RESNET_SPECS[arch]

    def relu(t):
        if gate_log is not None:
            gate_log.append(t.detach())
        return F.relu(t)

    x = _bconv(sd, prefix + "conv1", xn, 2, 3, b, detach)
    x = relu(_bnu(sd, prefix + "bn1", x))
    if stem_pool == "avg":   # standard_changes: maxpool -> nn.AvgPool2d(3, 2, 1)   (experiment_parameters.py:99)
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=1)
    else:
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, nblocks in enumerate(blocks, start=1):
        for bi in range(nblocks):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            identity = x
            if kind == "basic":
                out = relu(_bnu(sd, p + "bn1", _bconv(sd, p + "conv1", x, stride, 1, b, detach)))
                out = _bnu(sd, p + "bn2", _bconv(sd, p + "conv2", out, 1, 1, b, detach))
            else:  # torchvision v1.5: stride on the 3x3
                out = relu(_bnu(sd, p + "bn1", _bconv(sd, p + "conv1", x, 1, 0, b, detach)))
                out = relu(_bnu(sd, p + "bn2", _bconv(sd, p + "conv2", out, stride, 1, b, detach)))
                out = _bnu(sd, p + "bn3", _bconv(sd, p + "conv3", out, 1, 0, b, detach))
            if (p + "downsample.0.linear.weight") in sd:
                identity = _bnu(sd, p + "downsample.1", _bconv(sd, p + "downsample.0", x, stride, 0, b, detach))
            x = relu(out + identity)
            if taps is not None:
                taps[f"layer{li}.{bi}"] = x
    x = _bconv(sd, prefix + "fc", x, 1, 0, b, detach)
    return x


def resnet_logits(sd, x6, arch, b=2, detach=False, mean=IMAGENET_MEAN_ADDINVERSE, std=IMAGENET_STD_ADDINVERSE,
                  logit_bias=-math.log(1000 - 1), logit_temperature=None, stem_pool="avg", gate_log=None):
    """BcosifyNetwork.forward (bcosify.py:50-53) for a B-cosified torchvision ResNet with logit layer."""
    xn = normalize6(x6, mean, std)
    f = resnet_features(sd, xn, arch, b=b, detach=detach, stem_pool=stem_pool, gate_log=gate_log)
    logits = F.adaptive_avg_pool2d(f, 1).flatten(1)
    return logit_layer(logits, logit_temperature, logit_bias)


# ----------------------------------------------------------------------------------------------
# explanation (bcos/common.py:92-188, 280-317; interpretability/explanation_methods/explainers/captum.py:29-32)
# ----------------------------------------------------------------------------------------------
def explain_batch(forward_fn, x: torch.Tensor, targets: Optional[torch.Tensor] = None):
    """Batched form of BcosUtilMixin.explain: forward in explanation mode, backward of the chosen logit.

    forward_fn(x, detach=True) -> logits [N,K].  Returns dict(logits, prediction, explained_class_idx,
    dynamic_linear_weights [N,C,H,W], contribution_map [N,H,W]).  Samples are independent in eval mode,
    so back-propagating the sum of the selected logits equals N separate explain() calls."""
    x = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        logits = forward_fn(x, detach=True)
        pred = logits.max(1)
        idx = pred.indices if targets is None else targets
        sel = logits.gather(1, idx.view(-1, 1)).sum()
        (grad,) = torch.autograd.grad(sel, x)
    return dict(logits=logits.detach(), prediction=pred.indices, explained_class_idx=idx,
                dynamic_linear_weights=grad, contribution_map=(x.detach() * grad).sum(1))


def gradient_to_image(image, linear_mapping, smooth=15, alpha_percentile=99.5):
    """bcos/common.py:387-436 -> RGBA [H,W,4] numpy."""
    contribs = (image * linear_mapping).sum(0, keepdim=True)
    rgb = linear_mapping / (linear_mapping.abs().max(0, keepdim=True).values + 1e-12)
    rgb = rgb.clamp(min=0)
    rgb = rgb[:3] / (rgb[:3] + rgb[3:] + 1e-12)
    alpha = linear_mapping.norm(p=2, dim=0, keepdim=True)
    alpha = torch.where(contribs < 0, 1e-12, alpha)
    if smooth:
        alpha = F.avg_pool2d(alpha, smooth, stride=1, padding=(smooth - 1) // 2)
    alpha = (alpha / torch.quantile(alpha, q=alpha_percentile / 100)).clip(0, 1)
    return torch.cat([rgb, alpha], dim=0).permute(1, 2, 0).detach().cpu().numpy()


# ----------------------------------------------------------------------------------------------
# B-cosified SimpleViT (bcos/models/vit.py:230-339 converted by bcosify_vit.py:45-154; factory
# bcos/experiments/ImageNet/vit_bcosification/model.py:7-31)
# ----------------------------------------------------------------------------------------------
def posemb_sincos_2d(h, w, dim, temperature=10_000, dtype=torch.float32):
    """PosEmbSinCos2d.forward: vit.py:69-86."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).type(dtype)


def vit_attention(sd, p, x, heads, b, detach):
    """Attention.forward: vit.py:143-158 (q, k detached in explanation mode; to_qkv is a plain Linear)."""
    dim = x.shape[-1]
    h = layer_norm_detachable(x, (dim,), sd[p + "norm.weight"], sd.get(p + "norm.bias"), 1e-5, detach)
    qkv = F.linear(h, sd[p + "to_qkv.weight"]).chunk(3, dim=-1)
    B, T, inner = qkv[0].shape
    q, k, v = (t.view(B, T, heads, inner // heads).transpose(1, 2) for t in qkv)
    if detach:
        q, k = q.detach(), k.detach()
    dots = torch.matmul(q, k.transpose(-1, -2)) * (inner // heads) ** -0.5
    out = torch.matmul(dots.softmax(dim=-1), v).transpose(1, 2).reshape(B, T, inner)
    return bcos_linear(out, sd[p + "to_out.linear.weight"], sd.get(p + "to_out.linear.bias"), b=b, detach=detach)


def vit_feedforward(sd, p, x, b, detach, gelu=True):
    dim = x.shape[-1]
    h = layer_norm_detachable(x, (dim,), sd[p + "norm.weight"], sd.get(p + "norm.bias"), 1e-5, detach)
    h = bcos_linear(h, sd[p + "linear1.linear.weight"], sd.get(p + "linear1.linear.bias"), b=b, detach=detach)
    if gelu:
        h = gelu_detachable(h, detach)
    return bcos_linear(h, sd[p + "linear2.linear.weight"], sd.get(p + "linear2.linear.bias"), b=b, detach=detach)


def simple_vit_logits(sd, x6, patch=16, heads=3, b=2, detach=False, gap_reorder=True, gelu=True,
                      mean=IMAGENET_MEAN_ADDINVERSE, std=IMAGENET_STD_ADDINVERSE, logit_bias=-math.log(1000 - 1),
                      logit_temperature=None, prefix="model."):
    """bcosify_vit.BcosifyNetwork.forward (bcosify_vit.py:80-83) around SimpleViT.forward (vit.py:322-339)."""
    xn = normalize6(x6, mean, std)
    B, C, H, W = xn.shape
    gh, gw = H // patch, W // patch
    # Rearrange 'b c (h p1) (w p2) -> b h w (p1 p2 c)'  (vit.py:290-294)
    tok = xn.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 3, 5, 1).reshape(B, gh, gw, patch * patch * C)
    x = bcos_linear(tok, sd[prefix + "to_patch_embedding.linear.linear.weight"],
                    sd.get(prefix + "to_patch_embedding.linear.linear.bias"), b=b, detach=detach)
    dim = x.shape[-1]
    x = x.reshape(B, gh * gw, dim) + posemb_sincos_2d(gh, gw, dim, dtype=x.dtype)
    depth = 0
    while (prefix + f"transformer.encoder_{depth}.attn.to_qkv.weight") in sd:
        depth += 1
    for i in range(depth):
        p = prefix + f"transformer.encoder_{i}."
        x = vit_attention(sd, p + "attn.", x, heads, b, detach) + x
        x = vit_feedforward(sd, p + "ff.net.", x, b, detach, gelu) + x

    def head(t):
        t = layer_norm_detachable(t, (dim,), sd[prefix + "linear_head.norm.weight"], sd.get(prefix + "linear_head.norm.bias"),
                                  1e-5, detach)
        return bcos_linear(t, sd[prefix + "linear_head.linear.linear.weight"], sd.get(prefix + "linear_head.linear.linear.bias"),
                           b=b, detach=detach)

    out = head(x).mean(dim=1) if gap_reorder else head(x.mean(dim=1))
    return logit_layer(out, logit_temperature, logit_bias)


# ----------------------------------------------------------------------------------------------
# B-cosified CLIP RN50 image encoder (CLIP/clip/model.py:94-154 converted by bcosify.py with clip_kd;
# factory bcos/experiments/ImageNet/clip_bcosification/model.py:8-25; head bcos/modules/bcosattnpool.py:22-59)
# ----------------------------------------------------------------------------------------------
CLIP_RN50_LAYERS = [3, 4, 6, 3]


def bcos_attention_pool(sd, p, x, num_heads, detach=False):
    """BcosAttentionPool2d.forward, pooled mode (bcosattnpool.py:33-59): no positional embedding, no biases, q and k
    detached in explanation mode, output projection = c_proj.weight used as a plain linear."""
    t = x.flatten(start_dim=2).permute(2, 0, 1)                       # (HW) N C
    t = torch.cat([t.mean(dim=0, keepdim=True), t], dim=0)            # (HW+1) N C
    q, k = t[:1], t
    if detach:
        q, k = q.detach(), k.detach()
    out, _ = F.multi_head_attention_forward(
        query=q, key=k, value=t, embed_dim_to_check=t.shape[-1], num_heads=num_heads,
        q_proj_weight=sd[p + "q_proj.weight"], k_proj_weight=sd[p + "k_proj.weight"], v_proj_weight=sd[p + "v_proj.weight"],
        in_proj_weight=None, in_proj_bias=None, bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0,
        out_proj_weight=sd[p + "c_proj.linear.weight"], out_proj_bias=None, use_separate_proj_weight=True,
        training=False, need_weights=False)
    return out.squeeze(0)


def clip_rn50_embed(sd, x6, b=2, detach=False, mean=CLIP_MEAN_ADDINVERSE, std=CLIP_STD_ADDINVERSE, num_heads=32,
                    layers=CLIP_RN50_LAYERS, prefix="model.", attn_unpool=False, gate_log=None):
    """bcosify.BcosifyNetwork.forward around ModifiedResNet.forward (CLIP/clip/model.py:139-154), B-cosified.
    `attn_unpool`: the un-pooled head (returns (HW) x N x D').  `gate_log`, if given, receives the pre-activation of every
    ReLU in execution order."""
    def relu(t):
        if gate_log is not None:
            gate_log.append(t.detach())
        return F.relu(t)

    return _clip_rn50_embed(sd, x6, b, detach, mean, std, num_heads, layers, prefix, attn_unpool, relu)


def _clip_rn50_embed(sd, x6, b, detach, mean, std, num_heads, layers, prefix, attn_unpool, relu):
    F_relu = relu                  # (only the ReLU is observed; everything else is the reference's statement order)
    x = normalize6(x6, mean, std)
    for i, (stride, pad) in zip((1, 2, 3), ((2, 1), (1, 1), (1, 1))):
        x = F_relu(_bnu(sd, f"{prefix}bn{i}", _bconv(sd, f"{prefix}conv{i}", x, stride, pad, b, detach)))
    x = F.avg_pool2d(x, 2)
    for li, nblocks in enumerate(layers, start=1):
        for bi in range(nblocks):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            out = F_relu(_bnu(sd, p + "bn1", _bconv(sd, p + "conv1", x, 1, 0, b, detach)))
            out = F_relu(_bnu(sd, p + "bn2", _bconv(sd, p + "conv2", out, 1, 1, b, detach)))
            if stride > 1:
                out = F.avg_pool2d(out, stride)
            out = _bnu(sd, p + "bn3", _bconv(sd, p + "conv3", out, 1, 0, b, detach))
            identity = x
            if (p + "downsample.1.linear.weight") in sd:     # BcosSequential renumbers (AvgPool, conv, bn) -> 0, 1, 2
                identity = F.avg_pool2d(x, stride) if stride > 1 else x
                identity = _bnu(sd, p + "downsample.2", _bconv(sd, p + "downsample.1", identity, 1, 0, b, detach))
            x = F_relu(out + identity)
    if attn_unpool:
        return bcos_attention_unpool(sd, prefix + "attnpool.", x, b=b, detach=detach)
    return bcos_attention_pool(sd, prefix + "attnpool.", x, num_heads, detach)


def bcos_attention_unpool(sd, p, x, b=2, detach=False):
    """BcosAttentionPool2d.forward, `attn_unpool` branch (bcosattnpool.py:23-32): no pooling -- every location is
    projected by v_proj (plain nn.Linear WITH its bias) and c_proj (B-cosified by the converter, bcosify.py), then
    L2-normalised over the feature dim (norm detached in explanation mode).  Returns (HW) x N x D'."""
    t = x.flatten(start_dim=2).permute(2, 0, 1)                       # (HW) N C
    t = F.linear(t, sd[p + "v_proj.weight"], sd.get(p + "v_proj.bias"))
    t = bcos_linear(t, sd[p + "c_proj.linear.weight"], sd.get(p + "c_proj.linear.bias"), b=b, detach=detach)
    norm = t.norm(dim=-1, keepdim=True)
    return t / (norm.detach() if detach else norm)


def zeroshot_logits(features, text_weights, attn_unpool=False, cos_power=1):
    """clip_evaluate (bcos/training/trainer.py:112-123): L2-normalise the image features, 100 * f @ W_text; for the
    `attn_unpool` head features are (HW) x N x D: logits * |logits|^(cos_power-1), summed over the locations."""
    f = features / features.norm(dim=-1, keepdim=True)
    logits = 100.0 * f @ text_weights
    if attn_unpool:
        logits = logits * (logits.abs().detach() ** (cos_power - 1))
        logits = logits.sum(0)
    return logits


def zeroshot_attribution(forward_fn, x6, zeroshot_weight, attn_unpool=False, pool_cosine=1, norm_max_cosine=False):
    """compute_attributions of interpretability/analyses/text_localisation.py:68-104, tensor part, image by image:
    forward in explanation mode, `img_features = outa / outa.norm(dim=-1)` (NOT detached), `logits = img_features @ W`,
    the attn_unpool pooling variants (:80-99), `logits.max(1).values.backward(inputs=[img])`.
    forward_fn(x, detach=True) -> head output.  Returns (gradients [N, 6, H, W], explained logit values [N])."""
    grads, vals = [], []
    for i in range(x6.shape[0]):
        imga = x6[i:i + 1].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            outa = forward_fn(imga, detach=True)
            img_features = outa / outa.norm(dim=-1, keepdim=True)
            logits = img_features @ zeroshot_weight
            if attn_unpool:
                logits = logits.reshape(-1, 1)
                if pool_cosine == 0:
                    num_features = logits.shape[0]
                    logits = logits.reshape(-1, num_features)
                    mask = torch.zeros_like(logits)
                    mask[torch.arange(logits.shape[0]), logits.argmax(dim=1)] = 1.0
                    logits = (logits * mask.detach()).reshape(1, num_features)
                if norm_max_cosine:
                    logits = logits / logits.abs().detach().max(dim=0, keepdim=True)[0]
                if pool_cosine > 1:
                    logits = logits * torch.pow(logits, pool_cosine - 1).abs().detach()
                logits = logits.mean(dim=0)
            if logits.dim() == 1:
                logits = logits.unsqueeze(0)
            val = logits.max(1).values
            (g,) = torch.autograd.grad(val.sum(), imga)
        grads.append(g.detach())
        vals.append(val.detach().view(-1)[0])
    return torch.cat(grads), torch.stack(vals)


# ----------------------------------------------------------------------------------------------
# Grid pointing game: interpretability/analyses/localisation.py
# ----------------------------------------------------------------------------------------------
def make_multi_image(imgs):
    """LocalisationAnalyser.make_multi_image, tensor part (localisation.py:434-446): [g*g, C, h, w] -> [1, C, g*h, g*w];
    image i = a*g + b lands in grid row b, column a."""
    g = int(round(imgs.shape[0] ** 0.5))
    return imgs.view(-1, g, g, *imgs.shape[-3:]).permute(0, 3, 2, 4, 1, 5).reshape(-1, imgs.shape[1], imgs.shape[2] * g,
                                                                                   imgs.shape[3] * g)


def attribute_selection_maps(forward_fn, img, targets):
    """explainer.attribute_selection(img, tgts).sum(1, keepdim=True) for IxG / Ours (explanation_methods/utils.py:82-99,
    explainers/captum.py:29-32, bcos/common.py:319-344): one full forward + backward per target.  -> [T, 1, H, W]."""
    maps = []
    for t in targets:
        x = img.detach().clone().requires_grad_(True)
        logits = forward_fn(x, True)
        (g,) = torch.autograd.grad(logits[:, int(t)].sum(), x)
        maps.append((x.detach() * g).sum(1, keepdim=True))
    return torch.cat(maps, 0)


def localisation_fractions(attributions, single_shape, smooth=0, neg=False):
    """localisation.py:313-321 and 387-401: smooth, (negate), keep positive attributions, per-cell means, shares.
    attributions [T,1,H,W] -> contribs [T, cells] (reference cell order), metric [T] = contribs[i, i]."""
    if smooth:
        attributions = F.avg_pool2d(attributions, smooth, stride=1, padding=(smooth - 1) // 2)
    if neg:
        attributions = -attributions
    attributions = attributions.clamp(min=0)
    contribs = F.avg_pool2d(attributions, single_shape, stride=single_shape).permute(0, 1, 3, 2).reshape(attributions.shape[0], -1)
    total = contribs.sum(1, keepdim=True)
    contribs = torch.where(total * contribs > 0, contribs / total, torch.zeros_like(contribs))
    metric = torch.stack([contribs[i, i] for i in range(contribs.shape[0])])
    return contribs, metric
