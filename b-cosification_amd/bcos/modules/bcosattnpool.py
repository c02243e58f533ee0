"""BcosAttentionPool2d: CLIP's attention-pool head with q and k detached in explanation mode
(reference bcos/modules/bcosattnpool.py:10-77).

Reference quirks reproduced on purpose: no positional embedding is added and no biases are passed (:33-58); in the
pooled mode the output projection uses `c_proj.weight` directly, i.e. the B-cos transform of the (B-cosified)
`c_proj` is bypassed (:53); `from_standard_module` copies the CLIP weights only for `attn_unpool` (:74-76).

MI355X path: the three projections and the output projection are plain fp32-MFMA GEMMs on the tapconv kernel, the
softmax attention is the LDS-resident attention kernel (32 heads x 64, 50 tokens); the explanation-mode gradient flows
through v only (bcos_attention_bwd_v).
"""
import torch
import torch.nn as nn

from . import _hipfn
from .common import DetachableModule

__all__ = ["BcosAttentionPool2d"]


class BcosAttentionPool2d(DetachableModule):
    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int = None, attn_unpool: bool = False):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        if not attn_unpool:
            self.k_proj = nn.Linear(embed_dim, embed_dim)
            self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads
        self.attn_unpool = attn_unpool
        self._caches = {k: _hipfn.WeightCache() for k in ("q", "k", "v", "c")}

    def _plain(self, key, x, lin):
        w = lin.weight          # nn.Linear.weight or the BcosifyLinear.weight property
        return _hipfn.plain_linear(x, w, None, self._caches[key], w)

    def forward(self, x):
        from bcos.models.vit import _AttentionCoreFn
        n, c, h, w = x.shape
        tokens = x.flatten(start_dim=2).permute(0, 2, 1)                      # [N, HW, C]
        if self.attn_unpool:
            t = tokens.permute(1, 0, 2).contiguous()                          # (HW) N C
            v = _hipfn.plain_linear(t, self.v_proj.weight, self.v_proj.bias, self._caches["v"], self.v_proj.weight)
            y = self.c_proj(v)                                                # (HW) N D'  (c_proj: B-cos if converted)
            norm = y.norm(dim=-1, keepdim=True)
            return y / (norm.detach() if self.detach else norm)
        tokens = torch.cat([tokens.mean(dim=1, keepdim=True), tokens], dim=1).contiguous()   # mean token first
        q = self._plain("q", tokens, self.q_proj)
        k = self._plain("k", tokens, self.k_proj)
        v = self._plain("v", tokens, self.v_proj)
        qkv = torch.cat([q, k, v], dim=-1)
        head_dim = c // self.num_heads
        out = _AttentionCoreFn.apply(qkv, self.num_heads, head_dim ** -0.5, self.detach)
        return self._plain("c", out[:, 0, :].contiguous(), self.c_proj)

    @classmethod
    def from_standard_module(cls, model, module, model_config):
        spacial_dim = model.input_resolution // 32
        embed_dim = model.conv1.out_channels * 64
        attn_unpool = model_config.get("attn_unpool", False)
        new = cls(spacial_dim, embed_dim, module.num_heads, model.output_dim, attn_unpool)
        if model_config.get("weights", None) is not None:
            for name, param in module.named_parameters():
                if attn_unpool and ("k_proj" not in name) and ("q_proj" not in name):
                    owner = new
                    *path, leaf = name.split(".")
                    for p in path:
                        owner = getattr(owner, p)
                    getattr(owner, leaf).data = param.data
        return new
