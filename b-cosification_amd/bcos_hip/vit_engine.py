"""Fused forward / explanation engine for B-cosified SimpleViT models (BASELINE.json configs[2]).

Per encoder block the launch plan is
    LN -> to_qkv GEMM -> attention -> to_out B-cos GEMM (+ residual in the epilogue)
    LN -> linear1 B-cos GEMM (+ GELU gate in the epilogue) -> linear2 B-cos GEMM (+ residual in the epilogue)
i.e. 3 fused B-cos launches, 1 plain GEMM, 2 LayerNorm and 1 attention kernel (reference: ~45 ATen launches,
bcos/models/vit.py:143-228 + bcosifylinear.py:61-94).  The patch embedding is run as a stride-16 16x16 B-cos
"convolution" over the NHWC input (identical to Rearrange + BcosifyLinear, vit.py:290-296), the classifier with
`gap_reorder` as a per-token B-cos GEMM followed by the mean + logit-bias kernel.

Explanation pass (explanation mode of bcos/common.py:163-181): LayerNorm variance, GELU gate, softmax matrix and the
B-cos scales are constants, so the backward is: per B-cos layer one GEMM with the stored scale applied in the
producer's epilogue, LayerNorm input-gradients (which also add the residual gradient and pre-multiply by the next
scale), and the attention v-gradient kernel.  The last GEMM yields the patch-major input gradient which one kernel turns
into W(x) [N,6,H,W] and the contribution map.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .lib import BCOS_EPI_FORCE_POW, BCOS_LINEAR_EPS, BcosHipError


class _Lin:
    """Kernel-side view of one BcosifyLinear / BcosLinear."""

    def __init__(self, mod):
        from bcos.modules.bcoslinear import BcosLinear
        if not isinstance(mod, BcosLinear) or mod.max_out != 1:
            raise BcosHipError(f"vit engine: expected a B-cos linear with max_out == 1, got {type(mod).__name__}")
        self.module = mod
        self.refresh()

    def refresh(self):
        w, bias = self.module._effective_weight_and_bias()
        w = w.detach()
        self.b = self.module._b_value()
        self.force_pow = bool(self.module._scaling()[1]) if hasattr(self.module, "_scaling") else False
        self.w = ops.mark_static(w.contiguous())       # [Cout, Cin]
        self.wt = ops.mark_static(w.t().contiguous())  # [Cin, Cout] for the input gradient
        self.bias = bias.detach().contiguous() if bias is not None else None
        self.cin, self.cout = w.shape[1], w.shape[0]

    def fold(self, ln):
        """The LayerNorm in front of this layer folded into its weights (`_fold_ln`): fwd(..., ln=stats) / dgrad_ln read the
        LayerNorm's INPUT."""
        wc, c = _fold_ln(self.w, ln)
        self.w_ln = ops.mark_static(wc)
        self.wt_ln = ops.mark_static(wc.t().contiguous())
        self.bias_ln = c if self.bias is None else (self.bias + c if c is not None else self.bias)

    def fwd(self, x2d, *, addend=None, act=0, want_scale=False, track=False, ln=None):
        """`track`: emit the row maxima of y (its reader is another contraction).  `ln` = (rstd, |LN(x)|^2) of ops.layernorm_stats:
        x2d is the input of the folded LayerNorm"""
        rows = x2d.shape[0]
        g = dict(N=1, H=1, W=rows, C=self.cin, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1, TH=1,
                 TW=1, OH=1, OW=rows, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=self.cout)
        y = torch.empty((rows, self.cout), device=x2d.device, dtype=torch.float32)
        t = torch.empty_like(y) if want_scale else None
        bcos = self.b != 1.0
        ops.tapconv(x2d, self.w if ln is None else self.w_ln, g, out=y, scale_out=t, bias=self.bias if ln is None else self.bias_ln,
                    addend=addend, bcos_mode=BCOS_LINEAR_EPS if bcos else 0, b=self.b, relu=act,
                    flags=BCOS_EPI_FORCE_POW if self.force_pow else 0, track_absmax=track and _F16X2,
                    row_scale=None if ln is None else ln[0], a_sumsq=None if ln is None or not bcos else ln[1])
        return y, t

    def dgrad(self, glin, *, mul=None, track=False):
        return ops.matmul_nt(glin, self.wt, mul=mul, track_absmax=track and _F16X2)

    def dgrad_ln(self, glin, rstd, *, addend=None, mul=None):
        """Input gradient through this layer AND the folded LayerNorm (variance held constant): (g * mul, g) with
        g = rstd (W'^T glin) + addend -- the first feeds the next input-gradient launch, the second is the residual-stream gradient."""
        return _dgrad_ln(glin, self.wt_ln, rstd, addend, mul)


def _fold_ln(w, ln):
    """W [Cout, D] behind a LayerNorm with affine (gamma, beta):  W LN(x) = rstd (W' x) + W beta  with
    W'[j, k] = gamma[k] W[j, k] - mean_k'(gamma[k'] W[j, k'])  (the mean subtraction of the LayerNorm moved into the weight rows;
    include/bcos_hip.h: bcos_epilogue.row_scale).  Returns (W' fp32, W beta or None); built in fp64, once per refresh."""
    w64 = w.double()
    wg = w64 * ln.w.double() if ln.w is not None else w64
    wc = (wg - wg.mean(dim=1, keepdim=True)).float().contiguous()
    c = (w64 @ ln.bias.double()).float().contiguous() if ln.bias is not None else None
    return wc, c


def _dgrad_ln(glin, wt_ln, rstd, addend, mul):
    g = torch.empty((glin.shape[0], wt_ln.shape[0]), device=glin.device, dtype=torch.float32)
    if mul is None:
        ops.matmul_nt(glin, wt_ln, out=g, addend=addend, row_scale=rstd, track_absmax=False)
        return g, g
    gm = torch.empty_like(g)
    ops.matmul_nt(glin, wt_ln, out=gm, out2=g, addend=addend, mul=mul, row_scale=rstd, track_absmax=_F16X2, track_absmax2=False)
    return gm, g


class _LN:
    def __init__(self, mod):
        from bcos.modules.norms.centered_norms import DetachableLayerNorm
        if not isinstance(mod, DetachableLayerNorm):
            raise BcosHipError(f"vit engine: expected DetachableLayerNorm, got {type(mod).__name__}")
        self.module = mod
        self.refresh()

    def refresh(self):
        m = self.module
        self.w = m.weight.detach().contiguous() if m.weight is not None else None
        self.bias = m.bias.detach().contiguous() if m.bias is not None else None
        self.eps = m.eps

    def fwd(self, x2d, keep):
        return ops.layernorm_fwd(x2d, self.w, self.bias, self.eps, want_rstd=keep, want_absmax=_F16X2)      # y feeds a contraction

    def stats(self, x2d, want_norm):
        """(rstd, |LN(x)|^2): what the contraction with the folded weights needs to read x2d itself (x2d gets its operand maxima)"""
        return ops.layernorm_stats(x2d, self.w, self.bias, self.eps, want_zsumsq=want_norm, want_absmax=_F16X2)

    def bwd(self, gy, rstd, *, addend=None, mul2=None, want_out=True, want_out2=False):
        return ops.layernorm_bwd_detached(gy, self.w, rstd, addend=addend, mul2=mul2, want_out=want_out, want_out2=want_out2,
                                          want_absmax2=_F16X2 and want_out2)                                # out2 = g_lin feeds one


import contextlib
import os


_F16X2 = os.environ.get("BCOS_VIT_F16X2", "1") != "0"
_VIT_SUBBATCH_STREAMS = int(os.environ.get("BCOS_VIT_SUBBATCH_STREAMS", os.environ.get("BCOS_SUBBATCH_STREAMS", "3")))
# Round 4: every LayerNorm of the plan is folded into the contraction that reads it -- forward: row statistics (one read of x, two
# floats per row out) + the GEMM over x with mean-centred, gamma-scaled weight rows, rstd as a row factor of the accumulator and
# W beta as bias; explanation pass: the detached-variance input gradient IS the input-gradient GEMM over the same weights, with
# the residual add and the next layer's stored scale in its epilogue.  No LayerNorm output or gradient tensor is written or
# read.  BCOS_VIT_LN_FUSED=0: the round-3 plan (layernorm_fwd / layernorm_bwd_detached kernels between the GEMMs).
_LN_FUSED = os.environ.get("BCOS_VIT_LN_FUSED", "1") != "0"
_HEAD_RANK1 = os.environ.get("BCOS_VIT_HEAD_RANK1", "1") != "0"      # 0: the round-4 head gradient (one-hot tensor + contraction), for A/B


def _absmax_policy():
    """Round 3: every contraction of the plan runs the split-f16 loop with LDS-DMA staging (3 matrix products instead of the 6
    of the bf16 split).  Its A operands need per-row maxima: GEMM epilogues emit them, LayerNorm / attention outputs get
    them from one extra pass (`_mx`).  BCOS_VIT_F16X2=0 restores the round-2 plan (no maxima, bf16x3 everywhere)."""
    return contextlib.nullcontext() if _F16X2 else ops.no_absmax()


def _mx(t, src=None):
    """operand maxima of a tensor a kernel other than a contraction produced: taken over from `src` (the tensor the kernel
    emitted them for, of which `t` is a reshaped view) or computed in one extra pass"""
    if not _F16X2:
        return t
    am = ops.absmax_of(src) if src is not None else None
    if am is not None and ops.absmax_of(t) is None:
        ops._attach_absmax(t, am)
    return ops.ensure_absmax(t)


class ViTEngine:
    """Launch plan for `bcosify_vit.BcosifyNetwork(SimpleViT(...))`."""

    def __init__(self, net):
        from bcos.models.vit import SimpleViT
        from bcosify_vit import MyGELU
        m = net.model
        if not isinstance(m, SimpleViT):
            raise BcosHipError(f"vit engine: {type(m).__name__} is not a SimpleViT")
        self.net = net
        self.patch = m.patch_size[0]
        if m.patch_size[0] != m.patch_size[1]:
            raise BcosHipError("vit engine: square patches only")
        self.embed_mod = m.to_patch_embedding.linear
        # conv-stem models (vitc_*: bcos/models/vit.py:342-426): [B-cos conv 3x3 -> DetachableGroupNorm2d -> MyGELU] x 4 / 6 take
        # the 224^2 image to a 14^2 map that is tokenised with patch size 1
        self.stem = []
        stem = getattr(m.to_patch_embedding, "conv_stem", None)
        if stem is not None:
            from bcos.modules.norms.centered_norms import DetachableGroupNorm2d
            from .engine import _Conv
            mods = list(stem.children())
            if len(mods) % 3:
                raise BcosHipError("vit engine: conv stem must be (conv, norm, activation) triples")
            for conv, gn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
                if not isinstance(gn, DetachableGroupNorm2d) or not isinstance(act, (MyGELU, nn.Identity)):
                    raise BcosHipError(f"vit engine: unsupported conv-stem layer ({type(gn).__name__}, {type(act).__name__})")
                self.stem.append(dict(conv=_Conv(conv, None), gn=gn, gelu=isinstance(act, MyGELU)))
        self.blocks = []
        for enc in m.transformer.children():
            act = enc.ff.net.act
            if isinstance(act, MyGELU):
                act_mode = 2
            elif isinstance(act, nn.Identity):
                act_mode = 0
            else:
                raise BcosHipError(f"vit engine: unsupported activation {type(act).__name__}")
            if enc.attn.to_qkv.bias is not None:
                raise BcosHipError("vit engine: to_qkv with bias is not supported")
            self.blocks.append(dict(ln1=_LN(enc.attn.norm), qkv=enc.attn.to_qkv, heads=enc.attn.heads,
                                    scale=enc.attn.scale, out=_Lin(enc.attn.to_out), ln2=_LN(enc.ff.net.norm),
                                    l1=_Lin(enc.ff.net.linear1), act=act_mode, l2=_Lin(enc.ff.net.linear2)))
        self.head_ln = _LN(m.linear_head.norm)
        self.head = _Lin(m.linear_head.linear)
        self.gap_reorder = bool(m.gap_reorder)
        norm = net.bcosifynormalize
        self._mean, self._std = tuple(norm.mean), tuple(norm.std)
        self._dev = {}
        ll = net.logit_layer
        self.logit_bias = ll.logit_bias if ll is not None else None
        self.logit_temperature = ll.logit_temperature if ll is not None else None
        self._absmax_arena = ops.AbsmaxArena()      # row maxima of one pass: one zero fill instead of one per tensor
        # three sub-batch streams for the token path (ResNets: two): its launches are shorter -- 394 tiles per half-batch GEMM at batch
        # 512 -- and a third stream fills more of their tails: same-node A/B at ViT-Ti batch 512, three pairs: 18.76 / 18.86 / 18.77 ms
        # against 19.23 / 19.29 / 19.14 with two (ResNet-50: no difference).  Fewer streams for batches under 3 x _SUBBATCH_MIN.
        self.subbatch_streams, self._side = _VIT_SUBBATCH_STREAMS, None
        self.refresh()

    def _fingerprint(self):
        """(storage, in-place version) of every parameter and buffer of the network the plan was compiled from."""
        m = self.net
        return tuple((t.data_ptr(), t._version) for t in list(m.parameters()) + list(m.buffers()))

    def _ensure_fresh(self):
        ops.publish_pending()          # (objects a training pass cached without publication are completed before an inference pass reads them)
        if self._fingerprint() != self._fp:
            self.refresh()
            ops.publish_cached(self.embed_w)    # (a refresh inside a sub-batch pass runs on that pass's side stream; the other one reads the result)

    def refresh(self):
        from bcos.modules.bcoslinear import BcosLinear
        self._fp = self._fingerprint()
        e = self.embed_mod
        if not isinstance(e, BcosLinear) or e.max_out != 1:
            raise BcosHipError("vit engine: patch embedding must be a B-cos linear")
        w, bias = e._effective_weight_and_bias()
        p = self.patch
        dim = w.shape[0]
        for layer in self.stem:
            layer["conv"].refresh()
            gn = layer["gn"]
            layer.update(gw=gn.weight.detach().contiguous() if gn.weight is not None else None,
                         gb=gn.bias.detach().contiguous() if gn.bias is not None else None, groups=gn.num_groups, eps=gn.eps)
        cin = self.stem[-1]["conv"].cout if self.stem else 6          # channels of the tensor that is cut into patches
        cpad = (cin + 3) & ~3 if self.stem else 8
        if w.shape[1] != p * p * cin:
            raise BcosHipError(f"vit engine: patch embedding must take {cin}-channel patches")
        self.embed_cpad = cpad
        w4 = torch.nn.functional.pad(w.detach().view(dim, p, p, cin), (0, cpad - cin))  # [dim, p, p, cpad]
        self.embed_w = ops.mark_static(w4.contiguous())                               # inference constants: pre-split images are kept
        self.embed_wt = ops.mark_static(w4.reshape(dim, p * p * cpad).t().contiguous())  # [p*p*cpad, dim]
        self.embed_bias = bias.detach().contiguous() if bias is not None else None
        self.embed_b = e._b_value()
        self.dim = dim
        for blk in self.blocks:
            for k in ("ln1", "out", "ln2", "l1", "l2"):
                blk[k].refresh()
            wq = blk["qkv"].weight.detach()
            inner = wq.shape[0] // 3
            blk["wqkv"] = ops.mark_static(wq.clone().contiguous())                   # [3*inner, dim]
            blk["wv_t"] = ops.mark_static(wq[2 * inner:].t().contiguous())           # [dim, inner]: gx = gv @ Wv
            blk["inner"] = inner
            wc, c = _fold_ln(wq, blk["ln1"])                                          # LN1 folded into to_qkv, LN2 into linear1
            blk["wqkv_ln"], blk["cqkv"] = ops.mark_static(wc), c
            blk["wv_t_ln"] = ops.mark_static(wc[2 * inner:].t().contiguous())
            blk["l1"].fold(blk["ln2"])
        self.head_ln.refresh()
        self.head.refresh()
        self.head.fold(self.head_ln)
        self._pe = {}

    def _consts(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.tensor(self._mean, dtype=torch.float32, device=device),
                              torch.tensor(self._std, dtype=torch.float32, device=device))
        return self._dev[key]

    def _posemb(self, gh, gw, device):
        key = (gh, gw, str(device))
        if key not in self._pe:
            pe = self.net.model.positional_embedding(torch.empty((1, gh, gw, self.dim), device=device))
            self._pe[key] = pe.contiguous()
            ops.publish_cached(self._pe[key])      # (made on whichever stream asked first; read by every later pass)
        return self._pe[key]

    # --------------------------------------------------------------------------------------------------------
    def _run_forward(self, x, keep):
        from .engine import _drive
        return _drive(self._run_forward_gen(x, keep))

    def _run_forward_gen(self, x, keep):
        """the forward pass as a generator (yields behind the embedding and behind every encoder block: engine._interleave)"""
        if x.dim() != 4 or x.shape[1] not in (3, 6):
            raise ValueError(f"expected [N,6,H,W] (or [N,3,H,W] to be AddInverse-encoded), got {tuple(x.shape)}")
        ops.require_device(x, "bcos_hip.vit_engine")
        self._ensure_fresh()
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()
        N, _, H, W = x.shape
        p = self.patch
        mean, std = self._consts(x.device)
        add_inverse = x.shape[1] == 3
        xn = ops.prep_input(x, mean, std, cpad=8, add_inverse=add_inverse, want_absmax=_F16X2)     # K = 2048 patch embedding
        stem_st = []
        for layer in self.stem:          # conv stem: B-cos conv, GroupNorm (variance held constant in explanation mode), GELU gate
            hw_in = (xn.shape[1], xn.shape[2])
            y, t = layer["conv"].fwd(xn, relu=False, want_scale=keep, track=False)
            yn, rstd = ops.groupnorm_fwd(y, layer["groups"], layer["gw"], layer["gb"], layer["eps"], want_rstd=keep)
            gate = None
            if layer["gelu"]:
                yn, gate = ops.gelu_gate(yn, want_gate=keep, out=yn)
            xn = _mx(yn) if layer is self.stem[-1] else ops.ensure_absmax(yn) if _F16X2 else yn
            stem_st.append(dict(t=t, rstd=rstd, gate=gate, hw_in=hw_in))
        H, W = xn.shape[1], xn.shape[2]
        gh, gw = H // p, W // p
        T = gh * gw
        # patch embedding = p x p / stride p B-cos conv with the linear layer's epsilon placement
        geom = ops.fwd_geom(N, H, W, self.embed_cpad, self.dim, p, p, p, p, 0, 0)
        tok = torch.empty((N * T, self.dim), device=x.device, dtype=torch.float32)
        t_embed = torch.empty_like(tok) if keep else None
        ops.tapconv(xn, self.embed_w, geom, out=tok, scale_out=t_embed, bias=self.embed_bias,
                    bcos_mode=BCOS_LINEAR_EPS if self.embed_b != 1.0 else 0, b=self.embed_b, track_absmax=False)
        ops.add_rows_bcast(tok, self._posemb(gh, gw, x.device))
        st = dict(x=x, add_inverse=add_inverse, N=N, T=T, t_embed=t_embed, blocks=[], stem=stem_st, tok_hw=(H, W)) if keep else None
        cur = tok
        yield
        for blk in self.blocks:
            if _LN_FUSED:
                rstd1, _ = blk["ln1"].stats(cur, False)
                qkv = ops.matmul_nt(cur, blk["wqkv_ln"], bias=blk["cqkv"], row_scale=rstd1, track_absmax=False)
            else:
                h, rstd1 = blk["ln1"].fwd(cur, keep)
                qkv = ops.matmul_nt(_mx(h), blk["wqkv"], track_absmax=False)
            a, stats = ops.attention_fwd(qkv.view(N, T, -1), blk["heads"], blk["scale"], want_stats=keep, want_absmax=_F16X2)
            x1, t_out = blk["out"].fwd(_mx(a.view(N * T, -1), a), addend=cur, want_scale=keep)
            if _LN_FUSED:
                st2 = blk["ln2"].stats(x1, True)
                rstd2 = st2[0]
                z, t1 = blk["l1"].fwd(x1, act=blk["act"], want_scale=keep, track=True, ln=st2)
            else:
                h2, rstd2 = blk["ln2"].fwd(x1, keep)
                z, t1 = blk["l1"].fwd(_mx(h2), act=blk["act"], want_scale=keep, track=True)      # z feeds linear2
            x2, t2 = blk["l2"].fwd(z, addend=x1, want_scale=keep)
            if keep:
                st["blocks"].append(dict(rstd1=rstd1, qkv=qkv, stats=stats, t_out=t_out, rstd2=rstd2, t1=t1, t2=t2))
            cur = x2
            yield
        if self.gap_reorder and _LN_FUSED:
            st_h = self.head_ln.stats(cur, True)
            rstd_h = st_h[0]
            f, t_head = self.head.fwd(cur, want_scale=keep, ln=st_h)
            logits = ops.global_avgpool_logits(f.view(N, T, 1, -1), self.logit_temperature, self.logit_bias)
        elif self.gap_reorder:
            hN, rstd_h = self.head_ln.fwd(cur, keep)
            f, t_head = self.head.fwd(_mx(hN), want_scale=keep)
            logits = ops.global_avgpool_logits(f.view(N, T, 1, -1), self.logit_temperature, self.logit_bias)
        else:
            pooled = ops.global_avgpool_logits(cur.view(N, T, 1, -1), None, None)
            hN, rstd_h = self.head_ln.fwd(pooled, keep)
            f, t_head = self.head.fwd(hN, want_scale=keep)
            logits = f
            if self.logit_temperature is not None:
                logits = logits / self.logit_temperature
            if self.logit_bias is not None:
                logits = logits + self.logit_bias
        if keep:
            st.update(rstd_h=rstd_h, t_head=t_head)
        return logits, st

    def _sub_batches(self, x, make_gen):
        """Run the pass generator make_gen(lo, hi) for contiguous sub-batches on side streams, their launches issued interleaved
        (bcos_hip/engine.py: _SUBBATCH_STREAMS, _interleave: images are independent, the sub-batches fill each other's launch tails)
        or once on the caller's stream."""
        from .engine import _SUBBATCH_MIN, _drive, _interleave
        N = x.shape[0]
        S = min(int(self.subbatch_streams), N // _SUBBATCH_MIN)
        if S <= 1 or not x.is_cuda or torch.cuda.is_current_stream_capturing():
            with _absmax_policy(), ops.absmax_arena(self._absmax_arena, x.device):
                return [_drive(make_gen(0, N))]
        # everything the passes cache lazily (refreshed plans, constants, the positional-embedding table) is brought up to date here,
        # on the caller's stream, which every side stream then waits for; streams are per device
        self._ensure_fresh()
        self._consts(x.device)
        if not self.stem and x.dim() == 4:
            self._posemb(x.shape[2] // self.patch, x.shape[3] // self.patch, x.device)
        key = str(x.device)
        if self._side is None:
            self._side = {}
        if key not in self._side or len(self._side[key][0]) < S:
            self._side[key] = ([torch.cuda.Stream(device=x.device) for _ in range(S)], [ops.AbsmaxArena() for _ in range(S)])
        streams, arenas = self._side[key]
        cur = torch.cuda.current_stream()
        for i in range(S):
            streams[i].wait_stream(cur)
        with _absmax_policy():
            outs = _interleave([make_gen((N * i) // S, (N * (i + 1)) // S) for i in range(S)], streams, arenas, x.device)
        for st in streams[:S]:
            cur.wait_stream(st)
        for o in outs:
            for t in (o.values() if isinstance(o, dict) else [o]):
                if torch.is_tensor(t):
                    t.record_stream(cur)
        return outs

    @torch.no_grad()
    def forward(self, x):
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()

        def one(lo, hi):
            logits, _ = yield from self._run_forward_gen(x[lo:hi], keep=False)
            return logits
        outs = self._sub_batches(x, one)
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    @torch.no_grad()
    def explain(self, x, targets: Optional[torch.Tensor] = None, want_weights: bool = True) -> Dict[str, torch.Tensor]:
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()
        targets = ops.check_targets(targets, self.head.cout)      # IndexError like the reference's out[0, idx]; negative indices wrap
        tg = None if targets is None else targets.to(device=x.device, dtype=torch.int64).contiguous()

        # the two image-sized results of the sub-batches land in ONE pair of tensors (engine.ResNetEngine._explain_subbatches): no
        # concatenation pass over [N, 6, H, W] behind the step (0.3 ms of a 17.4 ms ViT-Ti step at batch 512)
        N, _, H, W = x.shape
        wts = torch.empty((N, 6, H, W), device=x.device, dtype=torch.float32) if want_weights else None
        contrib = torch.empty((N, H, W), device=x.device, dtype=torch.float32)

        def one(lo, hi):
            return self._explain_gen(x[lo:hi], None if tg is None else tg[lo:hi], want_weights,
                                     outs=(wts[lo:hi] if want_weights else None, contrib[lo:hi]))
        outs = self._sub_batches(x, one)
        res = {k: (outs[0][k] if len(outs) == 1 else torch.cat([o[k] for o in outs])) for k in ("logits", "prediction", "explained_class_idx")}
        res.update(dynamic_linear_weights=wts, contribution_map=contrib)
        return res

    def _explain(self, x, targets, want_weights):
        from .engine import _drive
        return _drive(self._explain_gen(x, targets, want_weights))

    def _explain_gen(self, x, targets, want_weights, outs=None):
        logits, st = yield from self._run_forward_gen(x, keep=True)
        N, T = st["N"], st["T"]
        pred, _ = ops.argmax_rows(logits)
        cls = pred if targets is None else targets.to(device=logits.device, dtype=torch.int64).contiguous()
        nb = len(self.blocks)
        t_last = st["blocks"][-1]["t2"] if nb else st["t_embed"]
        if self.gap_reorder and _LN_FUSED and _HEAD_RANK1 and self.dim % 4 == 0:
            # d mean-logit[cls] / d (head input) is rank one per image: one class column of the stored scale times one row of the folded
            # weights -- a streaming launch instead of the [N T, K] one-hot tensor and a K-long contraction over it
            g_lin, g_x = ops.head_rank1_grad(cls, st["t_head"].view(N, T, -1), self.head.w_ln, self.logit_temperature, row_scale=st["rstd_h"],
                                             mul=t_last, want_out2=True, want_absmax=_F16X2)
        elif self.gap_reorder:
            g_head = ops.head_onehot_grad(cls, st["t_head"].view(N, T, 1, -1), self.logit_temperature)     # [N,T,1,K]
            if _LN_FUSED:
                g_lin, g_x = self.head.dgrad_ln(_mx(g_head.view(N * T, -1)), st["rstd_h"], mul=t_last)
            else:
                g_hN = self.head.dgrad(_mx(g_head.view(N * T, -1)))
                g_x, g_lin = self.head_ln.bwd(g_hN, st["rstd_h"], mul2=t_last, want_out=True, want_out2=True)
        else:
            g_head = ops.head_onehot_grad(cls, st["t_head"].view(N, 1, 1, -1), self.logit_temperature)     # [N,1,1,K]
            g_hN = self.head.dgrad(g_head.view(N, -1))
            g_pool, _ = self.head_ln.bwd(g_hN, st["rstd_h"])
            g_x = (g_pool / T).repeat_interleave(T, dim=0).contiguous()          # gradient of the token mean
            g_lin = ops.mul(g_x, t_last)
        # invariant at the top of each block iteration: g_x = d logit / d (block output), g_lin = g_x * t2 of the block
        for bi in range(nb - 1, -1, -1):
            blk, rec = self.blocks[bi], st["blocks"][bi]
            g_z = blk["l2"].dgrad(_mx(g_lin), mul=rec["t1"], track=True)        # = g_lin of linear1 (GELU gate inside t1)
            if _LN_FUSED:
                g_lin_out, g_x1 = blk["l1"].dgrad_ln(g_z, rec["rstd2"], addend=g_x, mul=rec["t_out"])
            else:
                g_h2 = blk["l1"].dgrad(g_z)
                g_x1, g_lin_out = blk["ln2"].bwd(g_h2, rec["rstd2"], addend=g_x, mul2=rec["t_out"], want_out2=True)
            g_a = blk["out"].dgrad(_mx(g_lin_out))
            g_v = ops.attention_bwd_v(rec["qkv"].view(N, T, -1), rec["stats"], g_a.view(N, T, -1), blk["heads"], blk["scale"],
                                      want_absmax=_F16X2)
            t_prev = st["blocks"][bi - 1]["t2"] if bi > 0 else st["t_embed"]
            if _LN_FUSED:
                g_lin, g_x = _dgrad_ln(_mx(g_v.view(N * T, -1), g_v), blk["wv_t_ln"], rec["rstd1"], g_x1, t_prev)
            else:
                g_h = ops.matmul_nt(_mx(g_v.view(N * T, -1), g_v), blk["wv_t"], track_absmax=False)
                g_x, g_lin = blk["ln1"].bwd(g_h, rec["rstd1"], addend=g_x1, mul2=t_prev, want_out=bi > 0, want_out2=True)
            st["blocks"][bi] = None
            yield
        gp = ops.matmul_nt(_mx(g_lin), self.embed_wt, track_absmax=False)      # [N*T, p*p*cpad] patch-major input gradient
        _, std = self._consts(x.device)
        if self.stem:
            # conv-stem models: patch size 1, so gp IS the NHWC gradient w.r.t. the stem's output; back through
            # (GELU gate, GroupNorm with constant variance, B-cos scale, convolution) of every stem layer
            if self.patch != 1:
                raise BcosHipError("vit engine: conv stems are tokenised with patch size 1")
            Hs, Ws = st["tok_hw"]
            g = gp.view(N, Hs, Ws, -1)
            for layer, rec in zip(reversed(self.stem), reversed(st["stem"])):
                cout = layer["conv"].cout
                g = g[..., :cout].contiguous() if g.shape[-1] != cout else g
                if rec["gate"] is not None:
                    g = ops.mul(g, rec["gate"])
                g = ops.groupnorm_bwd_detached(g, layer["groups"], layer["gw"], rec["rstd"])
                g = ops.mul(g, rec["t"]) if rec["t"] is not None else g
                hi, wi = rec["hw_in"]
                g = layer["conv"].dgrad.run(ops.ensure_absmax(g) if _F16X2 else g, hi, wi)
            gxn = g if g.shape[-1] == 8 else torch.nn.functional.pad(g, (0, 8 - g.shape[-1]))
            wts, contrib = ops.finalize_explanation(gxn.contiguous(), st["x"], std, add_inverse=st["add_inverse"],
                                                    want_weights=want_weights, want_contrib=True,
                                                    weights_out=outs[0] if outs else None, contrib_out=outs[1] if outs else None)
        else:
            wts, contrib = ops.finalize_explanation_patches(gp, st["x"], std, self.patch, add_inverse=st["add_inverse"],
                                                            want_weights=want_weights, want_contrib=True,
                                                            weights_out=outs[0] if outs else None, contrib_out=outs[1] if outs else None)
        return dict(logits=logits, prediction=pred, explained_class_idx=cls, dynamic_linear_weights=wts,
                    contribution_map=contrib)


def attach(net) -> ViTEngine:
    eng = ViTEngine(net)
    object.__setattr__(net, "_bcos_engine", eng)
    return eng
