"""Training step of a B-cosified SimpleViT as ONE launch plan.

The reference trains its ViTs with the same modules it explains them with (bcos/training/trainer.py:666-784 in train() mode over
bcos/models/vit.py:143-228, bcosifylinear.py:61-94, centered_norms.py:187-245, bcosify_vit.py:27-32; nothing detached).  With a
ViTEngine attached, `net.train(); net(x)` is ONE autograd node whose forward / backward walk the engine's block list and issue the HIP
launches directly on [rows, dim] token matrices -- the per-layer path's arithmetic launch for launch (bcos/modules/_hipfn.py:
BcosLinearFn, centered_norms.py: _LayerNormFn, bcos/models/vit.py: _AttentionCoreFn, bcosify_vit.py: _GeluFn), without one autograd
node, one module call and one set of reshapes per layer, with the operand maxima of every contraction operand coming out of the
launch that writes it (so forward and input-gradient contractions run the 3-product split-f16 loop), and one zero fill per pass for
all of them.

Per encoder block (vit.py:143-158):
  forward    LayerNorm (rstd kept) -> to_qkv GEMM -> attention (softmax statistics kept) -> to_out B-cos linear with the scale NOT
             detached (y, s, |x| kept) -> + x;  LayerNorm -> linear1 B-cos -> GELU -> linear2 B-cos -> + x
  backward   per B-cos linear: derivative of the dynamic scale (bcos_train_scale_bwd), the |x| term (bcos_patch_norm_bwd, added by the
             input-gradient launch's epilogue), weight gradient (bcos_conv2d_wgrad, 1 x 1), input gradient; the full GELU, attention
             and LayerNorm gradients (bcos_gelu_bwd, bcos_attention_bwd, bcos_layernorm_bwd + two column sums for gamma / beta).

Scope: plain SimpleViT (patch embedding by rearrangement, no convolution stem), B-cos linears with max_out == 1, a fixed exponent and
plain (not unit-norm) weights, MyGELU or no activation.  Anything else, and any module switched to explanation mode (`detach`) while
the network is in train(), keeps the per-layer path.

Like the inference plans (and the reference's own training loop) a plan is single-threaded: its arenas of operand maxima, the zero-fill
arena of the weight gradients and the second stream belong to ONE pass at a time.  Several forward passes before their backward passes
(gradient accumulation) are fine -- a pass's state lives in its autograd node, a stale arena slice is refused by ops.absmax_of -- but
two threads driving the same network are not.
"""
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .lib import BCOS_LINEAR_EPS, BCOS_EPI_FORCE_POW, BcosHipError


def _pad4(t: torch.Tensor) -> torch.Tensor:
    r = (-t.shape[-1]) % 4
    return F.pad(t, (0, r)) if r else t


class _LinState:
    __slots__ = ("mod", "x", "y", "scale", "norm", "w", "b", "force_pow")


class ViTTrainPlan:
    def __init__(self, eng):
        ok, why = self.supported(eng)
        if not ok:
            raise BcosHipError(f"vit train plan: {why}")
        self.eng, self.net = eng, eng.net
        self._arena_f, self._arena_b = ops.AbsmaxArena(), ops.AbsmaxArena()
        from .train_plan import ParamGradQueue
        self._pq = ParamGradQueue()           # weight gradients and column sums on a second stream
        self._wbatch, self._wbanks = None, {}  # ops.WeightPrepBatch of the linear layers; id(parameter) -> (parameter, forward bank, transposed bank)
        self._zeros = ops.ZeroArena()         # their accumulators from one zero fill per pass

    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _linears(eng):
        mods = [eng.embed_mod]
        for blk in eng.blocks:
            mods += [blk["out"].module, blk["l1"].module, blk["l2"].module]
        return mods + [eng.head.module]

    @staticmethod
    def _norms(eng):
        return [ln.module for blk in eng.blocks for ln in (blk["ln1"], blk["ln2"])] + [eng.head_ln.module]

    @staticmethod
    def supported(eng):
        from bcos.modules.bcoslinear import NormedLinear
        if eng.stem:
            return False, "convolution stems keep the per-layer path"
        for m in ViTTrainPlan._linears(eng):
            if isinstance(getattr(m, "b", None), torch.Tensor) and m.b.requires_grad:
                return False, "learnable exponent"
            if isinstance(m.linear, NormedLinear):
                return False, "native unit-norm layers"
            if int(getattr(m, "max_out", 1)) != 1:
                return False, "MaxOut layers keep the per-layer path"
        for m in ViTTrainPlan._linears(eng)[1:]:
            if m.linear.in_features % 4:
                return False, "feature widths must be multiples of 4"
        for blk in eng.blocks:
            if blk["qkv"].in_features % 4 or blk["qkv"].out_features % 4:
                return False, "feature widths must be multiples of 4"
        return True, ""

    def active(self) -> bool:
        """False while any module of the network is in explanation mode (detached scales / variances / gates / softmax): that
        combination with train() keeps the per-layer path"""
        mods = self._linears(self.eng) + self._norms(self.eng) + [enc.attn for enc in self.net.model.transformer.children()]
        mods += [enc.ff.net.act for enc in self.net.model.transformer.children()]
        return not any(bool(getattr(m, "detach", False)) for m in mods)

    def parameters(self) -> List[nn.Parameter]:
        ps, seen = [], set()
        cand = []
        for m in self._linears(self.eng):
            cand += [m.linear.weight, m.linear.bias]
        for m in self._norms(self.eng):
            cand += [m.weight, m.bias]
        for blk in self.eng.blocks:
            cand += [blk["qkv"].weight]
        for p in cand:
            if isinstance(p, nn.Parameter) and id(p) not in seen:
                seen.add(id(p))
                ps.append(p)
        return ps

    # ------------------------------------------------------------------------------------------------------------------
    def _weights(self, device):
        """every linear layer's forward and transposed weight bank, with their split images, from ONE launch at the top of the forward pass
        (ops.WeightPrepBatch; train_plan.ResNetTrainPlan._weights) -- a ViT-Ti step made ~100 banks by ~250 small launches"""
        from .train_plan import _WEIGHT_BATCH
        if not _WEIGHT_BATCH or torch.device(device).type != "cuda" or ops._l.get_contraction_mode() != "f16x2":
            self._wbanks = {}
            return
        if self._wbatch is None or self._wbatch.device != torch.device(device):
            batch, banks = ops.WeightPrepBatch(device), {}
            ws = [m.linear.weight for m in self._linears(self.eng) if m is not self.eng.embed_mod] + [blk["qkv"].weight for blk in self.eng.blocks]
            for p in ws:
                if isinstance(p, nn.Parameter) and p.dim() == 2 and p.is_contiguous() and id(p) not in banks:
                    pair = batch.add_linear(p)
                    if pair is not None:
                        banks[id(p)] = (p,) + pair
            self._wbatch, self._wbanks = batch, banks
        self._wbatch.run()

    def _bank(self, param, w, transposed: bool):
        """the step's bank of `param` (forward or transposed) when `w` -- the weight a launch is about to read -- is that parameter"""
        hit = self._wbanks.get(id(param))
        if hit is None or hit[0] is not param or w.data_ptr() != param.data_ptr() or tuple(w.shape) != tuple(param.shape):
            return None
        return hit[2] if transposed else hit[1]

    @staticmethod
    def _lin_setup(mod, st):
        w, bias = mod._effective_weight_and_bias()
        st.mod, st.w = mod, w.detach()
        st.b = float(mod._b_value())
        st.force_pow = bool(mod._scaling()[1]) if hasattr(mod, "_scaling") else False
        return bias.detach().contiguous() if bias is not None else None

    def _lin_fwd(self, mod, x2):
        """B-cos linear on [rows, Cin] with the scale differentiated: keeps x, y, s and |x| (+ eps) for the backward"""
        st = _LinState()
        bias = self._lin_setup(mod, st)
        wk = self._bank(mod.linear.weight, st.w, False)
        if wk is None:
            wk = ops.mark_static(st.w if st.w.is_contiguous() else st.w.contiguous())
        bcos = st.b != 1.0
        y, scale, norm = ops.linear_fwd(x2, wk, bias=bias, b=st.b, want_scale=bcos, want_norm=bcos,
                                        flags=BCOS_EPI_FORCE_POW if st.force_pow else 0)
        st.x, st.y, st.scale, st.norm = x2, y, scale, norm
        return y, st

    def _lin_glin(self, st, gy, want_absmax=True):
        """gradient w.r.t. the linear map's output, and d loss / d |x| / |x| per row (None for b == 1)"""
        Cout = st.y.shape[1]
        gy = gy if gy.is_contiguous() else gy.contiguous()
        if st.b == 1.0:
            return (ops.ensure_absmax(gy) if want_absmax else gy), None
        if Cout % 4 == 0:
            glin, rnorm, _ = ops.train_scale_bwd(gy, st.y, st.scale, st.norm.view(-1), BCOS_LINEAR_EPS, st.b, st.force_pow,
                                                 want_absmax=want_absmax)
            return glin, rnorm
        from bcos.modules._hipfn import _scale_bwd_cols
        glin, rnorm, _ = _scale_bwd_cols(gy, st.y, st.scale, st.norm.view(-1), BCOS_LINEAR_EPS, dict(b=st.b, force_pow=st.force_pow), False)
        return glin, rnorm

    def _lin_bwd(self, st, gy, grads, need_x=True):
        Cout, Cin = st.w.shape
        rows = st.x.shape[0]
        glin, rnorm = self._lin_glin(st, gy)
        gl4 = _pad4(glin)
        gl4 = gl4 if gl4.is_contiguous() else gl4.contiguous()
        lin = st.mod.linear
        x = st.x
        if lin.weight.requires_grad:
            # (the ordered weight gradient WRITES its result: no zeroed accumulator; the atomics kernel -- mode f32, odd sizes -- needs one)
            acc = self._zeros.take((Cout, 1, 1, Cin), x.device) if not (ops.wgrad_is_ordered() and (Cout * 1 * 1 * Cin) % 4 == 0) else None
            grads[lin.weight] = self._pq.run(lambda: ops.conv2d_wgrad(gl4.view(1, 1, rows, gl4.shape[1]), x.view(1, 1, rows, Cin), Cin, Cout,
                                                                      (1, 1), (1, 1), (0, 0), (1, 1), out=acc).view(Cout, Cin), (gl4, x, acc))
        if lin.bias is not None and lin.bias.requires_grad:
            grads[lin.bias] = self._pq.run(lambda: ops.colsum(gl4)[:Cout].contiguous(), (gl4,))
        if not need_x:
            return None
        wt = self._bank(st.mod.linear.weight, st.w, True)
        if wt is None:
            wt = ops.mark_static(_pad4(st.w.t()).contiguous())           # [Cin, Cout (+ pad)]
        if rnorm is not None:             # (the |x| term of the scale's derivative: added by the launch's epilogue)
            return ops.matmul_nt_with_row_term(gl4, wt, st.x.view(rows, Cin), rnorm)
        return ops.matmul_nt(gl4, wt, track_absmax=False)

    @staticmethod
    def _ln_fwd(ln, x2, want_absmax=True):
        m = ln.module
        w = m.weight.detach().contiguous() if m.weight is not None else None
        b = m.bias.detach().contiguous() if m.bias is not None else None
        y, rstd = ops.layernorm_fwd(x2, w, b, m.eps, want_rstd=True, want_absmax=want_absmax)
        return y, (m, x2, w, rstd)

    def _ln_bwd(self, rec, gy, grads, addend=None):
        """-> gradient w.r.t. the LayerNorm's input (+ `addend`: what reaches the same tensor around the sub-block)"""
        m, x2, w, rstd = rec
        gy = gy if gy.is_contiguous() else gy.contiguous()
        need_w = isinstance(m.weight, nn.Parameter) and m.weight.requires_grad
        need_b = isinstance(m.bias, nn.Parameter) and m.bias.requires_grad
        gx, xhat = ops.layernorm_bwd(gy, x2, w, rstd, want_xhat=need_w, addend=addend)
        D = gy.shape[1]
        if need_w:
            grads[m.weight] = self._pq.run(lambda: ops.colsum(gy, xhat) if D % 4 == 0 else (gy * xhat).sum(0), (gy, xhat))
        if need_b:
            grads[m.bias] = self._pq.run(lambda: ops.colsum(gy) if D % 4 == 0 else gy.sum(0), (gy,))
        return gx

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        eng = self.eng
        if x.dim() != 4 or x.shape[1] not in (3, 6):
            raise ValueError(f"expected [N,6,H,W] (or [N,3,H,W] to be AddInverse-encoded), got {tuple(x.shape)}")
        ops.require_device(x, "bcos_hip.vit_train_plan")
        xd = x.detach()
        xd = xd if xd.is_contiguous() else xd.contiguous()
        with ops.transient_weights(), ops.no_absmax(), ops.absmax_arena(self._arena_f, xd.device):
            self._weights(xd.device)
            return self._forward(xd)

    def _forward(self, xd):
        eng = self.eng
        N, _, H, W = xd.shape
        p = eng.patch
        if H % p or W % p:
            raise ValueError(f"image size {H} x {W} is not a multiple of the patch size {p} (vit.py:166-167)")
        gh, gw = H // p, W // p
        T = gh * gw
        mean, std = eng._consts(xd.device)
        st = dict(head="vit", add_inverse=xd.shape[1] == 3, N=N, T=T, H=H, W=W)
        xn = ops.prep_input(xd, mean, std, cpad=8, add_inverse=st["add_inverse"], want_absmax=True)
        # patch embedding: Rearrange "(p1 p2 c)" + B-cos linear = a p x p / stride p B-cos convolution with the linear layer's epsilon
        e = _LinState()
        bias = self._lin_setup(eng.embed_mod, e)
        dim = e.w.shape[0]
        if e.w.shape[1] != p * p * 6:
            raise BcosHipError("vit train plan: the patch embedding must take 6-channel patches")
        w4 = ops.mark_static(F.pad(e.w.view(dim, p, p, 6), (0, 2)).contiguous())           # [dim, p, p, 8]
        geom = ops.fwd_geom(N, H, W, 8, dim, p, p, p, p, 0, 0)
        bcos = e.b != 1.0
        y = torch.empty((N * T, dim), device=xd.device, dtype=torch.float32)
        e.scale = torch.empty_like(y) if bcos else None
        e.norm = torch.empty((N * T,), device=xd.device, dtype=torch.float32) if bcos else None
        ops.tapconv(xn, w4, geom, out=y, scale_out=e.scale, norm_out=e.norm, bias=bias, bcos_mode=BCOS_LINEAR_EPS if bcos else 0, b=e.b,
                    flags=BCOS_EPI_FORCE_POW if e.force_pow else 0, track_absmax=False)
        e.x, e.y = xn, y
        st["embed"], st["w4"] = e, w4
        pe = eng._posemb(gh, gw, xd.device)
        cur = (y.view(N, T, dim) + pe).view(N * T, dim)
        blocks = []
        for blk in eng.blocks:
            rec = {}
            h1, rec["ln1"] = self._ln_fwd(blk["ln1"], cur)
            wq = blk["qkv"].weight.detach()
            rec["wq"] = wq
            wq_f = self._bank(blk["qkv"].weight, wq, False)
            qkv = ops.matmul_nt(h1, wq_f if wq_f is not None else ops.mark_static(wq if wq.is_contiguous() else wq.contiguous()), track_absmax=False)
            rec["h1"] = h1
            a, stats = ops.attention_fwd(qkv.view(N, T, -1), blk["heads"], blk["scale"], want_stats=True, want_absmax=True)
            rec["qkv"], rec["stats"], rec["a"] = qkv, stats, a
            a2 = a.view(N * T, -1)
            am = ops.absmax_of(a)
            if am is not None:
                ops._attach_absmax(a2, am)
            y_o, rec["out"] = self._lin_fwd(blk["out"].module, a2)
            x1 = y_o + cur
            h2, rec["ln2"] = self._ln_fwd(blk["ln2"], x1)
            y1, rec["l1"] = self._lin_fwd(blk["l1"].module, h2)
            if blk["act"] == 2:
                z, _ = ops.gelu_gate(y1)
                z = ops.ensure_absmax(z)
            else:
                z = ops.ensure_absmax(y1)
            y2, rec["l2"] = self._lin_fwd(blk["l2"].module, z)
            cur = y2 + x1
            blocks.append(rec)
        st["blocks"] = blocks
        if eng.gap_reorder:
            hN, st["head_ln"] = self._ln_fwd(eng.head_ln, cur)
            f, st["head_lin"] = self._lin_fwd(eng.head.module, hN)
            logits = ops.global_avgpool_logits(f.view(N, T, 1, -1), eng.logit_temperature, eng.logit_bias)
        else:
            pooled = cur.view(N, T, -1).mean(dim=1)
            hN, st["head_ln"] = self._ln_fwd(eng.head_ln, pooled)
            logits, st["head_lin"] = self._lin_fwd(eng.head.module, hN)
            if eng.logit_temperature is not None:
                logits = logits / eng.logit_temperature
            if eng.logit_bias is not None:
                logits = logits + eng.logit_bias
        return logits, st

    def backward(self, st, g_logits: torch.Tensor, need_x: bool = True):
        self._zeros.begin(g_logits.device)
        self._pq.begin(g_logits.device)
        try:
            with ops.transient_weights(), ops.no_absmax(), ops.absmax_arena(self._arena_b, g_logits.device):
                return self._backward(st, g_logits, need_x)
        finally:
            self._pq.end()
            self._zeros.end()

    def _backward(self, st, g_logits, need_x):
        eng = self.eng
        grads: Dict = {}
        N, T = st["N"], st["T"]
        inv_t = 1.0 if eng.logit_temperature is None else 1.0 / float(eng.logit_temperature)
        g_logits = g_logits.to(torch.float32)
        if eng.gap_reorder:
            K = g_logits.shape[1]
            gf = (g_logits * (inv_t / float(T))).view(N, 1, K).expand(N, T, K).contiguous().view(N * T, K)
            g_hN = self._lin_bwd(st["head_lin"], gf, grads)
            g = self._ln_bwd(st["head_ln"], g_hN, grads)
        else:
            g_hN = self._lin_bwd(st["head_lin"], (g_logits * inv_t).contiguous(), grads)
            g_pool = self._ln_bwd(st["head_ln"], g_hN, grads)
            g = (g_pool / float(T)).view(N, 1, -1).expand(N, T, g_pool.shape[1]).contiguous().view(N * T, -1)
        for bi in range(len(eng.blocks) - 1, -1, -1):
            blk, rec = eng.blocks[bi], st["blocks"][bi]
            gz = self._lin_bwd(rec["l2"], g, grads)
            gy1 = ops.gelu_bwd(gz, rec["l1"].y) if blk["act"] == 2 else gz
            gh2 = self._lin_bwd(rec["l1"], gy1, grads)
            g_x1 = self._ln_bwd(rec["ln2"], gh2, grads, addend=g if g.is_contiguous() else g.contiguous())
            ga = self._lin_bwd(rec["out"], g_x1, grads)
            gqkv = ops.attention_bwd(rec["qkv"].view(N, T, -1), rec["stats"], rec["a"], ga.view(N, T, -1), blk["heads"], blk["scale"])
            gq2 = gqkv.view(N * T, -1)
            wp = blk["qkv"].weight
            if wp.requires_grad:
                rows, Cq = gq2.shape
                h1 = rec["h1"]
                Cin = h1.shape[1]
                # (the ordered weight gradient WRITES its result: no zeroed accumulator; the atomics kernel -- mode f32, odd sizes -- needs one)
                acc = self._zeros.take((Cq, 1, 1, Cin), gq2.device) if not (ops.wgrad_is_ordered() and (Cq * 1 * 1 * Cin) % 4 == 0) else None
                grads[wp] = self._pq.run(lambda: ops.conv2d_wgrad(gq2.view(1, 1, rows, Cq), h1.view(1, 1, rows, Cin), Cin, Cq, (1, 1), (1, 1),
                                                                  (0, 0), (1, 1), out=acc).view(Cq, Cin), (gq2, h1, acc))
            wq_t = self._bank(wp, rec["wq"], True)
            gh1 = ops.matmul_nt(ops.ensure_absmax(gq2), wq_t if wq_t is not None else ops.mark_static(rec["wq"].t().contiguous()), track_absmax=False)
            g = self._ln_bwd(rec["ln1"], gh1, grads, addend=g_x1)
            st["blocks"][bi] = None
        # patch embedding (the positional table is a constant): a p x p / stride p convolution over the NHWC input
        e = st["embed"]
        p = eng.patch
        H, W = st["H"], st["W"]
        gh_, gw_ = H // p, W // p
        dim = e.w.shape[0]
        glin, rnorm = self._lin_glin(e, g, want_absmax=need_x)
        gl4 = _pad4(glin)
        gl4 = (gl4 if gl4.is_contiguous() else gl4.contiguous()).view(N, gh_, gw_, -1)
        lin = e.mod.linear
        if lin.weight.requires_grad:
            grads[lin.weight] = self._pq.run(lambda: ops.conv2d_wgrad(gl4, e.x, 8, dim, (p, p), (p, p), (0, 0), (1, 1))     # [dim, p, p, 8]
                                             [..., :6].reshape(dim, p * p * 6), (gl4, e.x))
        if lin.bias is not None and lin.bias.requires_grad:
            grads[lin.bias] = self._pq.run(lambda: ops.colsum(gl4.view(-1, gl4.shape[3]))[:dim].contiguous(), (gl4,))
        gx = None
        if need_x:
            addend = None
            if rnorm is not None:
                addend = ops.patch_norm_bwd(e.x, rnorm.view(N, gh_, gw_), 8, (p, p), (p, p), (0, 0), (1, 1))
            w_oihw = st["w4"].permute(0, 3, 1, 2)                                              # [dim, 8, p, p]
            r = (-dim) % 4
            if r:
                w_oihw = torch.cat([w_oihw, w_oihw.new_zeros((r,) + tuple(w_oihw.shape[1:]))], 0)
            plan = ops.DgradPlan(w_oihw, (p, p), (0, 0), (1, 1))
            am = ops.absmax_of(glin)
            g4 = gl4 if (r or dim % 4) else glin.view(N, gh_, gw_, dim)
            if am is not None and ops.absmax_of(g4) is None and g4.shape[-1] == dim:
                ops._attach_absmax(g4, am)
            gxn = plan.run(g4, H, W, addend=addend)                                            # [N, H, W, 8]
            _, std = eng._consts(g_logits.device)
            g6 = gxn[..., :6].permute(0, 3, 1, 2) / std.view(1, 6, 1, 1)
            gx = (g6[:, :3] - g6[:, 3:]).contiguous() if st["add_inverse"] else g6.contiguous()
        return gx, grads


def train_forward(eng, x: torch.Tensor):
    """`net(x)` in train() mode through the plan, or None when the network is outside the plan's scope (per-layer path then)."""
    from .train_plan import _TrainStepFn
    plan = getattr(eng, "_train_plan", None)
    if plan is None:
        ok, _ = ViTTrainPlan.supported(eng)
        plan = ViTTrainPlan(eng) if ok else False
        eng._train_plan = plan
    if plan is False or not plan.active():
        return None
    return _TrainStepFn.apply(plan, x, *plan.parameters())
