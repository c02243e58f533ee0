"""Training step of a B-cosified ResNet as ONE launch plan.

The reference's training step (bcos/training/trainer.py:666-784: forward in train() mode, criterion, backward; DDP at :916-918)
runs the same modules as inference.  Here `net.train(); net(x)` on a network with an attached ResNetEngine no longer falls back
to one autograd node per layer: the whole network is ONE autograd.Function (`_TrainStepFn`) whose forward and backward walk the
engine's layer list and issue the HIP launches directly, on NHWC tensors, keeping exactly what the backward needs.

Per convolution unit (BcosifyConv2d -> BatchNormUncentered2d [-> + shortcut] [-> ReLU], bcosifyconv2d.py:50-102,
batchnorm_uncentered.py:36-44):
  forward    contraction + B-cos scale with the scale NOT detached (one bcos_tapconv launch: y, s, patch norms), batch statistics of
             y in a fixed summation order (bcos_colsum_ws: mean, centred variance), running_var update, then ONE elementwise launch
             for normalisation + affine [+ shortcut] + ReLU (bcos_channel_affine / bcos_channel_affine_add)
  backward   ReLU gate, the two column sums of the BatchNorm gradient, its input gradient (bcos_channel_axpby), the derivative
             of the dynamic scale (bcos_train_scale_bwd), the patch-norm term (bcos_patch_norm_bwd, added by the input-gradient
             launch's epilogue), the input gradient (ops.DgradPlan) and the weight gradient (bcos_conv2d_wgrad).
The arithmetic is the per-layer path's (bcos/modules/_hipfn.py: BcosConv2dFn, batchnorm_uncentered.py: _BatchStatsFn) launch for
launch, so both are held to the same reference-recorded fixtures (tests/golden/resnet18_train_step.npz).

Scope: torchvision-topology ResNets with the GAP + 1x1 `fc` head (BASELINE configs) and, since round 5, CLIP's ModifiedResNet
(CLIP/clip/model.py:10-154: three-convolution stem, anti-aliasing AvgPool2d between conv2 and conv3 and in front of the shortcut
convolution; its attention-pool head runs as the module it is, under autograd, on the feature map the plan returns); groups == 1,
max_out == 1, fixed B.  Anything else -- the `attn_unpool` head, grouped / MaxOut layers, a learnable exponent, native unit-norm
layers -- is refused by `supported()` and keeps the per-layer path.
"""
import contextlib
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .lib import BCOS_CONV_EPS, BCOS_EPI_FORCE_POW, BCOS_NONE, BcosHipError


def _pad4(t: torch.Tensor) -> torch.Tensor:
    r = (-t.shape[-1]) % 4
    return F.pad(t, (0, r)) if r else t


_SIDE_STREAM = os.environ.get("BCOS_TRAIN_SIDE_STREAM", "1") != "0"
_PLAN_ARENAS = os.environ.get("BCOS_TRAIN_ARENAS", "1") != "0"        # development A/B: 0 = every operand-maxima tensor of a training pass its own zero fill (round 5)
_WEIGHT_BATCH = os.environ.get("BCOS_TRAIN_WEIGHT_BATCH", "1") != "0"  # development A/B: 0 = every layer's banks and images from launches of its own, in front of the launch that reads them (round 5)


class ParamGradQueue:
    """Parameter gradients off the critical path of a backward pass: the chain of input gradients runs on the caller's stream, the
    launches nothing else in the pass waits for -- weight gradients, bias / affine column sums -- on a second stream that follows it
    (they are short of a machine-filling size at the reference's batch 64 per GPU: 200-600 workgroups, and 20-25 % of the kernel time
    of a step).  `run(fn, reads)` issues fn() there once everything enqueued so far on the caller's stream has finished; `end()` makes
    the caller's stream wait for all of it.  The tensors a launch reads are recorded on the side stream (the caching allocator must
    not hand their blocks out again before it has passed), the results on the caller's.  CPU tensors (emulated kernels): inline."""

    def __init__(self):
        self._streams = {}
        self.main = self.side = None
        self._outs = []

    def begin(self, device):
        self.main = self.side = None
        self._outs = []
        device = torch.device(device)
        if device.type != "cuda" or not _SIDE_STREAM or torch.cuda.is_current_stream_capturing():
            return
        key = str(device)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        self.main, self.side = torch.cuda.current_stream(device), self._streams[key]

    def run(self, fn, reads=()):
        if self.side is None:
            return fn()
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            out = fn()
        for t in reads:
            if t is not None:
                t.record_stream(self.side)
        self._outs.append(out)
        return out

    def end(self):
        if self.side is None:
            return
        self.main.wait_stream(self.side)
        for o in self._outs:
            for t in (o if isinstance(o, (tuple, list)) else (o,)):
                if torch.is_tensor(t):
                    t.record_stream(self.main)
        self._outs = []
        self.main = self.side = None


class _UnitState:
    __slots__ = ("x", "y", "scale", "norm", "w", "bias", "b", "force_pow", "mean", "rstd", "g", "act", "relu", "bn", "conv",
                 "has_addend", "in_hw", "batch_stats", "dplan")


class ResNetTrainPlan:
    def __init__(self, eng):
        self.eng = eng
        self.net = eng.net
        ok, why = self.supported(eng)
        if not ok:
            raise BcosHipError(f"train plan: {why}")
        self._pq = ParamGradQueue()
        self._zeros = ops.ZeroArena()
        # operand maxima of a pass from ONE zero fill (a forward emitted 54 fills of its own on a ResNet-50): one arena per direction -- a
        # second forward before the first one's backward resets the forward arena only, and nothing of the backward reads forward maxima
        self._arena_f, self._arena_b = ops.AbsmaxArena(), ops.AbsmaxArena()
        self._nbt = []                 # num_batches_tracked buffers of the pass: advanced by ONE launch at its end (53 on a ResNet-50 before)
        self._wbatch = None            # ops.WeightPrepBatch of the plan's layers (None: not built yet / not applicable), see _weights
        self._wunits = {}              # id(conv) -> (weight parameter, forward bank or None, input-gradient plan or None)

    # ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def supported(eng):
        from bcos.modules.bcosconv2d import NormedConv2d
        if getattr(eng, "head_kind", None) not in ("gap_fc", "attnpool"):
            return False, "only the global-average-pool + fc head and the (pooled) attention-pool head"
        if any(b.hybrid or not b.relu for b in eng.blocks):
            return False, "grouped / MaxOut / ReLU-free blocks keep the per-layer path"
        if not all(relu for _, relu in eng.stem):
            return False, "stem convolutions with ReLU"
        for c in eng._all_convs():
            m = c.module
            if isinstance(getattr(m, "b", None), torch.Tensor) and m.b.requires_grad:
                return False, "learnable exponent"
            if isinstance(m.linear, NormedConv2d):
                return False, "native unit-norm layers"
            if getattr(c, "groups", 1) != 1:
                return False, "grouped layers keep the per-layer path"
            if int(getattr(m, "max_out", 1)) != 1:          # (fused MaxOut nodes of the inference plan: the unit below takes cout_all for Cout)
                return False, "MaxOut layers keep the per-layer path"
        return True, ""

    def _weights(self, device):
        """Every weight bank and split image of the step from ONE launch (ops.WeightPrepBatch) at the top of the forward pass: the forward
        banks, and the input-gradient plans the backward pass will run (kept across steps: only their contents change).  A layer whose
        effective weight is not its parameter itself (or a contraction mode without f16x2 images) keeps the per-layer preparation."""
        if not _WEIGHT_BATCH or torch.device(device).type != "cuda" or ops._l.get_contraction_mode() != "f16x2":
            self._wunits = {}
            return
        if self._wbatch is None or self._wbatch.device != torch.device(device):
            batch, units = ops.WeightPrepBatch(device), {}
            first = next(iter(self.eng._all_convs()))
            for c in self.eng._all_convs():
                conv = c.module
                w, _ = conv._effective_weight_and_bias()
                p = conv.linear.weight
                if not (isinstance(p, nn.Parameter) and w.data_ptr() == p.data_ptr() and tuple(w.shape) == tuple(p.shape) and p.is_contiguous()
                        and p.dim() == 4 and getattr(c, "groups", 1) == 1):
                    continue
                wk = batch.add_forward(p)          # (the Parameter itself: the batch follows a replaced .data)
                dplan = None
                if c is not first and c.cin > 8:          # (the narrow stem gradient runs its one-launch forms over banks of their own)
                    dplan = batch.add_dgrad(p, c.stride, c.padding, c.dilation)
                units[id(c)] = (p, wk, dplan)
            self._wbatch, self._wunits = batch, units
        self._wbatch.run()

    def parameters(self) -> List[nn.Parameter]:
        """every parameter the plan reads, in a fixed order (the autograd Function's inputs)"""
        ps, seen = [], set()
        for c in self.eng._all_convs():
            for p in (c.module.linear.weight, c.module.linear.bias, getattr(c.bn, "weight", None), getattr(c.bn, "bias", None)):
                if isinstance(p, nn.Parameter) and id(p) not in seen:
                    seen.add(id(p))
                    ps.append(p)
        return ps

    # ------------------------------------------------------------------------------------------------------------------
    def _unit_fwd(self, c, x, addend=None, relu=True) -> (torch.Tensor, _UnitState):
        """one conv (+ BatchNorm) (+ shortcut) (+ ReLU) on NHWC tensors.  Every BatchNormUncentered2d follows ITS OWN train / eval state
        (batchnorm_uncentered.py:80-99: `self.training or no running statistics`): a layer put in eval() under net.train() -- frozen-BN
        fine-tuning -- normalises with its running variance, and neither running_var nor num_batches_tracked is touched."""
        conv, bn = c.module, c.bn
        batch_stats = bn is not None and (bn.training or (bn.running_mean is None and bn.running_var is None))
        w, bias = conv._effective_weight_and_bias()
        wd = w.detach()
        pre = self._wunits.get(id(c))
        if pre is not None and pre[1] is not None and pre[0] is conv.linear.weight and w.data_ptr() == pre[0].data_ptr():
            wk = pre[1]                    # made by the step's one weight launch (_weights)
        else:
            pre = None
            wk = ops.mark_static(_pad4(wd.permute(0, 2, 3, 1)).contiguous())
        N, H, W, Cp = x.shape
        Cout = wd.shape[0]
        kh, kw = c.k
        Ho, Wo = c.out_hw(H, W)
        b = float(conv._b_value())
        force_pow = bool(conv._scaling()[1]) if hasattr(conv, "_scaling") else False
        y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
        scale = torch.empty_like(y) if b != 1.0 else None
        norm = torch.empty((N, Ho, Wo, 1), device=x.device, dtype=torch.float32) if b != 1.0 else None
        geom = ops.fwd_geom(N, H, W, Cp, Cout, kh, kw, c.stride[0], c.stride[1], c.padding[0], c.padding[1], c.dilation[0], c.dilation[1])
        ops.tapconv(x, wk, geom, out=y, scale_out=scale, norm_out=norm, bias=bias.detach() if bias is not None else None,
                    bcos_mode=BCOS_NONE if b == 1.0 else BCOS_CONV_EPS, b=b, flags=BCOS_EPI_FORCE_POW if force_pow else 0,
                    track_absmax=False)
        st = _UnitState()
        st.conv, st.bn, st.x, st.y, st.scale, st.norm, st.w, st.bias, st.b, st.force_pow = c, bn, x, y, scale, norm, wd, bias, b, force_pow
        st.relu, st.has_addend, st.in_hw, st.batch_stats = relu, addend is not None, (H, W), batch_stats
        st.dplan = pre[2] if pre is not None else None
        st.mean = st.rstd = st.g = None
        if bn is None:
            out = y
            if addend is not None:
                out = out + addend
            if relu:
                out = torch.relu(out)
            st.act = out if relu else None
            return out, st
        y2 = y.view(-1, Cout)
        if batch_stats:
            momentum = 0.0 if bn.momentum is None else bn.momentum
            if bn.track_running_stats and bn.num_batches_tracked is not None:
                if bn.momentum is None:                          # cumulative average: the count is needed on the host, now
                    bn.num_batches_tracked.add_(1)
                    momentum = 1.0 / float(bn.num_batches_tracked)
                else:
                    self._nbt.append(bn.num_batches_tracked)
            # ONE pass over y: mean, centred variance (x.var(unbiased=False), batchnorm_uncentered.py:36-44), 1 / std, weight / std and
            # the running_var update -- was two column-sum passes and ~10 torch launches per layer
            rv = bn.running_var if (bn.track_running_stats and bn.running_var is not None) else None
            mean, _, rstd, g = ops.bn_batch_stats(y2, bn.weight.detach() if bn.weight is not None else None, bn.eps, running_var=rv,
                                                  momentum=momentum)
        else:
            if bn.running_var is None:
                raise BcosHipError("train plan: a BatchNormUncentered2d in eval() needs its running_var")
            mean = torch.zeros(Cout, device=x.device)
            rstd = torch.rsqrt(bn.running_var.detach() + bn.eps)
            g = (rstd if bn.weight is None else bn.weight.detach() * rstd).contiguous()
        shift = bn.bias.detach().contiguous() if isinstance(bn.bias, torch.Tensor) else None
        # normalisation + affine (+ shortcut) + ReLU in one pass that also leaves the per-pixel maxima of its output: the contraction
        # that reads it runs the 3-product split-f16 loop (bf16x3 -- 6 products -- without them)
        out = ops.channel_affine_rows(y, g, shift, None if addend is None else (addend if addend.is_contiguous() else addend.contiguous()),
                                      relu=relu)
        st.mean, st.rstd, st.g, st.act = mean, rstd, g, (out if relu else None)
        return out, st

    def _unit_bwd(self, st: _UnitState, ga: torch.Tensor, grads: Dict, need_x: bool = True, extra: Optional[torch.Tensor] = None):
        """-> (gradient w.r.t. the unit's input or None, gradient w.r.t. the shortcut addend or None).  `extra`: a gradient that reaches
        the unit's INPUT by another path (the shortcut of a residual block) -- added by the launches that finish the input gradient
        (the patch-norm term, or the input-gradient launch's epilogue) instead of by an elementwise pass of its own."""
        c, bn = st.conv, st.bn
        conv = c.module
        N, Ho, Wo, Cout = st.y.shape
        y2 = st.y.view(-1, Cout)
        ga = ga if ga.is_contiguous() else ga.contiguous()
        if bn is not None:
            # ONE pass: the ReLU gate (also the gradient that reaches the shortcut) and the column sums of the BatchNorm backward
            # (batchnorm_uncentered.py: _BatchStatsFn.backward), the weight gradient and the variance-term coefficient from its finish
            has_w, has_b = isinstance(bn.weight, nn.Parameter), isinstance(bn.bias, nn.Parameter)
            need_var = not (bn.detach or not st.batch_stats)      # (variance a constant in explanation mode, or for a layer in eval())
            ga2, sgx, sg, gw, coef = ops.relu_bwd_colsums(ga.view(-1, Cout), st.act.view(-1, Cout) if st.relu else None, y2,
                                                          rstd=st.rstd, gvec=st.g, want_sg=has_b and bn.bias.requires_grad,
                                                          want_gw=has_w and bn.weight.requires_grad, want_coef=need_var)
            ga = ga2.view(N, Ho, Wo, Cout)
            if gw is not None:
                grads[bn.weight] = gw
            if sg is not None:
                grads[bn.bias] = sg
            # the norm's input gradient gy = ga g + (y - mean) coef: formed inside the scale derivative's launch where there is one
            # (bcos_train_scale_bwd_bn), else by its own elementwise pass
            fuse_bn = st.b != 1.0 and not conv.detach and Cout % 4 == 0
            bn_terms = (st.g, st.mean.contiguous() if need_var else None, coef if need_var else None)
            if fuse_bn:
                gy = None
            elif not need_var:
                gy = ops.channel_affine(ga, st.g, None)
            else:
                gy = ops.channel_axpby(ga, st.g, st.y, st.mean.contiguous(), coef)
        else:
            if st.relu:
                ga = ops.relu_bwd(ga, st.act)
            gy = ga
            fuse_bn = False
        g_addend = ga if st.has_addend else None
        g2 = ga.reshape(-1, Cout)
        m = y2.shape[0]
        # the convolution: scale derivative, patch-norm term, weight / bias / input gradients (_hipfn.BcosConv2dFn.backward)
        x = st.x
        H, W = st.in_hw
        cin = c.cin
        addend = extra if (extra is None or extra.is_contiguous()) else extra.contiguous()
        pn_rnorm = None                                   # the patch-norm term's per-patch factor: the input-gradient launch below adds the term
        if st.b != 1.0 and not conv.detach:
            if Cout % 4 == 0:          # (+ the per-pixel maxima of glin for the input-gradient launches)
                gl2, rnorm, _ = ops.train_scale_bwd((ga if fuse_bn else gy).reshape(-1, Cout).contiguous(), y2, st.scale.view(-1, Cout),
                                                    st.norm.view(-1), BCOS_CONV_EPS, st.b, st.force_pow, want_absmax=True,
                                                    bn=bn_terms if fuse_bn else None)
                glin = gl2.view(N, Ho, Wo, Cout)
                am = ops.absmax_of(gl2)
                if am is not None:
                    ops._attach_absmax(glin, am)
            else:
                from bcos.modules._hipfn import _scale_bwd_cols
                glin, rnorm, _ = _scale_bwd_cols(gy.reshape(-1, Cout), y2, st.scale.view(-1, Cout), st.norm.view(-1), BCOS_CONV_EPS,
                                                 dict(b=st.b, force_pow=st.force_pow), False)
                glin = glin.view(N, Ho, Wo, Cout)
            pn_rnorm = rnorm
        elif st.b != 1.0:
            glin = ops.mul(gy, st.scale)
        else:
            glin = gy
        lin = conv.linear
        gl4 = _pad4(glin).contiguous()
        if lin.weight.requires_grad:              # (on the side stream: nothing in the pass waits for a parameter gradient)
            # (the ordered weight gradient WRITES its result: no zeroed accumulator; the atomics kernel -- mode f32, odd sizes -- needs one)
            acc = self._zeros.take((Cout, c.k[0], c.k[1], cin), x.device) if not (ops.wgrad_is_ordered() and (Cout * c.k[0] * c.k[1] * cin) % 4 == 0) else None
            grads[lin.weight] = self._pq.run(lambda: ops.conv2d_wgrad(gl4, x, cin, Cout, c.k, c.stride, c.padding, c.dilation, out=acc)
                                             .permute(0, 3, 1, 2).contiguous(), (gl4, x, acc))     # [Cout,kh,kw,Cin] -> OIHW; `acc` too: a torch.zeros of the first pass
                                                                                                   # lives in the caller's pool and is accumulated into on the side stream
        if lin.bias is not None and lin.bias.requires_grad:
            grads[lin.bias] = self._pq.run(lambda: ops.colsum(gl4.view(-1, gl4.shape[3]))[:Cout].contiguous(), (gl4,))
        gx = None
        if need_x:
            plan = st.dplan
            if plan is None:
                wq = st.w
                r = (-wq.shape[0]) % 4                            # the dgrad K dimension (Cout) padded with zero filters
                if r:
                    wq = torch.cat([wq, wq.new_zeros((r,) + tuple(wq.shape[1:]))], 0)
                plan = ops.DgradPlan(wq, c.stride, c.padding, c.dilation)
            g_in = gl4 if Cout % 4 else glin.contiguous()
            if pn_rnorm is not None:
                gx = plan.run_with_patch_norm(g_in, x, pn_rnorm, cin, H, W, addend=addend)
            else:
                gx = plan.run(g_in, H, W, addend=addend)
        return gx, g_addend

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        self._nbt = []
        try:
            with ops.transient_weights(), (ops.absmax_arena(self._arena_f, x.device) if _PLAN_ARENAS else contextlib.nullcontext()):
                self._weights(x.device)
                return self._forward(x)
        finally:
            if self._nbt:
                torch._foreach_add_(self._nbt, 1)
                self._nbt = []

    def _forward(self, x: torch.Tensor):
        eng = self.eng
        if x.dim() != 4 or x.shape[1] not in (3, 6):
            raise ValueError(f"expected [N,6,H,W] (or [N,3,H,W] to be AddInverse-encoded), got {tuple(x.shape)}")
        ops.require_device(x, "bcos_hip.train_plan")
        xd = x.detach()
        xd = xd if xd.is_contiguous() else xd.contiguous()
        mean, std = eng._consts(x.device)
        st = dict(x=xd, add_inverse=xd.shape[1] == 3, H=xd.shape[2], W=xd.shape[3])
        xn = ops.prep_input(xd, mean, std, cpad=8, add_inverse=st["add_inverse"], want_absmax=True)
        a0, st["stem"] = xn, []
        for conv, _ in eng.stem:                      # torchvision: one 7 x 7 convolution; CLIP (CLIP/clip/model.py:94-154): three 3 x 3
            a0, u = self._unit_fwd(conv, a0, relu=True)
            st["stem"].append(u)
        st["a0_hw"] = (a0.shape[1], a0.shape[2])
        k, s, p = eng.pool
        cur = ops.avgpool2d_fwd(a0, k, s, p, want_absmax=True)
        blocks = []
        for blk in eng.blocks:
            inp = cur
            rec = dict(units=[], shortcut=None, in_hw=(inp.shape[1], inp.shape[2]), pre_pool_hw=None)
            h = inp
            for c in blk.convs[:-1]:
                h, u = self._unit_fwd(c, h, relu=True)
                rec["units"].append(u)
            if blk.pool:                               # CLIP's anti-aliasing pool between conv2 and conv3 (CLIP/clip/model.py:25,47)
                rec["pre_pool_hw"] = (h.shape[1], h.shape[2])
                h = ops.avgpool2d_fwd(h, blk.pool, blk.pool, 0, want_absmax=True)
            if blk.shortcut is not None:
                sc_in = ops.avgpool2d_fwd(inp, blk.shortcut_pool, blk.shortcut_pool, 0, want_absmax=True) if blk.shortcut_pool else inp
                idn, rec["shortcut"] = self._unit_fwd(blk.shortcut, sc_in, relu=False)
            else:
                idn = inp
            out, u = self._unit_fwd(blk.convs[-1], h, addend=idn, relu=True)
            rec["units"].append(u)
            blocks.append(rec)
            cur = out
        st["blocks"] = blocks
        if eng.head is None:                           # attention-pool head: the feature map leaves the plan, the head runs under autograd
            st["head"] = None
            return cur.permute(0, 3, 1, 2), st
        f, st["head"] = self._unit_fwd(eng.head, cur, relu=False)
        st["feat_hw"] = (f.shape[1], f.shape[2])
        logits = ops.global_avgpool_logits(f, eng.logit_temperature, eng.logit_bias)
        return logits, st

    def backward(self, st, g_logits: torch.Tensor, need_x: bool = True):
        self._zeros.begin(g_logits.device)
        self._pq.begin(g_logits.device)
        try:
            with ops.transient_weights(), (ops.absmax_arena(self._arena_b, g_logits.device) if _PLAN_ARENAS else contextlib.nullcontext()):
                return self._backward(st, g_logits, need_x)
        finally:
            self._pq.end()
            self._zeros.end()

    def _backward(self, st, g_logits: torch.Tensor, need_x: bool = True):
        eng = self.eng
        grads: Dict = {}
        if st["head"] is None:                         # gradient of the feature map [N, C, H, W] from the head's own autograd nodes
            g = g_logits.to(torch.float32).permute(0, 2, 3, 1).contiguous()
        else:
            fh, fw = st["feat_hw"]
            N = g_logits.shape[0]
            inv_t = 1.0 if eng.logit_temperature is None else 1.0 / float(eng.logit_temperature)
            gf = (g_logits.to(torch.float32) * (inv_t / float(fh * fw))).view(N, 1, 1, -1).expand(N, fh, fw, g_logits.shape[1]).contiguous()
            g, _ = self._unit_bwd(st["head"], gf, grads)
        for bi in range(len(eng.blocks) - 1, -1, -1):
            blk, rec = eng.blocks[bi], st["blocks"][bi]
            units = rec["units"]
            gh, g_idn = self._unit_bwd(units[-1], g, grads)
            if rec["shortcut"] is not None:
                g_idn, _ = self._unit_bwd(rec["shortcut"], g_idn, grads)
                if blk.shortcut_pool:
                    g_idn = ops.avgpool2d_bwd(g_idn.contiguous(), rec["in_hw"][0], rec["in_hw"][1], blk.shortcut_pool, blk.shortcut_pool, 0)
            if blk.pool:
                gh = ops.avgpool2d_bwd(gh.contiguous(), rec["pre_pool_hw"][0], rec["pre_pool_hw"][1], blk.pool, blk.pool, 0)
            # the shortcut's gradient joins the main path's inside the launches that finish the block input's gradient
            for u in reversed(units[1:-1]):
                gh, _ = self._unit_bwd(u, gh, grads)
            g, _ = self._unit_bwd(units[0], gh, grads, extra=g_idn)
        k, s, p = eng.pool
        a_h, a_w = st["a0_hw"]
        gl = ops.avgpool2d_bwd(g.contiguous(), a_h, a_w, k, s, p)
        for si in range(len(st["stem"]) - 1, -1, -1):
            gl, _ = self._unit_bwd(st["stem"][si], gl, grads, need_x=(need_x or si > 0))
        gxn = gl                                        # [N,H,W,6]: w.r.t. the normalised, AddInverse-encoded input
        gx = None
        if need_x:
            _, std = eng._consts(g_logits.device)
            g6 = gxn.permute(0, 3, 1, 2) / std.view(1, 6, 1, 1)            # d/dx of (x - mean) / std (bcosify.py:43)
            gx = (g6[:, :3] - g6[:, 3:]).contiguous() if st["add_inverse"] else g6.contiguous()      # AddInverse: [x, 1 - x]
        return gx, grads


class _TrainStepFn(torch.autograd.Function):
    """logits = plan(x) with the whole backward pass as ONE autograd node: d/dx and d/d(every parameter) from the plan's own launches"""

    @staticmethod
    def forward(ctx, plan, x, *params):
        logits, st = plan.forward(x)
        ctx.plan, ctx.st, ctx.params = plan, st, params
        ctx.need_x = ctx.needs_input_grad[1]
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        gx, grads = ctx.plan.backward(ctx.st, g_logits if ctx.st["head"] is None else g_logits.contiguous(), need_x=ctx.need_x)
        ctx.st = None
        return (None, gx) + tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)


def train_forward(eng, x: torch.Tensor) -> Optional[torch.Tensor]:
    """`net(x)` in train() mode through the plan, or None when the network is outside the plan's scope (per-layer path then)."""
    plan = getattr(eng, "_train_plan", None)
    if plan is None:
        ok, _ = ResNetTrainPlan.supported(eng)
        plan = ResNetTrainPlan(eng) if ok else False
        eng._train_plan = plan
    if plan is False:
        return None
    out = _TrainStepFn.apply(plan, x, *plan.parameters())
    if eng.head is None:
        # attention-pool head (bcos/modules/bcosattnpool.py:33-59 with nothing detached): the module itself, under autograd, on the
        # feature map the plan hands out; then the LogitLayer as BcosifyNetwork.forward applies it (bcosify.py:50-53)
        out = eng.attnpool(out)
        ll = eng.net.logit_layer
        out = ll(out) if ll else out
    return out
