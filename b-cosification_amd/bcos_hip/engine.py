"""Fused forward / explanation engine for B-cosified ResNets (torchvision topology).

The nn.Module graph stays the single source of truth for parameters (state dicts load into it unchanged);
`attach(net)` compiles it into a flat launch plan in which
  * every BcosifyConv2d + BatchNormUncentered2d (+ residual add) (+ ReLU) is ONE fused implicit-GEMM launch
    (reference: 9 + 3 + 1 + 1 ATen launches, bcosifyconv2d.py:68-101, batchnorm_uncentered.py:46-60),
  * AddInverse + Normalize + NCHW->NHWC is one streaming kernel, AdaptiveAvgPool + LogitLayer another,
  * the explanation pass (bcos/common.py:163-181) is one input-gradient launch per layer whose epilogue
    already multiplies by the stored scale of the layer below, adds the shortcut gradient and applies the
    ReLU gate, so each gradient tensor is written once and read once,
  * the last kernel turns the gradient w.r.t. the normalised NHWC input into W(x) [N,6,H,W] and the
    contribution map [N,H,W].
Activations are NHWC fp32; the stem's 6 input channels are zero-padded to 8.

What is stored between forward and backward (explanation mode), per B-cos layer: t = s * bn_scale * relu_gate,
the derivative of the layer's (post-BN, post-ReLU) output w.r.t. its pre-scale contraction `lin` with the
dynamic scale s = |lin| / ||patch|| held constant -- exactly what `.detach()` does in the reference.
"""
import math
import os
import threading
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .lib import (BCOS_CONV_EPS, BCOS_EPI_FORCE_POW, BCOS_EPI_GATE2_FROM_MUL, BCOS_EPI_MUL_FROM_ACT, BCOS_EPI_SCALE_GATE_LSB,
                  BCOS_NONE, BcosHipError)

_GATE_TENSOR = bool(os.environ.get("BCOS_GATE_TENSOR"))   # development switch: ReLU gates as tensors, not as the bit in t
# The multiplier t of conv1 / conv2 of every block (B = 2, own ReLU, no residual) is REBUILT from the kept activation, the patch
# norms and the BN scale / shift (BCOS_EPI_MUL_FROM_ACT) instead of being stored: one output-sized write less per such forward
# launch (2.8 GB per ResNet-50 step at batch 256) and 32 output-sized tensors less to keep.  With the general epilogue the
# rebuild cost +0.76 ms in the issue-bound gradient epilogues against -0.57 ms in the forward; with the specialised kinds
# (csrc/bcos_tapconv.hip: EF_MULACT) the step time is the same or slightly lower (same-node A/B: 29.60 vs 29.60, 29.48 vs 29.58
# ms).  BCOS_STORE_T=1 keeps the stored multipliers (same results to 1e-5: tests run both).
_HEAD_RANK1 = os.environ.get("BCOS_HEAD_RANK1", "1") != "0"     # 0: the head gradient of rounds 1-4 (one-hot tensor + K = 1000 contraction), for A/B
_SUB_ADDEND = not os.environ.get("BCOS_NO_SUB_ADDEND")     # development switch: scatter shortcut gradients into full-size tensors
_STORE_T = bool(os.environ.get("BCOS_STORE_T"))
# The rebuild recovers s from |a - bn_shift|: where |bn_scale s lin| << |bn_shift| the subtraction cancels (a = fl(csc v + csh)
# has absorbed v), leaving an ABSOLUTE error of ~sqrt(ulp(csh) / (|csc| norm)) in t where the stored multiplier was exact.  It is
# therefore only used for layers whose BN shift is small against their BN scale -- max|csh| <= this factor x median|csc| --
# (the calibrated synthetic networks; real checkpoints with large BN biases keep the stored multipliers).
_REBUILD_MAX_SHIFT = float(os.environ.get("BCOS_REBUILD_MAX_SHIFT", "0.5"))
# Images are independent in eval mode: a batch of at least 2 x _SUBBATCH_MIN images is split into _SUBBATCH_STREAMS contiguous
# sub-batches whose passes are issued on their own HIP streams.  The launches of one sub-batch fill the tails of the other's
# (a launch ends with most CUs idle while its last tiles finish; 117 + ~15 launches per step) and its HBM-bound launches overlap
# the other's matrix-bound ones: same-node A/B at ResNet-50 batch 256: 26.13 -> 24.63 ms per step with 2 streams (4: 24.68),
# results bit-identical (an image's result does not depend on what else is in its batch: test_determinism_and_batch_independence).
_SUBBATCH_STREAMS = int(os.environ.get("BCOS_SUBBATCH_STREAMS", "2"))
_SUBBATCH_MIN = int(os.environ.get("BCOS_SUBBATCH_MIN", "32"))


def _drive(gen):
    """run a pass generator to its end and hand back its return value"""
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


_ISSUING = threading.Lock()        # ops._ARENA is a module global switched per generator resume: ONE pass sequence is issued at a time


def _interleave(gens, streams, arenas, device):
    """Issue the passes of several sub-batches INTERLEAVED: generator i is resumed under stream i and maxima arena i, one block at a
    time, round-robin.  Issued one whole pass after the other, the second stream's first launch is queued only when the host is done
    with the first pass -- 4 ms into a ResNet-50 step that takes 23 (scripts/probe/host_bound_probe.py); back-to-back steps hide that
    behind the previous step, an isolated call does not.  The launches of a pass and their order on its stream are unchanged.
    Threading: like the reference's `explanation_mode` context (bcos/common.py:347-384 mutates module state) this is single-threaded by
    design -- the maxima arena is a module global of bcos_hip.ops; a second Python thread entering while a pass sequence is being
    issued raises instead of corrupting both passes silently.
    Errors: when one pass generator raises, the others are closed and the caller's stream is ordered behind every side stream BEFORE
    the exception propagates -- the half-issued launches write into output tensors the caller may free or reuse."""
    if not _ISSUING.acquire(blocking=False):
        raise BcosHipError("engine: explain() / forward() re-entered from a second thread while a pass sequence is being issued "
                           "(the fused plan is single-threaded, like the reference's explanation_mode context)")
    n = len(gens)
    results, live = [None] * n, list(range(n))
    prev = ops._ARENA
    try:
        for i in range(n):
            with torch.cuda.stream(streams[i]):          # (the zero fill of pass i's maxima is ordered on pass i's stream)
                arenas[i].reset(device)
        while live:
            for i in list(live):
                ops.set_absmax_arena(arenas[i])
                with torch.cuda.stream(streams[i]):
                    try:
                        next(gens[i])
                    except StopIteration as stop:
                        results[i] = stop.value
                        live.remove(i)
    except BaseException:
        for g in gens:
            g.close()
        cur = torch.cuda.current_stream()
        for st in streams[:n]:
            cur.wait_stream(st)
        raise
    finally:
        ops.set_absmax_arena(prev)
        _ISSUING.release()
    return results


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


class _Conv:
    """Kernel-side view of one BcosifyConv2d (+ the BatchNormUncentered2d that follows it)."""

    def __init__(self, conv, bn, cin_pad_to=4, main_path=False):
        from bcos.modules.bcosconv2d import BcosConv2d
        if not isinstance(conv, BcosConv2d):
            raise BcosHipError(f"engine: expected a B-cos conv, got {type(conv).__name__}")
        lin = conv.linear
        if not _fusable(conv) or (conv.max_out != 1 and not main_path):      # (only a block's main path walks MaxOut layers)
            raise BcosHipError("engine: this grouped / MaxOut layer runs through the module path only")
        if lin.padding_mode != "zeros":
            raise BcosHipError("engine: only zero padding")
        self.module, self.bn = conv, bn
        self.stride, self.padding, self.dilation = tuple(lin.stride), tuple(lin.padding), tuple(lin.dilation)
        self.k = tuple(lin.kernel_size)
        # MaxOut (bcosconv2d.py:166-170; round 4): the contraction is max_out times as wide as the layer's output -- the launch takes
        # the max over each unit's adjacent filters before the B-cos scale and keeps the scale at the winning filter (d out / d lin)
        self.max_out = int(conv.max_out)
        self.reads_image_range = ops.reads_image_range(lin.kernel_size, lin.stride, lin.dilation, lin.groups)    # (forward and input gradient alike)
        self.cin, self.cout_all = lin.in_channels, lin.out_channels
        self.cout = lin.out_channels // self.max_out
        self._wide = None
        # grouped layers (ResNeXt-style `groups`, bcosconv2d.py:84-140; round 4): one grouped launch forward, one per parity class
        # backward (bcos_tapconv_geom.groups) -- every group with its own patch norm, B-cos scale and transposed filters
        self.groups = int(lin.groups)
        if self.groups > 1 and not _grouped_ok(lin):
            raise BcosHipError("engine: grouped layers need in_channels / groups and out_channels / groups to be multiples of 4")
        self.refresh()

    def fingerprint(self):
        """(storage, in-place version) of every tensor refresh() reads: a changed fingerprint means the kernel-side copies
        (weight layouts, pre-split images, dgrad plans, BN scale/shift, B) are stale.  Edits through `.data` of an existing
        storage do not bump the version counter and stay invisible -- call refresh() after those."""
        conv, bn = self.module, self.bn
        lin = conv.linear
        ts = [getattr(lin, "weight", None), getattr(lin, "bias", None), getattr(conv, "b", None), getattr(conv, "scale", None)]
        if bn is not None:
            ts += [bn.weight, bn.bias, bn.running_var, bn.running_mean]
        return tuple((t.data_ptr(), t._version) if isinstance(t, torch.Tensor) else t for t in ts) + \
            (getattr(conv, "use_weight_norm", None), bn.training if bn is not None else None)

    def refresh(self):
        conv, bn = self.module, self.bn
        self._fp = self.fingerprint()
        w, bias = conv._effective_weight_and_bias()
        w = w.detach()
        self.b = conv._b_value()
        # learnable-B variants take the general pow form even at B = 2 (bcosifyconv2d.py:91-98)
        self.force_pow = bool(conv._scaling()[1]) if hasattr(conv, "_scaling") else False
        cin_pad = (-self.cin) % 4 if self.groups == 1 else 0
        wk = w.permute(0, 2, 3, 1)
        if cin_pad:
            wk = torch.nn.functional.pad(wk, (0, cin_pad))
        self.w_fwd = ops.mark_static(wk.contiguous())     # rebuilt by refresh() when parameters change
        self.bias = bias.detach().contiguous() if bias is not None else None
        self.dgrad = ops.DgradPlan(w, self.stride, self.padding, self.dilation, groups=self.groups)
        if bn is not None:
            if bn.training:
                raise BcosHipError("engine: BatchNormUncentered2d must be in eval mode (call model.eval())")
            self.ch_scale, self.ch_shift = bn.channel_scale_shift()
        else:
            self.ch_scale = self.ch_shift = None
        # may the multiplier of this layer be rebuilt from its activation (see _REBUILD_MAX_SHIFT)?  Decided once per refresh.
        self.rebuild_ok = self.groups == 1          # (the rebuild takes ONE patch norm per pixel: a grouped layer has one per group)
        if self.rebuild_ok and self.ch_shift is not None and self.ch_scale is not None:
            self.rebuild_ok = bool(float(self.ch_shift.abs().max()) <= _REBUILD_MAX_SHIFT * float(self.ch_scale.abs().median()))
        ops.publish_cached(self.w_fwd)      # (a refresh inside a sub-batch pass runs on that pass's side stream; the other one reads the result)

    def out_hw(self, H, W):
        return (ops.conv_out_size(H, self.k[0], self.stride[0], self.padding[0], self.dilation[0]),
                ops.conv_out_size(W, self.k[1], self.stride[1], self.padding[1], self.dilation[1]))

    @property
    def k_fwd(self):
        """K of the forward contraction (its A operand is the layer input)."""
        return (self.cin // self.groups) * self.k[0] * self.k[1]

    @property
    def k_dgrad(self):
        """largest K of the input-gradient launches (their A operand is the gradient w.r.t. this layer's `lin`)."""
        return (self.cout_all // self.groups) * self.k[0] * self.k[1]

    def fwd(self, x, *, addend=None, relu=False, want_scale=False, gates=None, flags=0, track=None, keep_act=False):
        """-> (y, t).  `keep_act` (explanation state of a layer whose output stays alive anyway is cheaper to REBUILD than to
        store): when the layer allows it -- B = 2, plain |lin| / norm scale, own ReLU decided by the value, no residual -- t
        is not written; the returned `t` is then an _ActScale record (activation, patch norms, BN scale / shift) from which
        the consuming input-gradient launch rebuilds the multiplier (BCOS_EPI_MUL_FROM_ACT)."""
        gate = gates.pop(0) if (relu and gates is not None) else None
        if self.max_out > 1:
            return self._fwd_maxout(x, addend, relu, want_scale, gate, flags, track)
        rebuild = (keep_act and want_scale and relu and gate is None and addend is None and self.b == 2.0 and not self.force_pow
                   and self.rebuild_ok)
        y, t, nrm = ops.conv2d_fwd(x, self.w_fwd, stride=self.stride, padding=self.padding, dilation=self.dilation,
                                   bias=self.bias, b=self.b, mode=BCOS_CONV_EPS, ch_scale=self.ch_scale,
                                   ch_shift=self.ch_shift, addend=addend, relu=relu, relu_gate=gate,
                                   want_scale=want_scale and not rebuild, want_norm=rebuild,
                                   flags=flags | (BCOS_EPI_FORCE_POW if self.force_pow else 0), track_absmax=track,
                                   groups=self.groups)
        self.last_gate = gate       # the replayed gate tensor, if any (else the output itself encodes the gate)
        if rebuild:
            t = _ActScale(y, nrm, self.ch_scale, self.ch_shift)
        return y, t


    def _fwd_maxout(self, x, addend, relu, want_scale, gate, flags, track):
        """MaxOut layer: one launch for contraction + unit maximum + B-cos scale (include/bcos_hip.h: bcos_epilogue.max_out), then the
        BatchNorm fold, shortcut and ReLU as one elementwise launch (the fused MaxOut epilogue carries none of them).  -> (activation,
        narrow multiplier = BN scale x ReLU gate, the gate in its low mantissa bit where the consumer reads it there); the WIDE
        multiplier -- the scale at each unit's winning filter -- is left in `take_wide()` for the layer's own input-gradient launch."""
        N, H, W, _ = x.shape
        Ho, Wo = self.out_hw(H, W)
        g = ops.fwd_geom(N, H, W, self.w_fwd.shape[3], self.cout_all, self.k[0], self.k[1], self.stride[0], self.stride[1],
                         self.padding[0], self.padding[1], self.dilation[0], self.dilation[1])
        y = torch.empty((N, Ho, Wo, self.cout), device=x.device, dtype=torch.float32)
        t_wide = torch.empty((N, Ho, Wo, self.cout_all), device=x.device, dtype=torch.float32) if want_scale else None
        ops.tapconv(x, self.w_fwd, g, out=y, scale_out=t_wide, bias=self.bias, bcos_mode=BCOS_NONE if self.b == 1.0 else BCOS_CONV_EPS,
                    b=self.b, flags=BCOS_EPI_FORCE_POW if self.force_pow else 0, max_out=self.max_out)
        csc = self.ch_scale if self.ch_scale is not None else torch.ones(self.cout, device=x.device)
        own_relu = relu and gate is None
        if addend is not None:
            act = ops.channel_affine_add(y, csc, self.ch_shift if self.ch_shift is not None else torch.zeros_like(csc), addend,
                                         relu=own_relu)
        elif self.ch_scale is not None or own_relu:
            act = ops.channel_affine(y, csc, self.ch_shift, relu=own_relu)
        else:
            act = y
        open_ = None
        if relu:
            open_ = (gate > 0) if gate is not None else (act > 0)
            if gate is not None:
                act = act * open_
        self.last_gate = gate
        t = None
        if want_scale:
            t = csc.expand_as(act)
            if open_ is not None and (flags & BCOS_EPI_SCALE_GATE_LSB):       # gate in the low mantissa bit, a closed gate stores exactly 0
                t = torch.where(open_, (t.contiguous().view(torch.int32) | 1).view(torch.float32), torch.zeros((), device=x.device))
            elif open_ is not None:
                t = t * open_
            t = t.contiguous()
        self._wide = t_wide
        if track and ops.DEFAULT_TRACK_ABSMAX and _l_mode() == "f16x2":
            ops.ensure_absmax(act)
        return act, t

    def take_wide(self):
        """the wide multiplier the last fwd() of a MaxOut layer left behind (None for every other layer)"""
        w, self._wide = self._wide, None
        return w

    def expand(self, gl, wide):
        """gradient w.r.t. the layer's (narrow) output, already times its narrow multiplier -> gradient w.r.t. the contraction's
        columns: every unit's value at its winning filter, times the B-cos scale there (bcos_maxout_expand)"""
        if self.max_out == 1:
            return gl
        N, H, W, Cn = gl.shape
        out = ops.maxout_expand(gl.reshape(-1, Cn), wide.view(-1, Cn * self.max_out), self.max_out).view(N, H, W, Cn * self.max_out)
        return ops.ensure_absmax(out) if (self.k_dgrad >= ops.F16X2_MIN_K and ops.DEFAULT_TRACK_ABSMAX and _l_mode() == "f16x2") else out


def _grouped_ok(lin) -> bool:
    """can a grouped convolution be a fused node of the plan?  (the grouped launches move 16-byte pieces of every group's channels)"""
    return lin.groups == 1 or ((lin.in_channels // lin.groups) % 4 == 0 and (lin.out_channels // lin.groups) % 4 == 0)


def _l_mode() -> str:
    from . import lib as _lib
    return _lib.get_contraction_mode()


def _fusable(conv) -> bool:
    """can this B-cos convolution be a fused node of the plan?  Grouped: group widths multiples of four; MaxOut: 2 or 4 filters per
    unit (what the fused epilogue takes), ungrouped."""
    lin = conv.linear
    mo = getattr(conv, "max_out", 1)
    if mo == 1:
        return _grouped_ok(lin)
    return mo in (2, 4) and lin.groups == 1 and lin.out_channels % 4 == 0


class _ActScale:
    """What the explanation pass needs to rebuild t = s * bn_scale * relu_gate of a layer from its kept activation."""

    def __init__(self, act, norm, ch_scale, ch_shift):
        self.act, self.norm, self.ch_scale, self.ch_shift = act, norm, ch_scale, ch_shift

    def kwargs(self):
        return dict(mul=self.act, mul_norm=self.norm, mul_csc=self.ch_scale, mul_csh=self.ch_shift, flags=BCOS_EPI_MUL_FROM_ACT)


def _mul_kwargs(t):
    """epilogue arguments that multiply a gradient by the layer multiplier `t` (stored tensor or _ActScale record)"""
    return t.kwargs() if isinstance(t, _ActScale) else dict(mul=t)


class _Block:
    """A residual block: main-path convs (each followed by its norm; ReLU after all but possibly the last, which adds the
    shortcut first), an optional shortcut conv, and -- CLIP's anti-aliased variant (CLIP/clip/model.py:25,36-40) -- an
    AvgPool2d(stride) in front of the last main conv and in front of the shortcut conv."""

    def __init__(self, block):
        names = [n for n in ("conv1", "conv2", "conv3") if hasattr(block, n)]
        # Grouped convolutions (ResNeXt-style `groups`, bcosconv2d.py:84-140) and MaxOut ones (:166-170, 2 or 4 filters per unit) on the
        # main path are fused nodes like any other since round 4 (_Conv.groups, _Conv._fwd_maxout).  What the fused launches do not
        # take -- group widths that are not multiples of four, other unit sizes, grouped MaxOut, a MaxOut shortcut -- makes the block a
        # HYBRID node: it runs layer by layer on the nn.Module path (one fused HIP launch per layer: bcos/modules/_hipfn.py) and its input
        # gradient comes from that path's own backward; the blocks around it stay fused.
        def needs_module_path(m, main):
            if getattr(m, "linear", None) is None:
                return False
            return not _fusable(m) or (not main and getattr(m, "max_out", 1) != 1)       # (MaxOut shortcuts: module path)
        self.hybrid = any(needs_module_path(getattr(block, n), True) for n in names)
        ds0 = getattr(block, "downsample", None)
        if ds0 is not None:
            self.hybrid = self.hybrid or any(needs_module_path(m, False) for m in ds0.children())
        self.module = block
        self.k_first = 0
        if self.hybrid:
            self.convs, self.shortcut, self.pool, self.shortcut_pool, self.relu = [], None, 0, 0, True
            return
        self.convs = [_Conv(getattr(block, n), getattr(block, n.replace("conv", "bn")), main_path=True) for n in names]
        self.k_first = self.convs[0].k_fwd
        relus = [getattr(block, r) for r in ("relu", "relu1", "relu2", "relu3") if hasattr(block, r)]
        self.relu = all(isinstance(r, nn.ReLU) for r in relus)
        if not self.relu and not all(isinstance(r, nn.Identity) for r in relus):
            raise BcosHipError("engine: mixed activations inside a block")
        self.pool = 0
        ap = getattr(block, "avgpool", None)
        if isinstance(ap, nn.AvgPool2d):
            self.pool = _pool_stride(ap)
        ds = block.downsample
        self.shortcut = None
        self.shortcut_pool = 0
        if ds is not None:
            mods = list(ds.children())
            if len(mods) == 3 and isinstance(mods[0], nn.AvgPool2d):
                self.shortcut_pool = _pool_stride(mods[0])
                mods = mods[1:]
            if len(mods) != 2:
                raise BcosHipError("engine: downsample must be ([AvgPool2d,] conv, norm)")
            self.shortcut = _Conv(mods[0], mods[1])

    def refresh(self):
        for c in self.convs:
            c.refresh()
        if self.shortcut is not None:
            self.shortcut.refresh()


def _pool_stride(pool: nn.AvgPool2d) -> int:
    k, s, p = _pair(pool.kernel_size), _pair(pool.stride), _pair(pool.padding)
    if k != s or k[0] != k[1] or p != (0, 0) or pool.ceil_mode or pool.divisor_override is not None:
        raise BcosHipError("engine: only AvgPool2d(stride) anti-aliasing pools are supported inside blocks")
    # AvgPool2d(1) -- what CLIP's Bottleneck puts in front of the shortcut convolution of a block that only widens (layer1.0: stride 1) -- is
    # the identity: no launch, no copy of the block input (it was a 205 MB read + 205 MB write per step at batch 256), forward and backward
    return 0 if k[0] == 1 else k[0]


class ResNetEngine:
    """Launch plan for `BcosifyNetwork(ResNetBcos(...))` (bcosify.py:22-53 + standard_models.py:36-54)."""

    def __init__(self, net):
        m = net.model
        self.net = net
        for attr in ("conv1", "bn1", "layer1", "layer2", "layer3", "layer4"):
            if not hasattr(m, attr):
                raise BcosHipError(f"engine: {type(m).__name__} has no `{attr}`: not a ResNet-style network")
        if hasattr(m, "attnpool"):
            # CLIP ModifiedResNet (CLIP/clip/model.py:94-154): 3-conv stem + AvgPool2d(2), attention-pool head
            self.stem = [(_Conv(getattr(m, f"conv{i}"), getattr(m, f"bn{i}")), isinstance(getattr(m, f"relu{i}"), nn.ReLU))
                         for i in (1, 2, 3)]
            pool = m.avgpool
            # pooled head, or the `attn_unpool` variant that projects every location and L2-normalises it (bcosattnpool.py:23-32)
            self.head_kind = "attn_unpool" if getattr(m.attnpool, "attn_unpool", False) else "attnpool"
            self.head = None
            self.attnpool = m.attnpool
        else:
            for attr in ("relu", "maxpool", "fc", "avgpool"):
                if not hasattr(m, attr):
                    raise BcosHipError(f"engine: {type(m).__name__} has no `{attr}`: not a torchvision-style ResNet")
            self.stem = [(_Conv(m.conv1, m.bn1), isinstance(m.relu, nn.ReLU))]
            pool = m.maxpool
            self.head_kind = "gap_fc"
            self.head = _Conv(m.fc, None)
        if not isinstance(pool, nn.AvgPool2d):
            raise BcosHipError("engine: the stem pool must be nn.AvgPool2d (the B-cosification recipe swaps MaxPool for "
                               "AvgPool2d(3,2,1): bcosification/experiment_parameters.py:99)")
        self.pool = (_pair(pool.kernel_size)[0], _pair(pool.stride)[0], _pair(pool.padding)[0])
        if pool.ceil_mode or not pool.count_include_pad or pool.divisor_override is not None:
            raise BcosHipError("engine: unsupported AvgPool2d options")
        self.blocks: List[_Block] = []
        for li in range(1, 5):
            for blk in getattr(m, f"layer{li}").children():
                self.blocks.append(_Block(blk))
        norm = net.bcosifynormalize
        self._mean, self._std = tuple(norm.mean), tuple(norm.std)
        self._dev_consts = {}
        ll = net.logit_layer
        self.logit_bias = ll.logit_bias if ll is not None else None
        self.logit_temperature = ll.logit_temperature if ll is not None else None
        self.supports_explain = True
        self._absmax_arena = ops.AbsmaxArena()      # per-pixel operand maxima of one pass (f16x2 contraction)
        self.subbatch_streams = _SUBBATCH_STREAMS   # 1 = every pass on the caller's stream
        # (three streams, measured on the CLIP image encoder at batch 256: 8 650 / 8 705 against 8 651 / 8 608 images/s with two, same
        #  node -- within the noise; the ViT plan, whose launches are shorter still, does gain: bcos_hip/vit_engine.py)
        self._side = None                           # (streams, arenas) of the sub-batch passes, created on first use
        if self.head_kind in ("attnpool", "attn_unpool"):
            self._refresh_attnpool()

    def _attnpool_fingerprint(self):
        ap = self.attnpool
        projs = (ap.v_proj, ap.c_proj) if self.head_kind == "attn_unpool" else (ap.q_proj, ap.k_proj, ap.v_proj, ap.c_proj)
        ts = [p.weight for p in projs] + [getattr(p, "bias", None) for p in projs] + [getattr(ap.c_proj, "b", None)]
        return tuple((t.data_ptr(), t._version) if isinstance(t, torch.Tensor) else t for t in ts)

    def _refresh_attnpool(self):
        ap = self.attnpool
        self._ap_fp = self._attnpool_fingerprint()
        if self.head_kind == "attn_unpool":
            # per location: v_proj (plain nn.Linear WITH its bias) -> c_proj (B-cosified by the converter: B-cos linear) -> L2
            # normalisation with the norm held constant in explanation mode (bcosattnpool.py:23-32)
            wv = ops.mark_static(ap.v_proj.weight.detach().clone().contiguous())
            bv = ap.v_proj.bias
            cp = ap.c_proj
            if hasattr(cp, "_effective_weight_and_bias"):
                wc, bc = cp._effective_weight_and_bias()
                self.ap_cb = float(cp._b_value())
            else:                                   # an un-converted (plain) c_proj: B = 1
                wc, bc, self.ap_cb = cp.weight, cp.bias, 1.0
            wc = ops.mark_static(wc.detach().clone().contiguous())
            self.ap_w = dict(v=wv, c=wc)
            self.ap_bias = dict(v=bv.detach().clone().contiguous() if bv is not None else None,
                                c=bc.detach().clone().contiguous() if bc is not None else None)
            C, D = wv.shape[1], wc.shape[0]
            self.ap_cconv = ops.DgradPlan(wc.view(D, C, 1, 1), (1, 1), (0, 0), (1, 1))
            self.ap_vconv = _HeadConv(ops.DgradPlan(wv.view(C, C, 1, 1), (1, 1), (0, 0), (1, 1)), C)
            return
        # (c_proj may be a BcosifyLinear: .weight property); inference-constant copies: pre-split images, f16x2 contraction
        w = lambda lin: ops.mark_static(lin.weight.detach().clone().contiguous())   # noqa: E731
        self.ap_w = dict(q=w(ap.q_proj), k=w(ap.k_proj), v=w(ap.v_proj), c=w(ap.c_proj))
        self.ap_heads = ap.num_heads
        # explanation mode detaches q and k (bcosattnpool.py:37-39): the gradient reaches the feature map through v only,
        # i.e. through v_proj seen as a 1x1 convolution over the HW positions (+ the mean token, folded in _backward)
        C = self.ap_w["v"].shape[1]
        self.ap_cT = ops.mark_static(self.ap_w["c"].t().contiguous())          # [C, D]: g_pooled = g_emb @ W_c
        self.ap_vconv = _HeadConv(ops.DgradPlan(self.ap_w["v"].view(C, C, 1, 1), (1, 1), (0, 0), (1, 1)), C)

    def refresh(self):
        """Re-read parameters after they changed (load_state_dict, calibration, ...).  forward()/explain() call this
        themselves when a parameter's storage or in-place version differs from what the plan was built from."""
        for c, _ in self.stem:
            c.refresh()
        if self.head is not None:
            self.head.refresh()
        else:
            self._refresh_attnpool()
        for b in self.blocks:
            b.refresh()

    def _all_convs(self):
        for c, _ in self.stem:
            yield c
        if self.head is not None:
            yield self.head
        for b in self.blocks:
            yield from b.convs
            if b.shortcut is not None:
                yield b.shortcut

    def _ensure_fresh(self):
        """Cheap staleness check (a few hundred integer compares): refresh the layers whose parameters changed."""
        ops.publish_pending()          # (objects a training pass cached without publication are completed before an inference pass reads them)
        for c in self._all_convs():
            if c.fingerprint() != c._fp:
                c.refresh()
        if self.head_kind in ("attnpool", "attn_unpool") and self._attnpool_fingerprint() != getattr(self, "_ap_fp", None):
            self._refresh_attnpool()

    def _consts(self, device):
        key = str(device)
        if key not in self._dev_consts:
            self._dev_consts[key] = (torch.tensor(self._mean, dtype=torch.float32, device=device),
                                     torch.tensor(self._std, dtype=torch.float32, device=device))
            ops.publish_cached(self._dev_consts[key][1])
        return self._dev_consts[key]

    # ------------------------------------------------------------------------------------------------
    def _run_forward(self, x: torch.Tensor, keep: bool, gates=None):
        return _drive(self._run_forward_gen(x, keep, gates))

    def _run_forward_gen(self, x: torch.Tensor, keep: bool, gates=None):
        """The forward pass as a generator: it yields (nothing) behind the stem and behind every block, so that the passes of several
        sub-batches can be ISSUED interleaved (see _interleave); its return value is (head output, kept state)."""
        if x.dim() != 4 or x.shape[1] not in (3, 6):
            raise ValueError(f"expected [N,6,H,W] (or [N,3,H,W] to be AddInverse-encoded), got {tuple(x.shape)}")
        ops.require_device(x, "bcos_hip.engine")
        self._ensure_fresh()
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()
        mean, std = self._consts(x.device)
        add_inverse = x.shape[1] == 3
        xn = ops.prep_input(x, mean, std, cpad=8, add_inverse=add_inverse, want_absmax=True)
        gates = list(gates) if gates is not None else None
        if gates is not None and any(b.hybrid for b in self.blocks):
            raise BcosHipError("engine: replayed ReLU gates are not available for networks with a block on the nn.Module path (hybrid node)")
        st = dict(x=x, add_inverse=add_inverse, H=x.shape[2], W=x.shape[3], stem_ts=[], stem_hws=[], blocks=[]) if keep else None
        a = xn
        need = lambda k: k >= ops.F16X2_MIN_K      # noqa: E731  will the reader of a tensor use its per-pixel maxima?
        for si, (conv, relu) in enumerate(self.stem):
            if keep:
                st["stem_hws"].append((a.shape[1], a.shape[2]))
            nxt = self.stem[si + 1][0].k_fwd if si + 1 < len(self.stem) else 0       # the last stem conv feeds a pool
            with ops.image_range_reader(si + 1 < len(self.stem) and self.stem[si + 1][0].reads_image_range):
                a, t = conv.fwd(a, relu=relu, want_scale=keep, gates=gates, track=need(nxt))
            if keep:
                st["stem_ts"].append(t)
        k, s, p = self.pool
        cur = ops.avgpool2d_fwd(a, k, s, p, want_absmax=need(self.blocks[0].k_first))      # (maxima from the pool's own launch)
        if keep:
            st["a0_hw"] = (a.shape[1], a.shape[2])
        del a
        yield
        for bi, blk in enumerate(self.blocks):
            inp = cur
            rec = dict(in_hw=(inp.shape[1], inp.shape[2])) if keep else None
            if blk.hybrid:
                k_after = (self.blocks[bi + 1].k_first if bi + 1 < len(self.blocks)
                           else (self.head.k_fwd if self.head is not None else ops.F16X2_MIN_K))
                cur = self._hybrid_forward(blk, inp, rec, need(k_after))
                if keep:
                    st["blocks"].append(rec)
                yield
                continue
            h = inp
            ts, hws, tw = [], [], []
            for ci, c in enumerate(blk.convs[:-1]):
                hws.append((h.shape[1], h.shape[2]))
                pooled_next = blk.pool and ci == len(blk.convs) - 2              # a pool sits between this conv and the next
                with ops.image_range_reader(blk.convs[ci + 1].reads_image_range and not pooled_next):     # (a 3 x 3 layer reads this output: its per-image scales from this launch)
                    h, t = c.fwd(h, relu=blk.relu, want_scale=keep, gates=gates, keep_act=not pooled_next and not _STORE_T,
                                 track=need(blk.convs[ci + 1].k_fwd) and not pooled_next)
                ts.append(t)
                tw.append(c.take_wide())
            pre_pool_hw = (h.shape[1], h.shape[2])
            if blk.pool:
                h = ops.avgpool2d_fwd(h, blk.pool, blk.pool, 0, want_absmax=need(blk.convs[-1].k_fwd))
            if blk.shortcut is not None:
                sc_in = (ops.avgpool2d_fwd(inp, blk.shortcut_pool, blk.shortcut_pool, 0, want_absmax=need(blk.shortcut.k_fwd))
                         if blk.shortcut_pool else inp)
                idn, td = blk.shortcut.fwd(sc_in, relu=False, want_scale=keep, track=False)      # only ever an addend
            else:
                idn, td = inp, None
            hws.append((h.shape[1], h.shape[2]))
            # the block's ReLU decision travels in the low mantissa bit of the stored multiplier (include/bcos_hip.h:
            # BCOS_EPI_SCALE_GATE_LSB): the explanation pass reads no separate gate tensor
            if bi + 1 < len(self.blocks):
                k_next = self.blocks[bi + 1].k_first
            else:
                k_next = self.head.k_fwd if self.head is not None else ops.F16X2_MIN_K
            out, t = blk.convs[-1].fwd(h, addend=idn, relu=blk.relu, want_scale=keep, gates=gates, track=need(k_next),
                                       flags=BCOS_EPI_SCALE_GATE_LSB if (keep and blk.relu and not _GATE_TENSOR) else 0)
            ts.append(t)
            tw.append(blk.convs[-1].take_wide())
            if keep:
                gate_t = None
                if _GATE_TENSOR and blk.relu:       # development switch: separate gate tensor instead of the bit
                    pinned = blk.convs[-1].last_gate
                    gate_t = pinned if pinned is not None else out
                rec.update(ts=ts, tw=tw, td=td, gated=bool(blk.relu), gate_t=gate_t, hws=hws, pre_pool_hw=pre_pool_hw)
                st["blocks"].append(rec)
            cur = out
            yield
        if self.head_kind in ("attnpool", "attn_unpool"):
            emb = self._attnpool_forward(cur, st) if self.head_kind == "attnpool" else self._attn_unpool_forward(cur, st)
            if keep:
                st["feat_hw"] = (cur.shape[1], cur.shape[2])
            return emb, st
        f, tf = self.head.fwd(cur, relu=False, want_scale=keep, track=False)
        logits = ops.global_avgpool_logits(f, self.logit_temperature, self.logit_bias)
        if keep:
            st.update(tf=tf, feat_hw=(cur.shape[1], cur.shape[2]))
        return logits, st

    @staticmethod
    def _hybrid_forward(blk, inp, rec, track):
        """A grouped / MaxOut block on the nn.Module path: NHWC tensor in, NHWC tensor out.  With `rec` (explanation pass) the
        block runs in explanation mode under autograd and (input leaf, output) are kept for `torch.autograd.grad`."""
        xin = inp.permute(0, 3, 1, 2)               # the module path's logical NCHW view of the channels-last tensor
        if rec is None:
            with torch.no_grad():
                y = blk.module(xin)
        else:
            xin = xin.detach().requires_grad_(True)
            mods = [m for m in blk.module.modules() if hasattr(m, "set_explanation_mode")]
            prev = [m.is_in_explanation_mode for m in mods]
            for m in mods:
                m.set_explanation_mode(True)
            try:
                with torch.enable_grad():
                    y = blk.module(xin)
            finally:
                for m, was in zip(mods, prev):
                    m.set_explanation_mode(was)
            rec.update(hybrid=(xin, y))
        out = y.detach().permute(0, 2, 3, 1)
        out = out if out.is_contiguous() else out.contiguous()
        if track:
            ops.ensure_absmax(out)
        return out

    def _attnpool_forward(self, feat, st=None):
        """BcosAttentionPool2d.forward, pooled mode (bcosattnpool.py:33-59): tokens = [mean; HW positions], plain q/k/v
        projections (no bias, no positional embedding), 32-head softmax attention of the mean token, plain c_proj."""
        N, H, W, C = feat.shape
        T = H * W + 1
        tokens = torch.empty((N, T, C), device=feat.device, dtype=torch.float32)
        tokens[:, 1:] = feat.view(N, H * W, C)
        tokens[:, 0] = ops.global_avgpool_logits(feat, None, None)
        flat = ops.ensure_absmax(tokens.view(N * T, C))
        qkv = torch.empty((N, T, 3 * C), device=feat.device, dtype=torch.float32)
        for i, key in enumerate("qkv"):
            ops.tapconv(flat, self.ap_w[key], _linear_geom(N * T, C, C, out_pitch=3 * C), out=qkv.view(N * T, 3 * C)[:, i * C:])
        out, stats = ops.attention_fwd(qkv, self.ap_heads, (C // self.ap_heads) ** -0.5, want_stats=st is not None)
        if st is not None:
            st["ap_qkv"], st["ap_stats"] = qkv, stats
        pooled = out[:, 0, :].contiguous()
        emb = ops.matmul_nt(pooled, self.ap_w["c"])
        if self.logit_temperature is not None:
            emb = emb / self.logit_temperature
        if self.logit_bias is not None:
            emb = emb + self.logit_bias
        return emb

    def _attn_unpool_forward(self, feat, st=None):
        """BcosAttentionPool2d.forward, `attn_unpool` branch (bcosattnpool.py:23-32): every location of the feature map goes
        through v_proj (plain, with bias) and the B-cos c_proj and is L2-normalised.  Returns [(HW), N, D'] like the
        reference (batch is dim 1: a strided view of the [N, HW, D'] rows the kernels produce)."""
        N, H, W, C = feat.shape
        rows = N * H * W
        D = self.ap_w["c"].shape[0]
        v = torch.empty((rows, C), device=feat.device, dtype=torch.float32)
        ops.tapconv(ops.ensure_absmax(feat.view(rows, C)), self.ap_w["v"], _linear_geom(rows, C, C), out=v, bias=self.ap_bias["v"])
        y, t, _ = ops.linear_fwd(ops.ensure_absmax(v), self.ap_w["c"], bias=self.ap_bias["c"], b=self.ap_cb, want_scale=st is not None)
        u, inv = ops.rows_normalize(y, want_y=True, want_inv=st is not None)
        if st is not None:
            st["ap_tc"], st["ap_inv"] = t, inv
        if self.logit_temperature is not None or self.logit_bias is not None:
            raise BcosHipError("engine: a LogitLayer behind an attn_unpool head is not supported")
        return u.view(N, H * W, D).permute(1, 0, 2)

    def _attn_unpool_backward(self, st, g_out, consume):
        """Cotangent g_out [(HW), N, D'] of the un-pooled head output -> gradient w.r.t. v_proj's output at the HW positions
        (explanation mode: the L2 norm and the B-cos scale of c_proj are constants, bcosattnpool.py:29-31)."""
        t, inv = st["ap_tc"], st["ap_inv"]
        if consume:
            st["ap_tc"] = st["ap_inv"] = None
        H, W = st["feat_hw"]
        HW, N, D = g_out.shape
        g_rows = g_out.permute(1, 0, 2).contiguous().view(N * HW, D)
        zero = torch.zeros((N * HW,), device=g_rows.device, dtype=torch.float32)
        g_y = ops.cosine_grad(g_rows, g_rows, zero, inv)                    # rows scaled by 1 / ||y||  (l = 0: inv * w)
        g_lin = ops.mul(g_y, t) if t is not None else g_y                   # through the B-cos scale of c_proj
        g_v = self.ap_cconv.run(ops.ensure_absmax(g_lin.view(N, H, W, D)), H, W)
        return ops.ensure_absmax(g_v)

    def _attnpool_backward(self, st, cls, consume, g_emb=None):
        """d emb[n, cls[n]] / d v_lin at the HW positions, with q and k detached (bcosattnpool.py:37-39): back through the
        plain c_proj, through the attention of the mean token (gradient w.r.t. v only), and the mean token's own v row
        spread over the positions it averages (tokens[0] = mean of the HW positions, :35).  The remaining step, through
        v_proj, is the 1x1-convolution input gradient the consumer runs."""
        qkv, stats = st["ap_qkv"], st["ap_stats"]
        if consume:
            st["ap_qkv"] = st["ap_stats"] = None
        N, T, C3 = qkv.shape
        C = C3 // 3
        H, W = st["feat_hw"]
        D = self.ap_cT.shape[1]
        if g_emb is None:
            g_emb = torch.zeros((N, D), device=qkv.device, dtype=torch.float32)
            g_emb.scatter_(1, cls.view(-1, 1), 1.0 if self.logit_temperature is None else 1.0 / float(self.logit_temperature))
        else:                    # an arbitrary cotangent of the embedding (e.g. the zero-shot cosine logit, clip_head.zeroshot_attribution)
            g_emb = g_emb.to(device=qkv.device, dtype=torch.float32).contiguous()
            if self.logit_temperature is not None:
                g_emb = g_emb / float(self.logit_temperature)
        g_out = torch.zeros((N, T, C), device=qkv.device, dtype=torch.float32)
        g_out[:, 0] = ops.matmul_nt(g_emb, self.ap_cT)                    # g_emb @ W_c
        g_v = ops.attention_bwd_v(qkv, stats, g_out, self.ap_heads, (C // self.ap_heads) ** -0.5)
        g_lin = g_v[:, 1:] + g_v[:, :1] / float(H * W)                   # positions + their share of the mean token
        return ops.ensure_absmax(g_lin.reshape(N, H, W, C).contiguous())

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        S = self._n_subbatches(x) if self.head_kind != "attn_unpool" else 1
        if S > 1:                                  # sub-batches on side streams (see _SUBBATCH_STREAMS)
            streams, arenas = self._side_for(x, S)
            cur = torch.cuda.current_stream()
            x = x.detach()
            x = x if x.is_contiguous() else x.contiguous()
            N = x.shape[0]
            for i in range(S):
                streams[i].wait_stream(cur)
            parts = [r[0] for r in _interleave([self._run_forward_gen(x[(N * i) // S:(N * (i + 1)) // S], keep=False) for i in range(S)],
                                               streams, arenas, x.device)]
            for st in streams[:S]:
                cur.wait_stream(st)
            for t in parts:
                t.record_stream(cur)
            return torch.cat(parts)
        with ops.absmax_arena(self._absmax_arena, x.device):
            return self._run_forward(x, keep=False)[0]

    @torch.no_grad()
    def explain(self, x: torch.Tensor, targets: Optional[torch.Tensor] = None, want_weights: bool = True,
                gates=None, cotangent=None) -> Dict[str, torch.Tensor]:
        """Forward in explanation mode + input-gradient pass of the explained logit of every image
        (batched bcos/common.py:163-181).  `targets` [N] int64 selects the logits (default: arg-max).
        `gates`: optional list of NHWC 0/1 tensors, one per ReLU in execution order, that REPLACE the v > 0
        decisions (replay of gates recorded elsewhere; used by the gate-pinned parity test).
        `cotangent` (attention-pool heads): a callable that receives the head output -- the embeddings [N, D], or [(HW), N, D']
        for an `attn_unpool` head -- and returns d(scalar to explain) / d(head output) of the same shape; the input-gradient
        pass then starts from it instead of from a one-hot coordinate (the zero-shot text logit of
        interpretability/analyses/text_localisation.py:68-126: bcos_hip.clip_head.zeroshot_attribution)."""
        if targets is not None and self.head_kind == "gap_fc":
            targets = ops.check_targets(targets, self.head.cout)       # IndexError like the reference's out[0, idx]; negative indices wrap
        S = self.n_streams(x, cotangent=cotangent)
        if S > 1:
            return self._explain_subbatches(x, targets, want_weights, S, gates)
        return self._explain_one(x, targets, want_weights, gates, cotangent, self._absmax_arena)

    def n_streams(self, x, cotangent=None) -> int:
        """On how many sub-batch streams does explain(x, ...) run?  `subbatch_streams` (default 2) for a batch of at least that many
        times _SUBBATCH_MIN images on a device -- replayed `gates` included: they are cut along the batch like the images -- and ONE
        in exactly three cases: a `cotangent` callable (it is handed the head output of the whole batch at once), the `attn_unpool`
        head (its output is token-major), and a pass that is being captured into a hipGraph.  Results
        do not depend on the answer: an image's bits are a function of the image alone."""
        if cotangent is not None or self.head_kind == "attn_unpool":
            return 1
        return self._n_subbatches(x)

    def _side_for(self, x, S):
        """(streams, arenas) of the sub-batch passes on x's device, created on first use.  Everything the passes cache lazily --
        refreshed layer plans, the attention-pool copies, the mean / std constants -- is brought up to date HERE, on the caller's
        stream, which every side stream then waits for: no sub-batch reads a cache another one is still producing."""
        self._ensure_fresh()
        self._consts(x.device)
        key = str(x.device)
        if self._side is None:
            self._side = {}
        if key not in self._side or len(self._side[key][0]) < S:
            self._side[key] = ([torch.cuda.Stream(device=x.device) for _ in range(S)], [ops.AbsmaxArena() for _ in range(S)])
        return self._side[key]

    def _n_subbatches(self, x) -> int:
        S = min(int(self.subbatch_streams), x.shape[0] // _SUBBATCH_MIN)      # (fewer streams for batches under S x _SUBBATCH_MIN images)
        if S <= 1 or not x.is_cuda or torch.cuda.is_current_stream_capturing():
            return 1
        return S

    def _explain_subbatches(self, x, targets, want_weights, S, gates=None):
        """explain() of S contiguous sub-batches on S side streams, written into ONE set of output tensors (see
        _SUBBATCH_STREAMS).  The side streams start behind the caller's stream and the caller's stream waits for them."""
        streams, arenas = self._side_for(x, S)
        cur = torch.cuda.current_stream()
        N, _, H, W = x.shape
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()
        wts = torch.empty((N, 6, H, W), device=x.device, dtype=torch.float32) if want_weights else None
        contrib = torch.empty((N, H, W), device=x.device, dtype=torch.float32)
        bounds = [(N * i) // S for i in range(S + 1)]
        tg = None if targets is None else targets.to(device=x.device, dtype=torch.int64).contiguous()
        gens = []
        for i in range(S):
            lo, hi = bounds[i], bounds[i + 1]
            streams[i].wait_stream(cur)
            sub_gates = None if gates is None else [gt[lo:hi] for gt in gates]       # (replayed ReLU decisions: [N, ...] like the activations)
            gens.append(self._explain_one_gen(x[lo:hi], None if tg is None else tg[lo:hi], want_weights, sub_gates, None,
                                              outs=(wts[lo:hi] if want_weights else None, contrib[lo:hi])))
        parts = _interleave(gens, streams, arenas, x.device)
        for st in streams[:S]:
            cur.wait_stream(st)
        res = {}
        for k in ("logits", "prediction", "explained_class_idx"):
            ts = [p[k] for p in parts]
            for t in ts:
                t.record_stream(cur)             # allocated in a side stream's pool, read by the cat on the caller's stream
            res[k] = torch.cat(ts)
        res.update(dynamic_linear_weights=wts, contribution_map=contrib)
        return res

    def _explain_one(self, x, targets, want_weights, gates, cotangent, arena, outs=None):
        with ops.absmax_arena(arena, x.device):
            return _drive(self._explain_one_gen(x, targets, want_weights, gates, cotangent, outs))

    def _explain_one_gen(self, x, targets, want_weights, gates, cotangent, outs=None):
        """forward + input-gradient pass of one (sub-)batch as a generator (yields behind every block of either pass); the caller owns
        the stream and the maxima arena the launches are issued under"""
        logits, st = yield from self._run_forward_gen(x, keep=True, gates=gates)
        if cotangent is not None:
            if self.head_kind not in ("attnpool", "attn_unpool"):
                raise BcosHipError("engine: `cotangent` needs an attention-pool head (CLIP image encoder)")
            g_head = cotangent(logits)
            if tuple(g_head.shape) != tuple(logits.shape):
                raise ValueError(f"cotangent: expected shape {tuple(logits.shape)}, got {tuple(g_head.shape)}")
            wts, contrib = yield from self._backward_gen(x, st, None, want_weights, consume=True, g_head=g_head, outs=outs)
            return dict(logits=logits, embedding=logits, dynamic_linear_weights=wts, contribution_map=contrib)
        if self.head_kind == "attn_unpool":
            raise BcosHipError("engine: an attn_unpool head has no class logits of its own: pass `cotangent` "
                               "(bcos_hip.clip_head.zeroshot_attribution builds it from the text embeddings)")
        pred, _ = ops.argmax_rows(logits)
        if targets is not None and self.head_kind != "gap_fc":
            targets = ops.check_targets(targets, logits.shape[1])      # (attention-pool head: the embedding width; gap_fc heads were checked in explain())
        cls = pred if targets is None else targets.to(device=logits.device, dtype=torch.int64).contiguous()
        wts, contrib = yield from self._backward_gen(x, st, cls, want_weights, consume=True, outs=outs)
        return dict(logits=logits, prediction=pred, explained_class_idx=cls, dynamic_linear_weights=wts,
                    contribution_map=contrib)

    @torch.no_grad()
    def explain_targets(self, x: torch.Tensor, targets: torch.Tensor, want_weights: bool = False) -> Dict[str, torch.Tensor]:
        """ONE forward in explanation mode, then one input-gradient pass per target column: `targets` [N, T] (or [T],
        shared by all images).  The reference's `attribute_selection` (interpretability/explanation_methods/utils.py:84-99,
        bcos/common.py:319-344) re-runs the forward for every target; the dynamic weights of the forward do not depend on
        the explained logit, so they are kept and only the backward is repeated.
        Returns logits [N,K], contribution_maps [N,T,H,W] and, if requested, dynamic_linear_weights [N,T,6,H,W]."""
        if self.head_kind == "gap_fc":
            targets = ops.check_targets(targets, self.head.cout)
        with ops.absmax_arena(self._absmax_arena, x.device):
            logits, st = self._run_forward(x, keep=True)
            targets = ops.check_targets(targets, logits.shape[1]) if self.head_kind != "gap_fc" else targets
            tg = targets.to(device=logits.device, dtype=torch.int64)
            if tg.dim() == 1:
                tg = tg.view(1, -1).expand(x.shape[0], -1)
            T = tg.shape[1]
            maps = torch.empty((x.shape[0], T, st["H"], st["W"]), device=x.device, dtype=torch.float32)
            wall = torch.empty((x.shape[0], T, 6, st["H"], st["W"]), device=x.device, dtype=torch.float32) if want_weights else None
            for k in range(T):
                wts, contrib = self._backward(x, st, tg[:, k].contiguous(), want_weights, consume=(k == T - 1))
                maps[:, k] = contrib
                if want_weights:
                    wall[:, k] = wts
        out = dict(logits=logits, contribution_maps=maps)
        if want_weights:
            out["dynamic_linear_weights"] = wall
        return out

    def _backward(self, x, st, cls, want_weights: bool, consume: bool, g_head=None, outs=None):
        return _drive(self._backward_gen(x, st, cls, want_weights, consume, g_head, outs))

    def _backward_gen(self, x, st, cls, want_weights: bool, consume: bool, g_head=None, outs=None):
        """Input-gradient pass of logit[cls[n]] for every image n over the state `st` of a kept forward; `consume` frees
        each saved multiplier as soon as it has been used (last / only pass over this state)."""
        # A "consumer" owns the g_lin tensors of the layers that read some activation X and can therefore
        # finish d logit / d X; its epilogue applies the multipliers of the block that PRODUCED X.
        if self.head_kind == "attnpool":
            consumer = _Consumer(self.ap_vconv, self._attnpool_backward(st, cls, consume, g_emb=g_head), None, None, 0)
        elif self.head_kind == "attn_unpool":
            consumer = _Consumer(self.ap_vconv, self._attn_unpool_backward(st, g_head, consume), None, None, 0)
        else:
            # d logit[cls] / d (head lin): one-hot * 1/(T*HW) * head scale
            hc = self.head
            if (_HEAD_RANK1 and hc.k == (1, 1) and hc.stride == (1, 1) and hc.padding == (0, 0) and getattr(hc, "groups", 1) == 1
                    and int(getattr(hc.module, "max_out", 1)) == 1 and hc.w_fwd.shape[-1] % 4 == 0 and hc.w_fwd.shape[-1] == hc.cin):
                # ... carried through the head's 1 x 1 convolution at once: the gradient is rank one per image (one class column of
                # the head's scale x one weight row), a streaming launch instead of the [N, 7, 7, 1000] one-hot tensor, a pass for its
                # row maxima and a K = 1000 contraction over 999 zero columns
                consumer = _HeadRank1Consumer(cls, st["tf"], hc.w_fwd.view(hc.w_fwd.shape[0], -1), self.logit_temperature)
            else:
                g_head = ops.ensure_absmax(ops.head_onehot_grad(cls, st["tf"], self.logit_temperature))
                consumer = _Consumer(self.head, g_head, None, None, 0)
            if consume:
                st["tf"] = None
        nb = len(self.blocks)
        for bi in range(nb - 1, -1, -1):
            blk, rec = self.blocks[bi], st["blocks"][bi]
            H, W = st["blocks"][bi + 1]["in_hw"] if bi + 1 < nb else st["feat_hw"]
            if blk.hybrid:
                # raw d logit / d out_b from the layers that read it, then the module path's own backward through the block
                v, _ = consumer.run(H, W, t_main=None, td=None, gated=False, track=False)
                xin, y = rec["hybrid"]
                gx = torch.autograd.grad(y, xin, v.permute(0, 3, 1, 2), retain_graph=not consume)[0]
                if consume:
                    rec["hybrid"] = None
                gx = gx.permute(0, 2, 3, 1)
                consumer = _RawConsumer(gx if gx.is_contiguous() else gx.contiguous())
                yield
                continue
            # v = d logit / d out_b;  G_main = v * t_last (bn scale, ReLU gate and s of the block's last conv),
            # G_sc = v * gate(out_b) [* t_d]  for the shortcut
            with ops.image_range_reader(blk.convs[-1].reads_image_range):
                G_main, G_sc = consumer.run(H, W, t_main=rec["ts"][-1], td=rec["td"], gated=rec["gated"], gate_t=rec["gate_t"],
                                            track=blk.convs[-1].k_dgrad >= ops.F16X2_MIN_K,
                                            track2=blk.shortcut is not None and blk.shortcut.k_dgrad >= ops.F16X2_MIN_K)
            if consume:
                rec["ts"][-1] = rec["td"] = rec["gate_t"] = None
            gl = G_main
            convs = blk.convs
            for ci in range(len(convs) - 1, 0, -1):
                h, w = rec["hws"][ci]
                gl = convs[ci].expand(gl, rec["tw"][ci])          # (MaxOut layers: to the contraction's width; anything else: as is)
                if consume:
                    rec["tw"][ci] = None
                if blk.pool and ci == len(convs) - 1:
                    # anti-aliasing pool between conv(ci-1) and conv(ci): gradient w.r.t. the pooled tensor, then
                    # the pool's input gradient times the scale of conv(ci-1)
                    gp = convs[ci].dgrad.run(gl, h, w, track_absmax=False)
                    ph, pw = rec["pre_pool_hw"]
                    gl = ops.avgpool2d_bwd(gp, ph, pw, blk.pool, blk.pool, 0, mul=rec["ts"][ci - 1],
                                           want_absmax=convs[ci - 1].k_dgrad >= ops.F16X2_MIN_K)
                else:
                    with ops.image_range_reader(convs[ci - 1].reads_image_range):
                        gl = convs[ci].dgrad.run(gl, h, w, track_absmax=convs[ci - 1].k_dgrad >= ops.F16X2_MIN_K,
                                                 **_mul_kwargs(rec["ts"][ci - 1]))
                if consume:
                    rec["ts"][ci - 1] = None
            gl = convs[0].expand(gl, rec["tw"][0])
            if consume:
                rec["tw"][0] = None
            consumer = _Consumer(convs[0], gl, blk.shortcut, G_sc, blk.shortcut_pool)
            yield
        # block 0 reads the stem pool output: raw gradient, pool backward (* t of the last stem conv), then the stem
        H0, W0 = st["blocks"][0]["in_hw"]
        g_pool, _ = consumer.run(H0, W0, t_main=None, td=None, gated=False, track=False)
        k, s, p = self.pool
        a_h, a_w = st["a0_hw"]
        ts = st["stem_ts"]
        gl = ops.avgpool2d_bwd(g_pool, a_h, a_w, k, s, p, mul=ts[-1], want_absmax=self.stem[-1][0].k_dgrad >= ops.F16X2_MIN_K)
        if consume:
            ts[-1] = None
        for si in range(len(self.stem) - 1, 0, -1):
            h, w = st["stem_hws"][si]
            with ops.image_range_reader(self.stem[si - 1][0].reads_image_range):
                gl = self.stem[si][0].dgrad.run(gl, h, w, mul=ts[si - 1],
                                                track_absmax=self.stem[si - 1][0].k_dgrad >= ops.F16X2_MIN_K)
            if consume:
                ts[si - 1] = None
        gxn = torch.empty((x.shape[0], st["H"], st["W"], 8), device=x.device, dtype=torch.float32)
        self.stem[0][0].dgrad.run(gl, st["H"], st["W"], out=gxn)     # channels 0..5 of the padded buffer
        _, std = self._consts(x.device)
        wts, contrib = ops.finalize_explanation(gxn, st["x"], std, add_inverse=st["add_inverse"],
                                                want_weights=want_weights, want_contrib=True,
                                                weights_out=outs[0] if outs is not None else None,
                                                contrib_out=outs[1] if outs is not None else None)
        return wts, contrib


class _HeadConv:
    """What a _Consumer needs from the layer that reads the last feature map: its input-gradient plan and width."""

    def __init__(self, dgrad, cin):
        self.dgrad, self.cin = dgrad, cin


class _Consumer:
    """The layers reading one activation X: a main conv (with its g_lin) and optionally a shortcut -- either a
    conv (g_sc = its g_lin; preceded by an AvgPool2d(sc_pool) in CLIP's blocks) or the identity (g_sc = gradient added
    as is)."""

    def __init__(self, conv, g_main, shortcut_conv, g_sc, sc_pool):
        self.conv, self.g_main, self.shortcut_conv, self.g_sc, self.sc_pool = conv, g_main, shortcut_conv, g_sc, sc_pool

    def run(self, H, W, t_main, td, gated, gate_t=None, track=None, track2=None):
        """-> (v * t_main, v * gate [* td]) with v = d logit / d X; both v when t_main is None.  `gated`: the block that
        produced X ends in a ReLU, whose decision is the low mantissa bit of t_main."""
        g = self.g_main
        kw, out2 = {}, None
        if t_main is not None:
            out2 = torch.empty((g.shape[0], H, W, self.conv.cin), device=g.device, dtype=torch.float32)
            kw = dict(mul=t_main, out2=out2, mul2=td, gate2=gate_t,
                      flags=BCOS_EPI_GATE2_FROM_MUL if (gated and gate_t is None) else 0)
        if self.shortcut_conv is not None:
            if self.sc_pool:
                pooled = self.shortcut_conv.dgrad.run(self.g_sc, H // self.sc_pool, W // self.sc_pool, track_absmax=False)
                addend = ops.avgpool2d_bwd(pooled, H, W, self.sc_pool, self.sc_pool, 0)
            elif _SUB_ADDEND and self.shortcut_conv.dgrad.subsampled and not self.conv.dgrad.has_empty and self.conv.dgrad.groups == 1:
                # 1x1 / stride-s shortcut: its input gradient is zero off the s-grid -- hand the grid pixels alone to the main
                # branch's launch (bcos_epilogue.addend_sub) instead of scattering them into a zero-filled full-size tensor
                # (ResNet-50: 1.44 GB of zero fill per step and as much again read back as an addend)
                addend = self.shortcut_conv.dgrad.run_compact(self.g_sc, track_absmax=False)
                kw["addend_sub"] = self.shortcut_conv.dgrad.subsampled
            else:
                addend = self.shortcut_conv.dgrad.run(self.g_sc, H, W, track_absmax=False)
        else:
            addend = self.g_sc
        out = self.conv.dgrad.run(g, H, W, addend=addend, track_absmax=track, track_absmax2=track2, **kw)
        return out, (out2 if out2 is not None else out)


class _HeadRank1Consumer:
    """The GAP + fc head as the reader of the last feature map: d logit[cls] / d X = coef * scale[n, hw, cls_n] * W[cls_n, :] per
    pixel (ops.head_rank1_grad), with the multipliers / gates of the producing block applied by the same launch."""

    def __init__(self, cls, tf, w, temperature):
        self.cls, self.tf, self.w, self.temperature = cls, tf, w, temperature

    def run(self, H, W, t_main, td, gated, gate_t=None, track=None, track2=None):
        N, K, D = self.tf.shape[0], self.tf.shape[-1], self.w.shape[1]
        tf = self.tf.view(N, H * W, K)
        if t_main is None:
            out, _ = ops.head_rank1_grad(self.cls, tf, self.w, self.temperature, want_absmax=bool(track))
            out = self._nhwc(out, N, H, W, D)
            return out, out
        flat = lambda t: None if t is None else (t if t.is_contiguous() else t.contiguous()).view(N * H * W, D)      # noqa: E731
        out, out2 = ops.head_rank1_grad(self.cls, tf, self.w, self.temperature, mul=flat(t_main), want_out2=True, mul2=flat(td), gate2=flat(gate_t),
                                        gate2_from_mul=bool(gated and gate_t is None), want_absmax=bool(track), want_absmax2=bool(track2))
        return self._nhwc(out, N, H, W, D), self._nhwc(out2, N, H, W, D)

    @staticmethod
    def _nhwc(t, N, H, W, D):
        v = t.view(N, H, W, D)
        am = ops.absmax_of(t)
        if am is not None:
            ops._attach_absmax(v, am)
        return v


class _RawConsumer:
    """d logit / d X already computed (X feeds a hybrid block): applies the multipliers of the block that produced X with plain
    elementwise launches."""

    def __init__(self, g):
        self.g = g

    def run(self, H, W, t_main, td, gated, gate_t=None, track=None, track2=None):
        g = self.g
        if t_main is None:
            return g, g
        out = g * t_main
        out2 = g
        if gated:       # the ReLU decision of the producing block: its gate tensor, or the low mantissa bit of its multiplier
            gate = (gate_t > 0) if gate_t is not None else (t_main.view(torch.int32) & 1).bool()
            out2 = g * gate
        if td is not None:
            out2 = out2 * td
        if track:
            ops.ensure_absmax(out)
        if track2:
            ops.ensure_absmax(out2)
        return out, out2


class CapturedPass:
    """One forward (+ explanation) pass of an engine for a FIXED input shape, recorded once into a hipGraph and replayed:
    the ~130 launches of a ResNet-50 step are then submitted as one graph instead of one by one.  Eager launches leave
    ~10 us of idle time between dependent kernels (1.3 ms per step); on ROCm 7.2 the graph replay does not close that
    gap -- measured 38.4 ms vs 37.9 ms per step eager at ResNet-50 batch 256, 4.00 vs 4.04 ms at ResNet-18 batch 8 --
    so bench.py keeps it opt-in (--graph).  It does remove the host-side launch work.
    Outputs live in static buffers that every replay overwrites; call the object with a new input batch of the same
    shape.  Weights must not be re-laid out between capture and replay (refresh() -> capture again)."""

    def __init__(self, eng, x: torch.Tensor, explain: bool = True, want_weights: bool = True, warmup: int = 2):
        ops.require_device(x, "bcos_hip.engine.CapturedPass")
        self.static_x = x.detach().clone()
        fn = (lambda: eng.explain(self.static_x, want_weights=want_weights)) if explain else \
             (lambda: dict(logits=eng.forward(self.static_x)))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm-up off the capture: allocator pool, sticky kernel attributes,
            for _ in range(max(1, warmup)):         # pre-split weight images
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self, x: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if x is not None and x.data_ptr() != self.static_x.data_ptr():
            self.static_x.copy_(x)
        self.graph.replay()
        return self.out


def _linear_geom(rows, cin, cout, out_pitch=0):
    return dict(N=1, H=1, W=rows, C=cin, P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1, TH=1, TW=1,
                OH=1, OW=rows, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=cout, out_pitch=out_pitch)


def attach(net) -> ResNetEngine:
    """Compile `net` (a BcosifyNetwork around a torchvision-style ResNet) and make `net(x)` (under no_grad) and
    `net.explain_batch(x)` use the fused plan.  Call `net._bcos_engine.refresh()` after changing parameters."""
    eng = ResNetEngine(net)
    object.__setattr__(net, "_bcos_engine", eng)
    return eng


def detach(net):
    if hasattr(net, "_bcos_engine"):
        object.__delattr__(net, "_bcos_engine")
