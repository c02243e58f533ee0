"""autograd bridge between the nn.Module API and the HIP kernels (bcos_hip.ops).

Forward = ONE fused launch per B-cos layer (contraction + patch norm + |cos|^(B-1) scaling;
reference bcosconv2d.py:153-194 issues 9 ATen launches for the same thing).
Backward:
  * explanation mode (`detach=True`): the dynamic scale is a constant, so d out / d x = conv_transpose(g * s, W) -- the
    "dynamic linear weights" W(x) of bcos/common.py:177-181 -- and d out / d W = wgrad(g * s, x);
  * training mode (grouped layers per group): the scale is differentiated too (bcosconv2d.py:176-194 without
    .detach()):  gx = dgrad(g * dy/dlin, W) + x (.) PatchSum^T(dL/dnorm / norm), gW = wgrad(g * dy/dlin, x),
    gbias = sum_pixels g * dy/dlin (csrc/bcos_train.hip); MaxOut routes g * dy/dlin to the winning filter of each unit
    (bcos_maxout_scatter); a learnable exponent (`b` an nn.Parameter, trainer.py:451-463) receives
    sum g y ln(|cos| + 1e-6) times d B_eff / d b; native layers differentiate through their unit-norm projection (UnitNormFn).
Grouped MaxOut layers train per group too; only units that would straddle two groups (out_channels % groups != 0) raise.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from bcos_hip import ops
from bcos_hip.lib import (BCOS_CONV_EPS, BCOS_EPI_FORCE_POW, BCOS_EPI_NORM_ONLY, BCOS_EPI_UNIT_NORM_W, BCOS_LINEAR_EPS, BCOS_NONE,
                          BcosHipError)


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def require_hip(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise BcosHipError(
            f"{who}: input is on {t.device}; this package only implements the B-cos hot path as HIP kernels "
            "for gfx950 and has no CPU fallback (move the model and input to 'cuda').")
    if t.dtype != torch.float32:
        raise BcosHipError(f"{who}: the hot path is fp32 end to end (got {t.dtype})")


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """logical [N,C,H,W] (any strides) -> dense [N,H,W,C]; free when x is channels_last."""
    return x.permute(0, 2, 3, 1).contiguous()


def from_nhwc(y: torch.Tensor) -> torch.Tensor:
    """dense [N,H,W,C] -> logical [N,C,H,W] view with channels_last strides."""
    return y.permute(0, 3, 1, 2)


def empty_cl(n, c, h, w, device):
    """(logical [N,C,H,W] channels_last tensor, its dense [N,H,W,C] view).  autograd Functions must return the
    base tensor, not a view of it (in-place ReLU on a view of a custom Function's output is forbidden)."""
    t = torch.empty((n, c, h, w), device=device, dtype=torch.float32, memory_format=torch.channels_last)
    return t, t.permute(0, 2, 3, 1)


def _pad_last(t: torch.Tensor, mult: int = 4) -> torch.Tensor:
    c = t.shape[-1]
    r = (-c) % mult
    return t if r == 0 else F.pad(t, (0, r))


class UnitNormFn(Function):
    """w -> gain * w / ||w||_2 per filter (NormedConv2d / NormedLinear, bcosconv2d.py:26-35, bcoslinear.py:25-27) with its
    backward (bcos_weight_rownorm_bwd): used in training mode, where the raw weight (and a trainable `scale`) need the
    gradient that arrives at the projected weight.  Eval / explanation use the cached, detached projection instead."""

    @staticmethod
    def forward(ctx, w, scale):
        require_hip(w, "NormedConv2d / NormedLinear")
        w2 = w.detach().contiguous().view(w.shape[0], -1)
        gain = scale.detach().reshape(-1).contiguous() if scale is not None else None
        ctx.save_for_backward(w2, *([gain] if gain is not None else []))
        ctx.has_gain = gain is not None
        ctx.scale_shape = tuple(scale.shape) if scale is not None else None
        return ops.weight_rownorm_scale(w2, gain).view_as(w)

    @staticmethod
    def backward(ctx, g):
        w2 = ctx.saved_tensors[0]
        gain = ctx.saved_tensors[1] if ctx.has_gain else None
        need_w, need_s = ctx.needs_input_grad[0], ctx.has_gain and ctx.needs_input_grad[1]
        gw, gg = ops.weight_rownorm_bwd(w2, g.contiguous().view(w2.shape[0], -1), gain, want_gw=need_w, want_ggain=need_s)
        return (gw.view_as(g) if gw is not None else None), (gg.view(ctx.scale_shape) if gg is not None else None)


FOLD_UNIT_NORM = True      # False: training steps of native layers project their weights with a separate launch again (UnitNormFn)


def _unit(cfg):
    """(epilogue flag, col_scale) of a launch whose unit-norm weight projection is folded into the contraction (FoldedUnitNormFn)"""
    return (BCOS_EPI_UNIT_NORM_W if cfg.get("unit_w") else 0), cfg.get("unit_gain")


class FoldedUnitNormFn(Function):
    """A native B-cos layer (unit-norm filters, bcosconv2d.py:26-35 / bcoslinear.py:25-27) in TRAINING mode, where the
    projection w / ||w|| is recomputed on every call: ONE forward launch on the raw weights -- the contraction gathers ||w_c||
    from the weight rows it stages anyway and scales its accumulator columns (BCOS_EPI_UNIT_NORM_W; the optional trainable
    `scale` rides in col_scale) -- instead of a projection kernel that writes a W-sized tensor which the contraction then reads.
    The backward materialises the projected weights once for the input-gradient plan and chains the weight gradient through
    the projection (bcos_weight_rownorm_bwd) like UnitNormFn did.  `inner` = BcosConv2dFn | BcosLinearFn."""

    @staticmethod
    def forward(ctx, x, w_raw, scale, cfg, b_param, inner):
        gain = scale.detach().reshape(-1).contiguous() if scale is not None else None
        ctx.fold = (w_raw.detach(), gain, tuple(scale.shape) if scale is not None else None, inner)
        return inner.forward(ctx, x, w_raw.detach(), None, dict(cfg, unit_w=True, unit_gain=gain), b_param)

    @staticmethod
    def backward(ctx, gy):
        w_raw, gain, scale_shape, inner = ctx.fold
        w2 = w_raw.contiguous().view(w_raw.shape[0], -1)
        ctx.w_eff = ops.weight_rownorm_scale(w2, gain).view_as(w_raw)              # for the input-gradient plan
        gx, g_eff, _, _, gbp = inner.backward(ctx, gy)
        gw = gg = None
        if g_eff is not None:
            gw, gg = ops.weight_rownorm_bwd(w2, g_eff.contiguous().view(w2.shape[0], -1), gain, want_gw=True,
                                            want_ggain=gain is not None and ctx.needs_input_grad[2])
            gw = gw.view_as(w_raw)
            gg = gg.view(scale_shape) if gg is not None else None
        return gx, gw, gg, None, gbp, None


def folds_projection(lin, detach) -> bool:
    """training step of a native layer whose raw weight is trained: FoldedUnitNormFn applies"""
    return bool(FOLD_UNIT_NORM and lin.bias is None and not detach and lin.weight.requires_grad
                and wants_projection_grad(lin, lin.weight, getattr(lin, "scale", None)))


def wants_projection_grad(module, *params) -> bool:
    """training-mode call of a unit-norm layer whose raw weight (or scale) is being trained"""
    return bool(module.training and torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params))


class WeightCache:
    """Kernel-layout copies of a module's weight ([Cout,kh,kw,Cin_pad], dgrad sub-kernels), rebuilt when
    the parameter changes (data_ptr / in-place version)."""

    def __init__(self):
        self._key = None
        self._fwd = None
        self._dgrad = {}
        self._transient = False

    def _sync(self, w, w_eff):
        # keyed on the parameter AND on the effective weight derived from it: NormedConv2d's unit-norm projection /
        # set_scale / toggle_weight_norm change w_eff while the parameter stays the same (bcosconv2d.py:26-41)
        key = (w.data_ptr(), w._version, tuple(w.shape), str(w.device), w_eff.data_ptr(), w_eff._version)
        if key != self._key:
            self._key, self._fwd, self._dgrad, self._transient = key, None, {}, False

    def fwd(self, w_eff, src):
        """w_eff: effective OIHW (or [O,I]) weight actually used (after unit-norm projection if any)."""
        self._sync(src, w_eff)
        if self._fwd is None:
            # a weight under autograd is about to change: its copies live for one step on this stream and are not published to other
            # streams (one device synchronisation per copy and image -- a training step of the per-layer path made ~150 of them)
            self._transient = bool(torch.is_grad_enabled() and src.requires_grad)
            w4 = w_eff if w_eff.dim() == 4 else w_eff[:, :, None, None]
            self._fwd = ops.mark_static(_pad_last(w4.detach().permute(0, 2, 3, 1)).contiguous(), self._transient)
            if not self._transient:
                ops.publish_cached(self._fwd)      # (made on this stream, read by launches on any stream later)
            else:
                ops.note_unpublished(self._fwd)    # (completed by the first reader outside a training pass, should this version live that long)
        elif not (torch.is_grad_enabled() and src.requires_grad):
            ops.publish_pending()
        return self._fwd

    def dgrad(self, w_eff, src, stride, padding, dilation, groups):
        self._sync(src, w_eff)
        k = (stride, padding, dilation, groups)
        if k not in self._dgrad:
            w4 = w_eff.detach() if w_eff.dim() == 4 else w_eff.detach()[:, :, None, None]
            cout = w4.shape[0]
            plans = []
            for g in range(groups):
                wg = w4[g * (cout // groups):(g + 1) * (cout // groups)]
                # pad Cout (the dgrad K dimension) to a multiple of 4 with zero filters
                r = (-wg.shape[0]) % 4
                if r:
                    wg = torch.cat([wg, wg.new_zeros((r,) + tuple(wg.shape[1:]))], 0)
                plans.append(ops.DgradPlan(wg, stride, padding, dilation, transient=self._transient))
            if not self._transient:
                ops.publish_cached(w4)
            else:
                ops.note_unpublished(w4)
            self._dgrad[k] = plans
        return self._dgrad[k]


def _b_gradient_applies(cfg, b: float) -> bool:
    """Does the output depend on the exponent at this value?  The reference's `self.b == 1` (plain linear output) and
    `self.b == 2` (|lin| / norm) branches do not touch B (bcosifyconv2d.py:78-79,91-92); only the general pow form does."""
    return b != 1.0 and (b != 2.0 or bool(cfg.get("force_pow")))


def learnable_b(module):
    """the module's exponent when it is a tensor that wants a gradient (trainer.py:451-463 turns `b` into an nn.Parameter)"""
    b = getattr(module, "b", None)
    return b if isinstance(b, torch.Tensor) and b.requires_grad and torch.is_grad_enabled() else None


def _scale_bwd_cols(g2, y2, s2, norm, mode, cfg, want_bgrad):
    """ops.train_scale_bwd over [rows, C] operands of ANY width: the kernel moves float4, so ragged widths are padded to a multiple
    of four columns (g = y = 0, s = 1 contribute nothing to the row sums or to the exponent gradient) and cut back."""
    co = g2.shape[1]
    pad = (-co) % 4
    if pad:
        g2, y2, s2 = F.pad(g2, (0, pad)), F.pad(y2, (0, pad)), F.pad(s2, (0, pad), value=1.0)
    glin, rnorm, bgrad = ops.train_scale_bwd(g2.contiguous(), y2.contiguous(), s2.contiguous(), norm, mode, float(cfg["b"]),
                                             bool(cfg.get("force_pow")), want_bgrad=want_bgrad)
    if pad:
        glin = glin[:, :co].contiguous()
    return glin, rnorm, bgrad


class BcosConv2dFn(Function):
    """y = bcos_conv(x): see module docstring.  cfg keys: stride, padding, dilation, groups, b, max_out,
    detach, cache (WeightCache), w_src (the parameter the cache is keyed on)."""

    @staticmethod
    def forward(ctx, x, w_eff, bias, cfg, b_param=None):
        require_hip(x, "BcosConv2d")
        stride, padding, dilation = cfg["stride"], cfg["padding"], cfg["dilation"]
        groups, b, max_out = cfg["groups"], float(cfg["b"]), cfg["max_out"]
        need_grad = ctx.needs_input_grad[0]
        N, Cin, H, W = x.shape
        Cout_all = w_eff.shape[0]
        kh, kw = w_eff.shape[2], w_eff.shape[3]
        xh = to_nhwc(x)
        wk = cfg["cache"].fwd(w_eff, cfg["w_src"])                     # [Cout_all,kh,kw,Cin_g_pad]
        Ho = ops.conv_out_size(H, kh, stride[0], padding[0], dilation[0])
        Wo = ops.conv_out_size(W, kw, stride[1], padding[1], dilation[1])
        cin_g = Cin // groups
        if groups == 1:
            xh = _pad_last(xh)
        elif cin_g % 4 != 0:
            raise BcosHipError(f"grouped B-cos conv needs in_channels/groups % 4 == 0 (got {cin_g})")
        need_w, need_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        need_bp = b_param is not None and ctx.needs_input_grad[4] and _b_gradient_applies(cfg, b)
        train = (need_grad or need_w or need_b or need_bp) and not cfg["detach"] and b != 1.0     # the scale is differentiated
        if (train or need_w or need_b) and groups != 1 and max_out != 1 and (Cout_all // max_out) % groups:
            raise NotImplementedError(
                "BcosConv2d: weight gradients / training-mode gradients of grouped MaxOut layers need out_channels % groups == 0 "
                "(a unit's filters must not straddle two groups); such layers support explanation-mode input gradients only")
        want_scale = bool((need_grad or need_w or need_b or need_bp) and b != 1.0)
        # MaxOut over 2 or 4 filters is taken inside the contraction's epilogue (one launch, bcosconv2d.py:166-170); other
        # unit sizes, grouped layers and training-mode calls (which need the unit-wide y, s and the winner indices) go through
        # the general path (full-width lin + bcos_maxout_scale)
        mo_fused = max_out in (2, 4) and groups == 1 and Cout_all % 4 == 0 and not train
        if mo_fused:
            return BcosConv2dFn._forward_maxout(ctx, x, xh, wk, w_eff, bias, cfg, (N, Cin, H, W), (kh, kw, Ho, Wo), Cout_all, need_grad)
        fused = max_out == 1
        y_cl, y = empty_cl(N, Cout_all, Ho, Wo, x.device)
        scale = torch.empty((N, Ho, Wo, Cout_all), device=x.device, dtype=torch.float32) if (want_scale and fused) else None
        if train and fused:
            norm = torch.empty((N, Ho, Wo, groups), device=x.device, dtype=torch.float32)
        else:
            norm = None if fused or b == 1.0 else torch.empty((N, Ho, Wo, groups), device=x.device, dtype=torch.float32)
        mode = BCOS_NONE if b == 1.0 else BCOS_CONV_EPS
        cout_g = Cout_all // groups
        # one launch for all groups (bcos_tapconv_geom.groups): group g contracts its channel slice with its filters and writes
        # its output columns / its patch norm
        geom = ops.fwd_geom(N, H, W, wk.shape[3], cout_g, kh, kw, stride[0], stride[1], padding[0], padding[1],
                            dilation[0], dilation[1])
        if groups > 1:
            geom.update(groups=groups, a_pitch=Cin, out_pitch=Cout_all, norm_pitch=groups)
        uflag, ugain = _unit(cfg)
        ops.tapconv(xh, wk, geom, out=y, scale_out=scale, norm_out=norm, bias=bias, bcos_mode=mode, b=b,
                    flags=(0 if fused else BCOS_EPI_NORM_ONLY) | (BCOS_EPI_FORCE_POW if cfg.get("force_pow") else 0) | uflag,
                    col_scale=ugain, **({"track_absmax": False} if groups > 1 else {}))
        argmax = None
        if not fused:
            Cout = Cout_all // max_out
            y_cl, y2 = empty_cl(N, Cout, Ho, Wo, x.device)
            _, scale, argmax = ops.maxout_scale(y.reshape(-1, Cout_all), norm.view(-1, groups) if norm is not None else None,
                                                Cout, max_out, b, groups=groups, want_scale=want_scale,
                                                want_argmax=need_grad or need_w or need_b, out=y2.view(-1, Cout))
            if scale is not None:
                scale = scale.view(N, Ho, Wo, Cout)
        ctx.cfg = cfg
        ctx.in_shape = (N, Cin, H, W)
        ctx.w_eff = w_eff
        ctx.train = bool(train)
        ctx.need_bp = bool(need_bp and train)
        ctx.mo_fused = False
        ctx.geom = (kh, kw, Ho, Wo)
        keep_x = xh if (train or need_w) else None                  # the padded NHWC input: weight gradient / norm term
        keep_y = y_cl if train else None
        keep_n = norm if train else None
        ctx.save_for_backward(*(t for t in (scale, argmax, keep_x, keep_y, keep_n) if t is not None))
        ctx.has = (scale is not None, argmax is not None, keep_x is not None, keep_y is not None, keep_n is not None)
        return y_cl

    @staticmethod
    def _forward_maxout(ctx, x, xh, wk, w_eff, bias, cfg, in_shape, geom, Cout_all, need_grad):
        """one launch: contraction + max over the M filters of each unit + B-cos scale; the stored multiplier keeps the
        contraction's width with the scale at the winning filter and zeros elsewhere (= d out / d lin)."""
        N, Cin, H, W = in_shape
        kh, kw, Ho, Wo = geom
        b, M = float(cfg["b"]), cfg["max_out"]
        stride, padding, dilation = cfg["stride"], cfg["padding"], cfg["dilation"]
        Cout = Cout_all // M
        y_cl, y = empty_cl(N, Cout, Ho, Wo, x.device)
        need_wb = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])
        t_full = torch.empty((N, Ho, Wo, Cout_all), device=x.device, dtype=torch.float32) if (need_grad or need_wb) else None
        gm = ops.fwd_geom(N, H, W, wk.shape[3], Cout_all, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1])
        uflag, ugain = _unit(cfg)
        ops.tapconv(xh, wk, gm, out=y, scale_out=t_full, bias=bias, bcos_mode=BCOS_NONE if b == 1.0 else BCOS_CONV_EPS, b=b,
                    flags=(BCOS_EPI_FORCE_POW if cfg.get("force_pow") else 0) | uflag, col_scale=ugain, max_out=M)
        ctx.cfg, ctx.in_shape, ctx.w_eff, ctx.train, ctx.geom, ctx.need_bp = cfg, in_shape, w_eff, False, geom, False
        keep_x = xh if need_wb else None                # explanation-mode weight gradient (scale held constant)
        ctx.save_for_backward(*(t for t in (t_full, keep_x) if t is not None))
        ctx.has = (t_full is not None, False, keep_x is not None, False, False)
        ctx.mo_fused = True
        return y_cl

    @staticmethod
    def backward(ctx, gy):
        cfg = ctx.cfg
        saved = list(ctx.saved_tensors)
        scale = saved.pop(0) if ctx.has[0] else None
        argmax = saved.pop(0) if ctx.has[1] else None
        xh = saved.pop(0) if ctx.has[2] else None
        y_cl = saved.pop(0) if ctx.has[3] else None
        norm = saved.pop(0) if ctx.has[4] else None
        N, Cin, H, W = ctx.in_shape
        kh, kw, Ho, Wo = ctx.geom
        groups, max_out = cfg["groups"], cfg["max_out"]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g = to_nhwc(gy)
        addend = None
        gbp = None
        if ctx.train and groups > 1:
            # grouped layer: every group has its own patch norms, so the scale derivative and the norm term run per group on
            # dense copies of the group's slices (a rarely used path: clarity over launch count)
            Cout = g.shape[3]
            cout_g, cin_g = Cout // groups, Cin // groups
            yh = to_nhwc(y_cl)
            glin = torch.empty((N, Ho, Wo, Cout), device=g.device, dtype=torch.float32)
            addend_g = [None] * groups
            btot = None
            for gi in range(groups):
                sl = slice(gi * cout_g, (gi + 1) * cout_g)
                gl_g, rn_g, bg = _scale_bwd_cols(g[..., sl].reshape(-1, cout_g), yh[..., sl].reshape(-1, cout_g),
                                                 scale[..., sl].reshape(-1, cout_g), norm[..., gi].reshape(-1).contiguous(),
                                                 BCOS_CONV_EPS, cfg, ctx.need_bp)
                glin[..., sl] = gl_g.view(N, Ho, Wo, cout_g)
                if bg is not None:
                    btot = bg if btot is None else btot + bg
                if need_x:
                    addend_g[gi] = ops.patch_norm_bwd(xh[..., gi * cin_g:(gi + 1) * cin_g].contiguous(), rn_g.view(N, Ho, Wo), cin_g,
                                                      (kh, kw), cfg["stride"], cfg["padding"], cfg["dilation"])
            if ctx.need_bp:
                gbp = (btot * float(cfg.get("b_chain", 1.0))).view(())
            addend = addend_g
        elif ctx.train:
            Cout = g.shape[3]
            glin, rnorm, bgrad = _scale_bwd_cols(g.reshape(-1, Cout), to_nhwc(y_cl).reshape(-1, Cout), scale.reshape(-1, Cout),
                                                 norm.view(-1), BCOS_CONV_EPS, cfg, ctx.need_bp)
            glin = glin.view(N, Ho, Wo, Cout)
            if ctx.need_bp:
                gbp = (bgrad * float(cfg.get("b_chain", 1.0))).view(())
            if need_x:      # gradient through calc_patch_norms: x * PatchSum^T(dL/dnorm / norm), added by the dgrad epilogue
                addend = ops.patch_norm_bwd(xh, rnorm.view(N, Ho, Wo), Cin, (kh, kw), cfg["stride"], cfg["padding"], cfg["dilation"])
        elif ctx.mo_fused:      # the gradient goes to the winning filter of each unit, times its scale
            Cn = g.shape[3]
            glin = ops.maxout_expand(g.reshape(-1, Cn), scale.view(-1, Cn * max_out), max_out).view(N, Ho, Wo, Cn * max_out)
        else:
            glin = ops.mul(g, scale) if scale is not None else g
        if argmax is not None:   # MaxOut (general path): route the gradient to the winning filter of each unit
            Cn = glin.shape[3]
            glin = ops.maxout_scatter(glin.reshape(-1, Cn), argmax.view(-1, Cn), max_out).view(N, Ho, Wo, Cn * max_out)
        gw = gb = None
        if need_w or need_b:
            gl4 = _pad_last(glin).contiguous()
            if need_w and groups > 1:      # per group: its output-gradient columns against its input channels
                cout_g, cin_g = glin.shape[3] // groups, Cin // groups
                parts = [ops.conv2d_wgrad(_pad_last(glin[..., gi * cout_g:(gi + 1) * cout_g]).contiguous(),
                                          xh[..., gi * cin_g:(gi + 1) * cin_g].contiguous(), cin_g, cout_g, (kh, kw), cfg["stride"],
                                          cfg["padding"], cfg["dilation"]) for gi in range(groups)]
                gw = torch.cat(parts, 0).permute(0, 3, 1, 2).contiguous()     # [Cout,kh,kw,Cin/G] -> OIHW
            elif need_w:
                gwk = ops.conv2d_wgrad(gl4, xh, Cin, glin.shape[3], (kh, kw), cfg["stride"], cfg["padding"], cfg["dilation"])
                gw = gwk.permute(0, 3, 1, 2).contiguous()                     # [Cout,kh,kw,Cin] -> OIHW
            if need_b:
                gb = ops.colsum(gl4.view(-1, gl4.shape[3]))[:glin.shape[3]].contiguous()
        if not need_x:
            return None, gw, gb, None, gbp
        plans = cfg["cache"].dgrad(ctx.w_eff, cfg["w_src"], cfg["stride"], cfg["padding"], cfg["dilation"], groups)
        cout_g = glin.shape[3] // groups
        cin_g = Cin // groups
        gx_cl, gx = empty_cl(N, Cin, H, W, gy.device)
        for gi, plan in enumerate(plans):
            gl = glin[..., gi * cout_g:(gi + 1) * cout_g]
            gl = _pad_last(gl).contiguous() if (groups > 1 or cout_g % 4) else gl
            if groups == 1:
                plan.run(gl, H, W, out=gx, addend=addend)
            else:
                gx[..., gi * cin_g:(gi + 1) * cin_g] = plan.run(gl, H, W, addend=addend[gi] if isinstance(addend, list) else None)
        return gx_cl, gw, gb, None, gbp


class BcosLinearFn(Function):
    """y = bcos_linear(x) over the last dimension.  cfg keys: b, max_out, detach, cache, w_src."""

    @staticmethod
    def forward(ctx, x, w_eff, bias, cfg, b_param=None):
        require_hip(x, "BcosLinear")
        b, max_out = float(cfg["b"]), cfg["max_out"]
        need_grad = ctx.needs_input_grad[0]
        Cin = x.shape[-1]
        x2 = x.reshape(-1, Cin)
        x2 = _pad_last(x2 if x2.is_contiguous() else x2.contiguous())
        wk = cfg["cache"].fwd(w_eff, cfg["w_src"]).view(w_eff.shape[0], -1)
        need_w, need_b = ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        need_bp = b_param is not None and ctx.needs_input_grad[4] and _b_gradient_applies(cfg, b)
        train = (need_grad or need_w or need_b or need_bp) and not cfg["detach"] and b != 1.0
        want_scale = bool((need_grad or need_w or need_b or need_bp) and b != 1.0)
        Cout_all = w_eff.shape[0]
        argmax = None
        norm = None
        mo_fused = max_out in (2, 4) and Cout_all % 4 == 0 and not train
        uflag, ugain = _unit(cfg)
        if mo_fused:
            rows = x2.shape[0]
            y = torch.empty((rows, Cout_all // max_out), device=x.device, dtype=torch.float32)
            scale = torch.empty((rows, Cout_all), device=x.device, dtype=torch.float32) if (need_grad or need_w or need_b) else None
            g = dict(N=1, H=1, W=rows, C=x2.shape[1], P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1,
                     TH=1, TW=1, OH=1, OW=rows, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=Cout_all)
            ops.tapconv(x2, wk, g, out=y, scale_out=scale, bias=bias, bcos_mode=BCOS_NONE if b == 1.0 else BCOS_LINEAR_EPS, b=b,
                        flags=(BCOS_EPI_FORCE_POW if cfg.get("force_pow") else 0) | uflag, col_scale=ugain, max_out=max_out)
        elif max_out == 1:
            y, scale, norm = ops.linear_fwd(x2, wk, bias=bias, b=b, want_scale=want_scale, want_norm=train,
                                            flags=(BCOS_EPI_FORCE_POW if cfg.get("force_pow") else 0) | uflag, col_scale=ugain)
        else:
            rows = x2.shape[0]
            lin = torch.empty((rows, Cout_all), device=x.device, dtype=torch.float32)
            norm = torch.empty((rows,), device=x.device, dtype=torch.float32) if b != 1.0 else None
            g = dict(N=1, H=1, W=rows, C=x2.shape[1], P=1, Q=rows, in_sh=1, in_sw=1, dh0=0, dw0=0, dstep_h=1, dstep_w=1,
                     TH=1, TW=1, OH=1, OW=rows, out_sh=1, out_sw=1, out_h0=0, out_w0=0, Cout=Cout_all)
            ops.tapconv(x2, wk, g, out=lin, norm_out=norm, bias=bias,
                        bcos_mode=BCOS_NONE if b == 1.0 else BCOS_LINEAR_EPS, b=b, flags=BCOS_EPI_NORM_ONLY | uflag, col_scale=ugain)
            y, scale, argmax = ops.maxout_scale(lin, norm, Cout_all // max_out, max_out, b, want_scale=want_scale,
                                                want_argmax=need_grad or need_w or need_b)
        ctx.cfg = cfg
        ctx.w_eff = w_eff
        ctx.in_shape = tuple(x.shape)
        ctx.train = bool(train)
        ctx.need_bp = bool(need_bp and train)
        ctx.mo_fused = bool(mo_fused)
        keep_x = x2 if (train or need_w) else None
        keep_y = y if train else None
        keep_n = norm if train else None
        ctx.save_for_backward(*(t for t in (scale, argmax, keep_x, keep_y, keep_n) if t is not None))
        ctx.has = (scale is not None, argmax is not None, keep_x is not None, keep_y is not None, keep_n is not None)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, gy):
        cfg = ctx.cfg
        saved = list(ctx.saved_tensors)
        scale = saved.pop(0) if ctx.has[0] else None
        argmax = saved.pop(0) if ctx.has[1] else None
        x2 = saved.pop(0) if ctx.has[2] else None
        y = saved.pop(0) if ctx.has[3] else None
        norm = saved.pop(0) if ctx.has[4] else None
        max_out = cfg["max_out"]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        Cin = ctx.in_shape[-1]
        g2 = gy.reshape(-1, gy.shape[-1])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        addend = None
        gbp = None
        if ctx.train:
            glin, rnorm, bgrad = _scale_bwd_cols(g2, y, scale, norm, BCOS_LINEAR_EPS, cfg, ctx.need_bp)
            if ctx.need_bp:
                gbp = (bgrad * float(cfg.get("b_chain", 1.0))).view(())
            if need_x:      # gradient through ||x||: x * dL/dnorm / ||x||, added by the dgrad epilogue
                rows = x2.shape[0]
                addend = ops.patch_norm_bwd(x2.view(1, 1, rows, x2.shape[1]), rnorm.view(1, 1, rows), Cin, (1, 1), (1, 1), (0, 0), (1, 1))
        elif ctx.mo_fused:
            glin = ops.maxout_expand(g2, scale, max_out)
        else:
            glin = ops.mul(g2, scale) if scale is not None else g2
        if argmax is not None:   # MaxOut (general path): route the gradient to the winning filter of each unit
            glin = ops.maxout_scatter(glin.contiguous(), argmax.view(glin.shape), max_out)
        gw = gb = None
        if need_w or need_b:
            gl4 = _pad_last(glin).contiguous()
            if need_w:
                rows = x2.shape[0]
                gw = ops.conv2d_wgrad(gl4.view(1, 1, rows, gl4.shape[1]), x2.view(1, 1, rows, x2.shape[1]), Cin, glin.shape[1],
                                      (1, 1), (1, 1), (0, 0), (1, 1)).view(glin.shape[1], Cin)
            if need_b:
                gb = ops.colsum(gl4)[:glin.shape[1]].contiguous()
        if not need_x:
            return None, gw, gb, None, gbp
        plan = cfg["cache"].dgrad(ctx.w_eff, cfg["w_src"], (1, 1), (0, 0), (1, 1), 1)[0]
        rows = glin.shape[0]
        glin = _pad_last(glin).contiguous()
        gx = plan.run(glin.view(1, 1, rows, glin.shape[1]), 1, rows, addend=addend)      # [1,1,rows,Cin]
        return gx.view(ctx.in_shape), gw, gb, None, gbp


def plain_conv2d(x, w_eff, bias, stride, padding, dilation, groups, cache, w_src):
    """Un-scaled convolution on the same kernel (NormedConv2d called on its own)."""
    cfg = dict(stride=stride, padding=padding, dilation=dilation, groups=groups, b=1, max_out=1, detach=True,
               cache=cache, w_src=w_src)
    return BcosConv2dFn.apply(x, w_eff, bias, cfg)


def plain_linear(x, w_eff, bias, cache, w_src):
    cfg = dict(b=1, max_out=1, detach=True, cache=cache, w_src=w_src)
    return BcosLinearFn.apply(x, w_eff, bias, cfg)
