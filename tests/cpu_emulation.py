"""CPU emulation of the C-ABI *semantics* for host-logic tests (TEST INFRASTRUCTURE, `-m "not gpu"` only).

The product has no CPU path.  To test the host side (launch descriptors, parity-class decomposition of strided
input gradients, the fused engine's forward/backward plan, sharding) without a GPU, the tests monkeypatch the
thin wrappers in `bcos_hip.ops` with the pure-torch interpreters below, which implement exactly what
include/bcos_hip.h documents for each entry point.  Nothing here is reachable from the product.
"""
import torch
import torch.nn.functional as F

from bcos_hip.lib import (BCOS_CONV_EPS, BCOS_EPI_FORCE_POW, BCOS_EPI_GATE2_FROM_MUL, BCOS_EPI_MUL_FROM_ACT, BCOS_EPI_NORM_ONLY,
                          BCOS_EPI_SCALE_GATE_LSB, BCOS_LINEAR_EPS, BCOS_NONE)


def tapconv(a, wt, geom, *, out=None, out2=None, scale_out=None, norm_out=None, bias=None, ch_scale=None,
            ch_shift=None, addend=None, mul=None, mul2=None, gate2=None, relu_gate=None, bcos_mode=BCOS_NONE,
            b=2.0, relu=False, flags=0, contraction=None, track_absmax=None, track_absmax2=None, max_out=1,
            mul_norm=None, mul_csc=None, mul_csh=None, addend_sub=0, col_scale=None, row_scale=None, a_sumsq=None, rowadd=None,
            rowadd_scale=None):
    # contraction / track_absmax*: how the device evaluates the products and which side tensors it emits for the next
    # launch's operand scaling -- no effect on the documented result
    g = dict(a_pitch=0, out_pitch=0, norm_pitch=0, out_cgroup=0, groups=0)
    g.update(geom)
    if g["groups"] > 1:       # grouped launch (include/bcos_hip.h: bcos_tapconv_geom.groups): the documented result = one launch per group
        G, C, Cout = g["groups"], g["C"], g["Cout"]
        sub = dict(g, groups=0, a_pitch=g["a_pitch"] or G * C, out_pitch=g["out_pitch"] or G * Cout, norm_pitch=g["norm_pitch"] or G)
        a4 = a if a.dim() == 4 else a.view(g["N"], g["H"], g["W"], -1)

        def cols(t, k):
            return None if t is None else (t if t.dim() == 4 else t.view(g["N"], g["OH"], g["OW"], -1))[..., k * Cout:]
        for k in range(G):
            tapconv(a4[..., k * C:], wt.reshape(G * Cout, -1)[k * Cout:(k + 1) * Cout], sub, out=cols(out, k), out2=cols(out2, k),
                    scale_out=cols(scale_out, k), norm_out=None if norm_out is None else norm_out.view(g["N"], g["OH"], g["OW"], -1)[..., k:],
                    bias=None if bias is None else bias[k * Cout:(k + 1) * Cout],
                    ch_scale=None if ch_scale is None else ch_scale[k * Cout:(k + 1) * Cout],
                    ch_shift=None if ch_shift is None else ch_shift[k * Cout:(k + 1) * Cout], addend=cols(addend, k), mul=cols(mul, k),
                    mul2=cols(mul2, k), gate2=cols(gate2, k), relu_gate=cols(relu_gate, k), bcos_mode=bcos_mode, b=b, relu=relu, flags=flags,
                    col_scale=None if col_scale is None else col_scale[k * Cout:(k + 1) * Cout])
        return
    N, H, W, C = g["N"], g["H"], g["W"], g["C"]
    P, Q, TH, TW, Cout = g["P"], g["Q"], g["TH"], g["TW"], g["Cout"]
    a4 = (a if a.dim() == 4 else a.view(N, H, W, -1))[..., :C]
    w = wt.reshape(Cout, TH, TW, C)
    acc = torch.zeros(N, P, Q, Cout, dtype=torch.float64)
    ss = torch.zeros(N, P, Q, dtype=torch.float64)
    ii = torch.arange(P) * g["in_sh"]
    jj = torch.arange(Q) * g["in_sw"]
    for th in range(TH):
        for tw in range(TW):
            ih = ii + g["dh0"] + th * g["dstep_h"]
            iw = jj + g["dw0"] + tw * g["dstep_w"]
            vh = (ih >= 0) & (ih < H)
            vw = (iw >= 0) & (iw < W)
            patch = a4[:, ih.clamp(0, H - 1)][:, :, iw.clamp(0, W - 1)].double()
            patch = patch * (vh[:, None] & vw[None, :])[None, :, :, None]
            acc += torch.einsum("npqc,oc->npqo", patch, w[:, th, tw].double())
            ss += (patch * patch).sum(-1)
    v = acc
    if flags & 32:        # BCOS_EPI_UNIT_NORM_W: every accumulator column divided by the norm of its (raw) weight row
        v = v / w.reshape(Cout, -1).double().norm(dim=1)
    if col_scale is not None:
        v = v * col_scale.double()
    if row_scale is not None:     # bcos_epilogue.row_scale / a_sumsq: indexed by output pixel (these launches map rows to pixels one to one)
        v = v * row_scale.double().view(N, P, Q, 1)
    if a_sumsq is not None:
        ss = a_sumsq.double().view(N, P, Q)
    if bias is not None:
        v = v + bias.double()
    if max_out > 1:       # fused MaxOut (include/bcos_hip.h: bcos_epilogue.max_out): out narrow, scale_out at the winner
        M = max_out
        vm, arg = v.view(N, P, Q, Cout // M, M).max(-1)            # first maximum, like torch.max
        sm = torch.ones_like(vm)
        if bcos_mode != BCOS_NONE:
            nrm = ss.sqrt() + 1e-12 if bcos_mode == BCOS_LINEAR_EPS else (ss + 1e-6).sqrt()
            sm = vm.abs() / nrm[..., None] if (b == 2.0 and not (flags & BCOS_EPI_FORCE_POW)) else \
                ((vm / nrm[..., None]).abs() + 1e-6).pow(b - 1)
            if norm_out is not None:
                norm_out.view(N, P, Q)[...] = nrm.to(norm_out.dtype)
        if out is not None:
            out.view(N, P, Q, -1)[..., :Cout // M] = (vm * sm).to(out.dtype)
        if scale_out is not None:
            t = torch.zeros(N, P, Q, Cout // M, M, dtype=torch.float64)
            t.scatter_(-1, arg[..., None], sm[..., None])
            scale_out.view(N, P, Q, Cout)[...] = t.view(N, P, Q, Cout).to(scale_out.dtype)
        return
    s = torch.ones_like(v)
    nrm = None
    if bcos_mode != BCOS_NONE:
        nrm = ss.sqrt() + 1e-12 if bcos_mode == BCOS_LINEAR_EPS else (ss + 1e-6).sqrt()
        if not (flags & BCOS_EPI_NORM_ONLY):
            if b == 2.0 and not (flags & BCOS_EPI_FORCE_POW):
                s = v.abs() / nrm[..., None]
            else:
                s = ((v / nrm[..., None]).abs() + 1e-6).pow(b - 1)
            v = v * s
    if ch_scale is not None:
        v = v * ch_scale.double()
        s = s * ch_scale.double()
    if ch_shift is not None:
        v = v + ch_shift.double()
    oh = torch.arange(P) * g["out_sh"] + g["out_h0"]
    ow = torch.arange(Q) * g["out_sw"] + g["out_w0"]

    def v4(t):
        return t if t.dim() == 4 else t.view(N, g["OH"], g["OW"], -1)

    cg = g["out_cgroup"]
    if cg:      # depth to space (include/bcos_hip.h: out_cgroup): column (dh * out_sw + dw) * cg + c -> pixel (i*out_sh + dh, j*out_sw + dw), channel c
        osh, osw = g["out_sh"], g["out_sw"]

        def d2s_view(t):
            return v4(t)[:, :P * osh, :Q * osw, :cg].reshape(N, P, osh, Q, osw, cg)

        def rd(t):
            return d2s_view(t).permute(0, 1, 3, 2, 4, 5).reshape(N, P, Q, Cout).double()

        def wr(t, val):
            v4(t)[:, :P * osh, :Q * osw, :cg] = val.to(t.dtype).view(N, P, Q, osh, osw, cg).permute(0, 1, 3, 2, 4, 5).reshape(
                N, P * osh, Q * osw, cg)
    else:
        def rd(t):
            return v4(t)[:, oh][:, :, ow][..., :Cout].double()

        def wr(t, val):
            v4(t)[:, oh[:, None], ow[None, :], :Cout] = val.to(t.dtype)

    if addend is not None and addend_sub > 1:
        # include/bcos_hip.h: bcos_epilogue.addend_sub -- the addend holds the output pixels on the s-grid, zero elsewhere
        sb = int(addend_sub)
        assert not cg and g["groups"] <= 1 and max_out <= 1
        sub = addend if addend.dim() == 4 else addend.view(N, -(-g["OH"] // sb), -(-g["OW"] // sb), -1)
        full = torch.zeros((N, g["OH"], g["OW"], sub.shape[-1]), dtype=sub.dtype)
        full[:, ::sb, ::sb] = sub
        v = v + rd(full)
    elif addend is not None:
        v = v + rd(addend)
    if rowadd is not None:        # include/bcos_hip.h: bcos_epilogue.rowadd -- a plain gradient launch: + rowadd_scale[output pixel] * rowadd
        assert bcos_mode == BCOS_NONE and mul is None and out2 is None and addend_sub <= 1 and not cg and rowadd_scale is not None
        rs = rowadd_scale.view(N, g["OH"], g["OW"], 1)
        v = v + rd(rowadd) * rs[:, oh][:, :, ow].double()
    if relu == 2:
        gate = 0.5 * (1 + torch.erf(v / 2 ** 0.5))
        s = s * gate
        v = v * gate
    elif relu:
        open_gate = (rd(relu_gate) > 0) if relu_gate is not None else (v > 0)
        s = torch.where(open_gate, s, torch.zeros_like(s))
        v = torch.where(open_gate, v, torch.zeros_like(v))
        if flags & BCOS_EPI_SCALE_GATE_LSB:       # gate decision in the low mantissa bit of the stored fp32 multiplier
            bits = s.float().contiguous().view(torch.int32)
            s = torch.where(open_gate, (bits | 1).view(torch.float32).double(), torch.zeros_like(s))
    mul_val = rd(mul) if mul is not None else None
    if mul is not None and (flags & BCOS_EPI_MUL_FROM_ACT):        # `mul` is the kept activation: rebuild t (include/bcos_hip.h)
        a = mul_val
        csc = mul_csc.double() if mul_csc is not None else torch.ones(Cout, dtype=torch.float64)
        csh = mul_csh.double() if mul_csh is not None else torch.zeros(Cout, dtype=torch.float64)
        mn = (mul_norm if mul_norm.dim() == 4 else mul_norm.view(N, g["OH"], g["OW"], 1))[:, oh][:, :, ow][..., :1].double()
        den = csc.abs() * mn
        mul_val = torch.where((a > 0) & (den > 0), csc * ((a - csh).abs() / den.clamp_min(1e-300)).sqrt(), torch.zeros_like(a))
    if out is not None:
        wr(out, v * mul_val if mul is not None else v)
    if out2 is not None:
        o2 = v
        if mul2 is not None:
            o2 = o2 * rd(mul2)
        if (flags & BCOS_EPI_GATE2_FROM_MUL) and mul is not None:
            mbits = rd(mul).float().contiguous().view(torch.int32)
            o2 = torch.where((mbits & 1).bool(), o2, torch.zeros_like(o2))
        elif gate2 is not None:
            o2 = torch.where(rd(gate2) > 0, o2, torch.zeros_like(o2))
        wr(out2, o2)
    if scale_out is not None:
        wr(scale_out, s)
    if norm_out is not None and nrm is not None:
        no = norm_out if norm_out.dim() == 4 else norm_out.view(N, g["OH"], g["OW"], 1)
        no[:, oh[:, None], ow[None, :], 0] = nrm.to(no.dtype)


def prep_input(x, mean6, std6, cpad=8, add_inverse=False, want_absmax=False):
    if add_inverse:
        x = torch.cat([x, 1 - x], 1)
    xn = (x - mean6.view(1, 6, 1, 1)) / std6.view(1, 6, 1, 1)
    return F.pad(xn.permute(0, 2, 3, 1), (0, cpad - 6)).contiguous()


def finalize_explanation(gxn, x, std6, add_inverse=False, want_weights=True, want_contrib=True, weights_out=None, contrib_out=None):
    if add_inverse:
        x = torch.cat([x, 1 - x], 1)
    w = gxn[..., :6].permute(0, 3, 1, 2) / std6.view(1, 6, 1, 1)
    wo, co = (w.contiguous() if want_weights else None), ((x * w).sum(1) if want_contrib else None)
    if weights_out is not None and wo is not None:
        wo = weights_out.copy_(wo)
    if contrib_out is not None and co is not None:
        co = contrib_out.copy_(co)
    return wo, co


def avgpool2d_fwd(x, k, s, p, out=None, want_absmax=False):
    y = F.avg_pool2d(x.permute(0, 3, 1, 2), k, s, p).permute(0, 2, 3, 1).contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


def avgpool2d_bwd(gy, H, W, k, s, p, mul=None, out=None, want_absmax=False):
    with torch.enable_grad():
        x = torch.zeros(gy.shape[0], gy.shape[3], H, W, requires_grad=True)
        y = F.avg_pool2d(x, k, s, p)
        (gx,) = torch.autograd.grad(y, x, gy.permute(0, 3, 1, 2))
    gx = gx.permute(0, 2, 3, 1).contiguous()
    if mul is not None:
        gx = gx * mul
    if out is not None:
        out.copy_(gx)
        return out
    return gx


def global_avgpool_logits(x, temperature=None, bias=None):
    y = x.mean((1, 2))
    if temperature is not None:
        y = y / temperature
    if bias is not None:
        y = y + bias
    return y


def head_onehot_grad(cls, scale, temperature=None, out=None):
    N, H, W, C = scale.shape
    coef = (1.0 if temperature is None else 1.0 / temperature) / (H * W)
    onehot = F.one_hot(cls, C).to(scale.dtype).view(N, 1, 1, C)
    return onehot * scale * coef


def head_rank1_grad(cls, scale, w, temperature=None, row_scale=None, mul=None, want_out2=False, want_absmax=False, mul2=None, gate2=None,
                    gate2_from_mul=False, want_absmax2=False):
    N, R, K = scale.shape
    coef = (1.0 if temperature is None else 1.0 / temperature) / R
    a = coef * scale[torch.arange(N), :, cls]                      # [N, R]
    if row_scale is not None:
        a = a * row_scale.view(N, R)
    v = (a.unsqueeze(-1) * w[cls].unsqueeze(1)).reshape(N * R, -1)
    out = v * mul.reshape(v.shape) if mul is not None else v
    out2 = None
    if want_out2 or mul2 is not None or gate2 is not None or gate2_from_mul or want_absmax2:
        out2 = v * mul2.reshape(v.shape) if mul2 is not None else v
        if gate2_from_mul:
            out2 = out2 * (mul.reshape(v.shape).contiguous().view(torch.int32) & 1).to(out2.dtype)
        elif gate2 is not None:
            out2 = out2 * (gate2.reshape(v.shape) > 0).to(out2.dtype)
    return out, out2


def argmax_rows(x2d):
    v, i = x2d.max(1)
    return i, v


def mul(a, b, out=None):
    return a * b


def channel_affine(x, scale, shift=None, relu=False, out=None):
    y = x * scale
    if shift is not None:
        y = y + shift
    if relu:
        y = y.clamp_min(0)
    if out is not None:
        out.copy_(y)
        return out
    return y


def channel_affine_add(x, scale, shift, addend, relu=False, out=None):
    y = x * scale
    if shift is not None:
        y = y + shift
    y = y + addend
    if relu:
        y = y.clamp_min(0)
    if out is not None:
        out.copy_(y)
        return out
    return y


def channel_affine_rows(x, scale, shift=None, addend=None, relu=False):
    r = x * scale + (shift if shift is not None else 0)
    if addend is not None:
        r = r + addend
    return torch.relu(r) if relu else r


def relu_bwd(g, act, out=None):
    r = torch.where(act > 0, g, torch.zeros_like(g))
    if out is not None:
        out.copy_(r)
        return out
    return r


def weight_rownorm_scale(w2d, gain=None):
    flat = w2d.reshape(w2d.shape[0], -1)
    out = flat / flat.norm(dim=1, keepdim=True)
    if gain is not None:
        out = gain.view(-1, 1) * out
    return out.view_as(w2d)


def rows_normalize(x2d, want_y=True, want_inv=False):
    nrm = x2d.norm(dim=1, keepdim=True)
    return (x2d / nrm) if want_y else None, (1.0 / nrm.view(-1)) if want_inv else None


def cosine_grad(u2d, w2d, l, inv, coef=None):
    k = inv if coef is None else coef * inv
    return k.view(-1, 1) * (w2d - l.view(-1, 1) * u2d)


def contrib_map(x, gx):
    return (x * gx).sum(1)


def render_explanations(x, weights, smooth=15, alpha_percentile=99.5, want_quantiles=False):
    import torch.nn.functional as F
    x6 = torch.cat([x, 1 - x], 1) if x.shape[1] == 3 else x
    contribs = (x6 * weights).sum(1, keepdim=True)
    d = (weights / (weights.abs().amax(1, keepdim=True) + 1e-12)).clamp(min=0)
    rgb = d[:, :3] / (d[:, :3] + d[:, 3:] + 1e-12)
    alpha = weights.norm(p=2, dim=1, keepdim=True)
    alpha = torch.where(contribs < 0, torch.full_like(alpha, 1e-12), alpha)
    if smooth and smooth > 1:
        alpha = F.avg_pool2d(alpha, smooth, stride=1, padding=(smooth - 1) // 2)
    qv = torch.stack([torch.quantile(a, q=alpha_percentile / 100) for a in alpha])
    alpha = (alpha / qv.view(-1, 1, 1, 1)).clip(0, 1)
    rgba = torch.cat([rgb, alpha], 1).permute(0, 2, 3, 1).contiguous()
    return (rgba, qv) if want_quantiles else rgba


def box_filter(maps, k):
    import torch.nn.functional as F
    return F.avg_pool2d(maps[:, None], k, stride=1, padding=(k - 1) // 2)[:, 0]


def localisation_fractions(attr, cell_h, cell_w, neg=False):
    import torch.nn.functional as F
    a = (-attr if neg else attr).clamp(min=0)[:, None]
    contribs = F.avg_pool2d(a, (cell_h, cell_w), stride=(cell_h, cell_w)).permute(0, 1, 3, 2).reshape(a.shape[0], -1)
    total = contribs.sum(1, keepdim=True)
    return torch.where(total * contribs > 0, contribs / total, torch.zeros_like(contribs))


def ensure_absmax(t):
    return t


def maxout_expand(gy2d, t2d, max_out):
    return t2d * gy2d.repeat_interleave(max_out, dim=1)


# ---- training-mode backward (include/bcos_hip.h: bcos_train_scale_bwd ... bcos_channel_axpby) ----------------------------
def train_scale_bwd(gy2d, y2d, s2d, norm, mode, b, force_pow=False, want_bgrad=False, want_absmax=False, bn=None):
    if bn is not None:                     # include/bcos_hip.h: bcos_train_scale_bwd_bn
        g, mean, coef = bn
        gy2d = gy2d * g
        if coef is not None:
            gy2d = gy2d + (y2d - (mean if mean is not None else 0)) * coef
    nrm = norm.view(-1, 1)
    bgrad = None
    if b == 2 and not force_pow:
        glin = gy2d * 2 * s2d
        dnorm = (gy2d * (-y2d / nrm)).sum(1)
    else:
        q = (y2d / s2d).abs() / nrm           # |lin| / norm, lin = y / s
        c = q + 1e-6
        ratio = q / c
        glin = gy2d * s2d * (1 + (b - 1) * ratio)
        dnorm = (gy2d * (-(b - 1) * y2d * ratio / nrm)).sum(1)
        if want_bgrad:
            bgrad = (gy2d.double() * y2d.double() * c.double().log()).sum().float().view(1)
    div = (norm - 1e-12).clamp(min=1e-30) if mode == BCOS_LINEAR_EPS else norm
    return glin, dnorm / div, bgrad


def weight_rownorm_bwd(w2d, g2d, gain=None, want_gw=True, want_ggain=False):
    nrm = w2d.norm(dim=1, keepdim=True)
    what = w2d / nrm
    dot = (what * g2d).sum(1, keepdim=True)
    gn = gain.view(-1, 1) if gain is not None else 1.0
    gw = gn / nrm * (g2d - what * dot) if want_gw else None
    return gw, (dot.view(-1) if want_ggain else None)


def layernorm_bwd(gy2d, x2d, weight, rstd, want_xhat=False, addend=None):
    x = x2d.double()
    xhat = (x - x.mean(1, keepdim=True)) * rstd.double().view(-1, 1)
    h = gy2d.double() * (weight.double() if weight is not None else 1.0)
    gx = rstd.double().view(-1, 1) * (h - h.mean(1, keepdim=True) - xhat * (h * xhat).mean(1, keepdim=True))
    if addend is not None:
        gx = gx.float().double() + addend.double()          # (the kernel rounds the LayerNorm term to fp32 first)
    return gx.float(), (xhat.float() if want_xhat else None)


def groupnorm_bwd(gy_nhwc, x_nhwc, groups, weight, rstd, want_xhat=False):
    N, H, W, Cc = gy_nhwc.shape
    xg = x_nhwc.reshape(N, H * W, groups, Cc // groups).double()
    rs = rstd.double().view(N, 1, groups, 1)
    xhat = (xg - xg.mean(dim=(1, 3), keepdim=True)) * rs
    h = (gy_nhwc.double() * (weight.double() if weight is not None else 1.0)).reshape(N, H * W, groups, Cc // groups)
    gx = rs * (h - h.mean(dim=(1, 3), keepdim=True) - xhat * (h * xhat).mean(dim=(1, 3), keepdim=True))
    return gx.reshape(N, H, W, Cc).float(), (xhat.reshape(N, H, W, Cc).float() if want_xhat else None)


def gelu_bwd(gy, x):
    v = x.double()
    Phi = 0.5 * (1 + torch.erf(v / 2 ** 0.5))
    phi = torch.exp(-0.5 * v * v) / (2 * torch.pi) ** 0.5
    return (gy.double() * (Phi + v * phi)).float()


def attention_bwd(qkv, stats, out, gout, heads, scale):
    B, T, three_inner = qkv.shape
    inner = three_inner // 3
    d = inner // heads
    q, k, v = (t.reshape(B, T, heads, d).permute(0, 2, 1, 3).double() for t in qkv.split(inner, dim=-1))
    go = gout.reshape(B, T, heads, d).permute(0, 2, 1, 3).double()
    P = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    dP = go @ v.transpose(-1, -2)
    D = (P * dP).sum(-1, keepdim=True)
    dS = P * (dP - D) * scale
    gq, gk, gv = dS @ k, dS.transpose(-1, -2) @ q, P.transpose(-1, -2) @ go
    pack = lambda t: t.permute(0, 2, 1, 3).reshape(B, T, inner)        # noqa: E731
    return torch.cat([pack(gq), pack(gk), pack(gv)], dim=-1).float()


def groupnorm_fwd(x_nhwc, groups, weight, bias, eps, want_rstd=False):
    N, H, W, Cc = x_nhwc.shape
    xg = x_nhwc.reshape(N, H * W, groups, Cc // groups).double()
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    rstd = 1.0 / (var + eps).sqrt()
    y = ((xg - mean) * rstd).reshape(N, H, W, Cc)
    if weight is not None:
        y = y * weight.double()
    if bias is not None:
        y = y + bias.double()
    return y.float(), (rstd.reshape(-1).float() if want_rstd else None)


def groupnorm_bwd_detached(gy_nhwc, groups, weight, rstd):
    N, H, W, Cc = gy_nhwc.shape
    h = gy_nhwc.double() * (weight.double() if weight is not None else 1.0)
    h = h.reshape(N, H * W, groups, Cc // groups) * rstd.double().view(N, 1, groups, 1)
    return (h - h.mean(dim=(1, 3), keepdim=True)).reshape(N, H, W, Cc).float()


def maxout_scatter(g2d, argmax2d, max_out):
    rows, Cout = g2d.shape
    full = torch.zeros(rows, Cout, max_out, dtype=g2d.dtype)
    full.scatter_(2, argmax2d.long().view(rows, Cout, 1), g2d.view(rows, Cout, 1))
    return full.view(rows, Cout * max_out)


def patch_norm_bwd(x, rnorm, C_used, kernel, stride, padding, dilation, addend=None):
    N, H, W, _ = x.shape
    with torch.enable_grad():          # the transposed patch sum = adjoint of a convolution with a kernel of ones
        z = torch.zeros(N, 1, H, W, requires_grad=True)
        patch_sum = F.conv2d(z, torch.ones(1, 1, kernel[0], kernel[1]), None, stride, padding, dilation)
        (t,) = torch.autograd.grad(patch_sum, z, rnorm[:, None].detach())
    out = x[..., :C_used] * t[:, 0, :, :, None]
    return out + addend if addend is not None else out


def conv2d_wgrad(glin, x, C_used, Cout, kernel, stride, padding, dilation, out=None):
    xn = x[..., :C_used].permute(0, 3, 1, 2).double()
    g = glin[..., :Cout].permute(0, 3, 1, 2).double()
    with torch.enable_grad():          # called from inside autograd.Function.backward, where grad mode is off
        w = torch.zeros(Cout, C_used, kernel[0], kernel[1], dtype=torch.float64, requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv2d(xn.detach(), w, None, stride, padding, dilation), w, g.detach())
    if out is not None:                # the kernel ACCUMULATES into a zeroed tensor
        assert not out.any()
        return out.add_(gw.permute(0, 2, 3, 1).float())
    return gw.permute(0, 2, 3, 1).float().contiguous()


def colsum(a2d, b2d=None, shift_a=None, shift_b=None):
    a = a2d.double() - (shift_a.double() if shift_a is not None else 0)
    if b2d is not None:
        a = a * (b2d.double() - (shift_b.double() if shift_b is not None else 0))
    return a.sum(0).float()


def bn_batch_stats(y2d, weight, eps, running_var=None, momentum=0.0):
    mean = y2d.mean(0)
    var = y2d.var(0, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    g = rstd if weight is None else weight * rstd
    if running_var is not None:
        running_var.copy_((1 - momentum) * running_var + momentum * var)
    return mean, var, rstd, g


def relu_bwd_colsums(g2d, act2d, y2d, rstd=None, gvec=None, want_sg=False, want_gw=False, want_coef=False):
    ga = g2d * (act2d > 0) if act2d is not None else g2d
    sgx = (ga * y2d).sum(0)
    sg = ga.sum(0) if want_sg else None
    gw = sgx * rstd if want_gw else None
    coef = -(gvec * sgx) * rstd * rstd / y2d.shape[0] if want_coef else None
    return ga, sgx, sg, gw, coef


def channel_axpby(a, sa, b=None, mb=None, sb=None, out=None):
    r = a * sa
    if b is not None:
        r = r + (b - (mb if mb is not None else 0)) * sb
    if out is not None:
        out.copy_(r)
        return out
    return r


def tapconv_group(a, wts, geoms, *, out, addend=None, mul=None):
    for w, g in zip(wts, geoms):
        tapconv(a, w, g, out=out, addend=addend, mul=mul)


def install(monkeypatch):
    """Patch bcos_hip.ops with the emulators (pytest monkeypatch fixture) and lift the HIP-device checks."""
    from bcos_hip import ops
    for name in ("tapconv", "prep_input", "finalize_explanation", "avgpool2d_fwd", "avgpool2d_bwd",
                 "global_avgpool_logits", "head_onehot_grad", "head_rank1_grad", "argmax_rows", "mul", "channel_affine", "channel_affine_add", "relu_bwd",
                 "weight_rownorm_scale", "rows_normalize", "cosine_grad", "contrib_map", "maxout_scale", "layernorm_fwd", "layernorm_stats", "layernorm_bwd_detached",
                 "gelu_gate", "add_rows_bcast", "attention_fwd", "attention_bwd_v", "finalize_explanation_patches",
                 "render_explanations", "box_filter", "localisation_fractions", "tapconv_group", "ensure_absmax",
                 "train_scale_bwd", "patch_norm_bwd", "conv2d_wgrad", "colsum", "channel_axpby", "bn_batch_stats", "relu_bwd_colsums", "channel_affine_rows", "maxout_expand",
                 "weight_rownorm_bwd", "maxout_scatter", "groupnorm_fwd", "groupnorm_bwd_detached",
                 "layernorm_bwd", "gelu_bwd", "attention_bwd", "groupnorm_bwd"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(ops, "require_device", lambda t, who="": None)
    from bcos.modules import _hipfn
    monkeypatch.setattr(_hipfn, "require_hip", lambda t, who="": None)


def maxout_scale(lin2d, norm, Cout, max_out, b, groups=1, want_scale=False, want_argmax=False, out=None):
    rows = lin2d.shape[0]
    u = lin2d.view(rows, Cout, max_out)
    best, arg = u.max(-1)
    s = torch.ones_like(best)
    if norm is not None:
        nrm = norm.view(rows, groups).repeat_interleave(Cout // groups, dim=1)
        s = best.abs() / nrm if b == 2.0 else ((best / nrm).abs() + 1e-6).pow(b - 1)
    y = s * best
    if out is not None:
        out.copy_(y)
        y = out
    return y, (s if want_scale else None), (arg.to(torch.int32) if want_argmax else None)


def layernorm_fwd(x2d, weight, bias, eps, want_rstd=False, out=None, want_absmax=False):
    var, mean = torch.var_mean(x2d.double(), dim=-1, unbiased=False, keepdim=True)
    sd = (var + eps).sqrt()
    y = (x2d.double() - mean) / sd
    if weight is not None:
        y = y * weight.double()
    if bias is not None:
        y = y + bias.double()
    return y.float(), ((1 / sd).float().view(-1) if want_rstd else None)


def layernorm_stats(x2d, weight, bias, eps, want_zsumsq=False, want_absmax=False):
    y, rstd = layernorm_fwd(x2d, weight, bias, eps, want_rstd=True)
    return rstd, ((y.double() ** 2).sum(-1).float() if want_zsumsq else None)


def layernorm_bwd_detached(gy2d, weight, rstd, addend=None, mul2=None, want_out=True, want_out2=False, out=None, want_absmax2=False):
    h = gy2d.double() * (weight.double() if weight is not None else 1.0) * rstd.double().view(-1, 1)
    g = h - h.mean(-1, keepdim=True)
    if addend is not None:
        g = g + addend.double()
    o2 = (g * mul2.double() if mul2 is not None else g).float() if want_out2 else None
    return (g.float() if want_out else None), o2


def gelu_gate(x, want_gate=False, out=None):
    gate = 0.5 * (1 + torch.erf(x / 2 ** 0.5))
    return gate * x, (gate if want_gate else None)


def add_rows_bcast(x, pe):
    x.view(-1, pe.numel()).add_(pe.reshape(1, -1))
    return x


def _attn_probs(qkv, heads, scale):
    B, T, three = qkv.shape
    inner = three // 3
    q, k, v = (t.view(B, T, heads, inner // heads).transpose(1, 2).double() for t in qkv.split(inner, dim=-1))
    return torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1), v, inner


def attention_fwd(qkv, heads, scale, want_stats=False, want_absmax=False):
    p, v, inner = _attn_probs(qkv, heads, scale)
    B, T, _ = qkv.shape
    out = (p @ v).transpose(1, 2).reshape(B, T, inner).float()
    return out, (torch.zeros(B, heads, T, 2) if want_stats else None)


def attention_bwd_v(qkv, stats, gout, heads, scale, want_absmax=False):
    p, v, inner = _attn_probs(qkv, heads, scale)
    B, T, _ = qkv.shape
    g = gout.view(B, T, heads, inner // heads).transpose(1, 2).double()
    return (p.transpose(-1, -2) @ g).transpose(1, 2).reshape(B, T, inner).float()


def finalize_explanation_patches(gp, x, std6, patch, add_inverse=False, want_weights=True, want_contrib=True, weights_out=None,
                                 contrib_out=None):
    if add_inverse:
        x = torch.cat([x, 1 - x], 1)
    N, _, H, W = x.shape
    gh, gw = H // patch, W // patch
    cpad = gp.shape[-1] // (patch * patch)
    g = gp.view(N, gh, gw, patch, patch, cpad)[..., :6].permute(0, 5, 1, 3, 2, 4).reshape(N, 6, H, W)
    w = g / std6.view(1, 6, 1, 1)
    wo = (w.contiguous() if weights_out is None else weights_out.copy_(w)) if want_weights else None
    co = ((x * w).sum(1) if contrib_out is None else contrib_out.copy_((x * w).sum(1))) if want_contrib else None
    return wo, co


class _Setter:
    """Stand-in for pytest's monkeypatch in spawned worker processes (no undo needed there)."""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def install_permanent():
    install(_Setter)
