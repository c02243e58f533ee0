"""Zero-shot classification head of the B-cosified CLIP image encoder.

Reference: `clip_evaluate` (bcos/training/trainer.py:104-132) and CLIP_benchmark's `run_classification`
(CLIP_benchmark/clip_benchmark/metrics/zeroshot_classification.py:112-141):
    f = model(images);  f /= ||f||_2;  logits = 100 * f @ W_text          (W_text [D, n_classes], loaded from file)
Both steps run on the library's kernels: the row normalisation is bcos_weight_rownorm_scale (one wavefront per row,
gain 100), the product the fp32-MFMA GEMM.  With images sharded over GPUs, each rank computes its rows and the
per-rank embeddings / logits are all-gathered once (bcos_hip.dist.all_gather_rows).
"""
import torch

from . import ops


def zeroshot_logits(features: torch.Tensor, text_weights: torch.Tensor, scale: float = 100.0,
                    attn_unpool: bool = False, cos_power: float = 1) -> torch.Tensor:
    """features [N, D] (un-normalised image embeddings), text_weights [D, K] -> logits [N, K].

    `attn_unpool` head (trainer.py:119-123, bcosattnpool.py:23-32): features are [(HW), N, D]; every location is
    scored, logits * |logits|^(cos_power - 1), and the locations are summed."""
    lead = features.shape[:-1]
    d = features.shape[-1]
    rows = features.reshape(-1, d).contiguous()
    gain = torch.full((rows.shape[0],), float(scale), device=features.device, dtype=torch.float32)
    f = ops.weight_rownorm_scale(rows, gain)                            # scale * f / ||f||
    logits = ops.matmul_nt(f, text_weights.t().contiguous()).reshape(*lead, -1)
    if attn_unpool:
        if cos_power != 1:
            logits = logits * logits.abs() ** (cos_power - 1)
        logits = logits.sum(0)
    return logits


def topk_accuracy(logits: torch.Tensor, target: torch.Tensor, topk=(1, 5)):
    """clip_accuracy (trainer.py:99-102)."""
    pred = logits.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [float(correct[:k].reshape(-1).float().sum().item()) for k in topk]


def zeroshot_attribution(engine, images: torch.Tensor, text_weights: torch.Tensor, targets=None, pool_cosine: float = 1,
                         norm_max_cosine: bool = False, want_weights: bool = True, gates=None):
    """Explanation of the zero-shot TEXT logit through the whole image encoder, batched: what
    `compute_attributions` of interpretability/analyses/text_localisation.py:68-104 computes image by image --
        outa = model(img);  f = outa / outa.norm(dim=-1);  logits = f @ W_text;  logits.max(1).values.backward(inputs=[img])
    in explanation mode (the feature normalisation is NOT detached: its derivative is part of the explanation).

    `engine`: a ResNetEngine over a B-cosified CLIP image encoder; `text_weights` [D, K].  `targets` [N] picks the text
    class per image (default: the best-scoring one, as in the reference).  For an `attn_unpool` head (output
    [(HW), N, D']: one unit vector per location, bcosattnpool.py:23-32) the location logits are pooled as in :80-99 --
    `pool_cosine` 0: the best location only; 1: mean; p > 1: logits * |logits|^(p-1) (the factor held constant), mean;
    `norm_max_cosine`: divided by the largest |logit| first -- and K must be 1 unless `targets` names the class.

    The encoder passes run on the fused engine (forward, then ONE input-gradient pass from the cotangent of the embedding);
    the cosine-logit chain rule uses bcos_rows_normalize / the GEMM kernel / bcos_cosine_grad.
    Returns dict(logits [N, K] cosine logits (unpool: the pooled logit [N, 1] of the explained class), explained_class_idx,
    embedding, dynamic_linear_weights [N, 6, H, W], contribution_map [N, H, W])."""
    wt_t = text_weights.t().contiguous()                      # [K, D]
    K = wt_t.shape[0]
    info = {}

    def pooled(emb):                                          # [N, D]
        u, inv = ops.rows_normalize(emb.contiguous(), want_y=True, want_inv=True)
        logits = ops.matmul_nt(u, wt_t)                       # [N, K]
        cls = logits.argmax(1) if targets is None else targets.to(device=emb.device, dtype=torch.int64)
        lsel = logits.gather(1, cls.view(-1, 1)).view(-1).contiguous()
        info.update(logits=logits, cls=cls)
        return ops.cosine_grad(u, wt_t.index_select(0, cls).contiguous(), lsel, inv)

    def unpooled(out):                                        # [(HW), N, D'] -> rows ordered (image, location)
        HW, N, D = out.shape
        if targets is None and K != 1:
            raise ValueError("zeroshot_attribution: an attn_unpool head explains ONE text embedding per image "
                             "(text_weights [D, 1], text_localisation.py:58-66, 80) -- or pass `targets`")
        cls = torch.zeros((N,), device=out.device, dtype=torch.int64) if targets is None else \
            targets.to(device=out.device, dtype=torch.int64)
        rows = out.permute(1, 0, 2).contiguous().view(N * HW, D)
        u, inv = ops.rows_normalize(rows, want_y=True, want_inv=True)          # `outa / outa.norm(...)` once more (:76)
        wsel = wt_t.index_select(0, cls.repeat_interleave(HW)).contiguous()    # the image's text embedding at each of its locations
        lg = (ops.matmul_nt(u, wt_t).view(N, HW, K).gather(2, cls.view(N, 1, 1).expand(N, HW, 1))).view(N, HW)
        # pooling weights of the locations, all held constant (`.detach()` in :83-99)
        if pool_cosine == 0:
            # :83-91 -- the best location alone; the mean of :99 then runs over a singleton dimension and `.max(1)` picks the
            # location (no 1 / HW).  With norm_max_cosine the reference divides every masked-out zero by itself: refused.
            if norm_max_cosine:
                raise ValueError("zeroshot_attribution: pool_cosine=0 with norm_max_cosine is 0 / 0 in the reference (:92-93)")
            coef = torch.zeros_like(lg).scatter_(1, lg.argmax(1, keepdim=True), 1.0)
            value = (lg * coef).sum(1, keepdim=True)
        else:
            c, eff = torch.ones_like(lg), lg
            if norm_max_cosine:
                mx = eff.abs().max(1, keepdim=True).values
                c, eff = c / mx, eff / mx
            if pool_cosine > 1:
                c = c * eff.abs() ** (pool_cosine - 1)
            coef = c / HW
            value = (lg * coef).sum(1, keepdim=True)
        coef = coef.reshape(-1).contiguous()
        info.update(logits=value, cls=cls)
        g = ops.cosine_grad(u, wsel, lg.reshape(-1).contiguous(), inv, coef)
        return g.view(N, HW, D).permute(1, 0, 2)

    head = unpooled if engine.head_kind == "attn_unpool" else pooled
    out = engine.explain(images, want_weights=want_weights, gates=gates, cotangent=head)
    return dict(logits=info["logits"], explained_class_idx=info["cls"], embedding=out["embedding"],
                dynamic_linear_weights=out["dynamic_linear_weights"], contribution_map=out["contribution_map"])
