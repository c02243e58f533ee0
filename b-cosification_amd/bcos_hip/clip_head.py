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
