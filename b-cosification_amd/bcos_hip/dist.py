"""Data-parallel sharding of the explanation path: one process per GPU, independent images.

Every image's forward and explanation is independent in eval mode, so a batch is split
into contiguous per-rank shards, each rank runs the fused engine on its shard with replicated weights, and ONE
collective at the end -- an all-gather over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" on CPU
for tests) -- assembles per-rank logits / maps on every rank.  No collective sits on the data path itself.
"""
import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None):
    """Initialise the default process group from the environment if WORLD_SIZE > 1."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `n` items for `rank`; the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(t: torch.Tensor, counts=None) -> torch.Tensor:
    """Concatenate per-rank tensors along dim 0 on every rank with a single all-gather.  Equal shard sizes use
    all_gather_into_tensor (one flat buffer, the direct one-hop exchange on the fully connected xGMI mesh);
    ragged shards are padded to the largest one."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    world = dist.get_world_size()
    if counts is None:
        cnt = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
        all_cnt = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(all_cnt, cnt)
        counts = [int(c.item()) for c in all_cnt]
    mx = max(counts)
    if all(c == mx for c in counts):
        out = torch.empty((world * mx,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        dist.all_gather_into_tensor(out, t.contiguous())
        return out
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    pad[: t.shape[0]] = t
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)], 0)


def explain_sharded(engine, images: torch.Tensor, targets: Optional[torch.Tensor] = None, gather=("logits", "contribution_map"),
                    want_weights: bool = False) -> Dict[str, torch.Tensor]:
    """Run `engine.explain` on this rank's shard of `images` (the full batch, identical on every rank) and
    all-gather the requested outputs."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_bounds(images.shape[0], rank, world)
    out = engine.explain(images[lo:hi], None if targets is None else targets[lo:hi], want_weights=want_weights)
    counts = [shard_bounds(images.shape[0], r, world)[1] - shard_bounds(images.shape[0], r, world)[0] for r in range(world)]
    res = dict(out)
    for k in gather:
        if out.get(k) is not None:
            res[k] = all_gather_rows(out[k], counts)
    res["shard"] = (lo, hi)
    return res
