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
            backend = os.environ.get("BCOS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def state_digest(module) -> dict:
    """{state-dict key: hash of the entry's exact bytes}, taken on the HOST.  (Not a device-side checksum: torch's multi-block
    reductions were measured to return wrong sums now and then when several processes time-slice one GPU -- DESIGN.md section 6 --
    and a digest that can be wrong by itself proves nothing about the replicas.)"""
    import hashlib
    out = {}
    for k, v in module.state_dict().items():
        t = v.detach().cpu().contiguous()
        out[k] = hashlib.blake2b(t.numpy().tobytes() if t.dtype != torch.bfloat16 else t.view(torch.int16).numpy().tobytes(),
                                 digest_size=16).hexdigest()
    return out


def replicate_parameters(module) -> list:
    """Give every rank rank 0's parameters and buffers (one broadcast per state-dict entry -- what loading one checkpoint
    on every rank gives in deployment), then prove the replicas identical: each rank's digest of every entry is exchanged
    and compared.  Returns the list of differing entries ("rank r: key"); empty = identical.  No-op for world 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    world = dist.get_world_size()
    on_cpu = dist.get_backend() == "gloo"
    for v in module.state_dict().values():
        t = v.detach()
        if on_cpu:
            t = t.cpu()
        elif not t.is_contiguous():
            t = t.contiguous()
        dist.broadcast(t, src=0)
        if t.device != v.device or t.data_ptr() != v.data_ptr():
            v.copy_(t.to(v.device))
    digest = state_digest(module)
    all_d = [None] * world
    dist.all_gather_object(all_d, digest)
    return [f"rank {r}: {k}" for r in range(world) for k in digest if all_d[r][k] != all_d[0][k]]


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


def all_gather_batch(t: torch.Tensor, dim: int = 0, counts=None) -> torch.Tensor:
    """all_gather_rows along an arbitrary batch dimension: the CLIP `attn_unpool` head returns [HW, N, D] -- batch is
    dim 1 (bcosattnpool.py:23-32) -- so its shards are concatenated along that dimension, not along dim 0."""
    if dim == 0:
        return all_gather_rows(t, counts)
    return all_gather_rows(t.movedim(dim, 0).contiguous(), counts).movedim(0, dim)


_OUT_TAILS = {"logits": lambda x, k: (k,), "contribution_map": lambda x, k: tuple(x.shape[2:]),
              "dynamic_linear_weights": lambda x, k: (6,) + tuple(x.shape[2:]), "prediction": lambda x, k: (),
              "explained_class_idx": lambda x, k: ()}
_OUT_DTYPES = {"prediction": torch.int64, "explained_class_idx": torch.int64}


def explain_sharded(engine, images: torch.Tensor, targets: Optional[torch.Tensor] = None, gather=("logits", "contribution_map"),
                    want_weights: bool = False, num_outputs: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Run `engine.explain` on this rank's shard of `images` (the full batch, identical on every rank) and
    all-gather the requested outputs.  A rank whose shard is empty (fewer images than ranks: the last ragged batch of an
    evaluation) skips the kernels and contributes zero-row tensors, so every rank still reaches the collective;
    `num_outputs` (the logit count) is only needed for that case when 'logits' is gathered."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    n = images.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    counts = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
    # Everything that decides WHICH collectives are issued is derived from arguments every rank holds alike -- never from
    # what this rank's engine happened to return -- and is validated BEFORE any rank enters a collective: a rank that raised
    # alone (or skipped a key the others gather) would leave the rest hanging in all_gather_into_tensor.
    unknown = [k for k in gather if k not in _OUT_TAILS]
    if unknown:
        raise KeyError(f"explain_sharded: cannot gather {unknown}; known outputs: {sorted(_OUT_TAILS)}")
    if "dynamic_linear_weights" in gather and not want_weights:
        raise ValueError("explain_sharded: gathering 'dynamic_linear_weights' needs want_weights=True")
    if min(counts) == 0 and "logits" in gather and num_outputs is None:
        raise ValueError("explain_sharded: fewer images than ranks: a rank without images needs num_outputs to shape its "
                         "empty logits (raised on every rank)")
    if hi > lo:
        out = engine.explain(images[lo:hi], None if targets is None else targets[lo:hi], want_weights=want_weights)
        missing = [k for k in gather if out.get(k) is None]
        if missing:         # an engine that does not produce a requested output: a bug on every non-empty rank alike
            raise KeyError(f"explain_sharded: engine.explain returned no {missing}")
    else:
        out = {k: torch.empty((0,) + _OUT_TAILS[k](images, num_outputs), device=images.device, dtype=_OUT_DTYPES.get(k, torch.float32))
               for k in gather}
    res = dict(out)
    for k in gather:
        res[k] = all_gather_rows(out[k], counts)
    res["shard"] = (lo, hi)
    return res


class OverlappedGather:
    """The path's one collective, taken off the critical path: every step's per-rank results (e.g. logits and
    contribution maps) are packed into ONE flat buffer, all-gathered with a single asynchronous
    `all_gather_into_tensor` (RCCL runs it on its own stream over xGMI) and only waited for when the slot is needed
    again -- so the exchange of step i overlaps the compute of step i+1.  `depth` slots are cycled (2 = double buffer).

        pipe = OverlappedGather(depth=2)
        for batch in batches:
            out = engine.explain(batch)
            done = pipe.submit({"logits": out["logits"], "contribution_map": out["contribution_map"]})
            ...                      # `done` = gathered tensors of the step submitted `depth` steps ago (or None)
        rest = pipe.flush()          # list of the gathered dicts still in flight, oldest first

    Every rank must submit tensors of identical shapes (equal shards).  With world size 1 it is a pass-through."""

    def __init__(self, depth: int = 2):
        self.depth = max(1, int(depth))
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._slots = [None] * self.depth          # (send, recv, work, layout)
        self._next = 0

    def _finish(self, slot):
        send, recv, work, layout = slot
        if work is not None:
            work.wait()                            # NCCL: makes the current stream wait; no host block
        out, off = {}, 0
        for name, shape, numel in layout:
            out[name] = recv[:, off:off + numel].reshape((self.world * shape[0],) + tuple(shape[1:]))
            off += numel
        return out

    def submit(self, tensors: Dict[str, torch.Tensor], copy_out: bool = True) -> Optional[Dict[str, torch.Tensor]]:
        """Queue this step's tensors; returns the gathered result of the step that used this slot `depth` submissions
        ago (None while the pipeline fills).  `copy_out=False` returns views into the slot's receive buffer, which the
        exchange started by THIS call overwrites -- only for callers that drop the return value."""
        if self.world == 1:
            return dict(tensors)
        i = self._next
        self._next = (i + 1) % self.depth
        prev = self._slots[i]
        done = self._finish(prev) if prev is not None else None
        if done is not None and copy_out:          # the slot's buffers are reused below: hand out copies on request
            done = {k: v.clone() for k, v in done.items()}
        for k, v in tensors.items():               # one packed fp32 buffer: an int64 tensor would be silently rounded
            if v.dtype != torch.float32:
                raise TypeError(f"OverlappedGather packs float32 tensors only ('{k}' is {v.dtype}); gather it separately")
        layout = [(k, tuple(v.shape), v.numel()) for k, v in tensors.items()]
        total = sum(n for _, _, n in layout)
        first = next(iter(tensors.values()))
        if prev is not None and prev[0].numel() == total and prev[0].device == first.device:
            send, recv = prev[0], prev[1]
        else:
            send = torch.empty((total,), device=first.device, dtype=torch.float32)
            recv = torch.empty((self.world, total), device=first.device, dtype=torch.float32)
        off = 0
        for (_, _, n), v in zip(layout, tensors.values()):
            send[off:off + n].copy_(v.reshape(-1))
            off += n
        work = dist.all_gather_into_tensor(recv.view(-1), send, async_op=True)
        self._slots[i] = (send, recv, work, layout)
        return done

    def flush(self):
        """Wait for everything in flight; returns the gathered dicts oldest first and empties the pipeline."""
        outs = []
        for k in range(self.depth):
            i = (self._next + k) % self.depth
            if self._slots[i] is not None:
                outs.append(self._finish(self._slots[i]))
                self._slots[i] = None
        return outs


def allreduce_gradients(parameters, bucket_bytes: int = 64 << 20, average: bool = True):
    """Data-parallel training: sum (or average) the `.grad` of every parameter over the ranks.
    What the reference gets from Lightning's DDP strategy (bcos/training/trainer.py:916-918) is done here explicitly:
    gradients are packed into flat fp32 buckets of ~`bucket_bytes` (few large collectives: the xGMI links are point to
    point, a ring all-reduce is bound per link), every bucket is all-reduced asynchronously on RCCL's stream (backend
    "nccl" on ROCm; "gloo" in the CPU tests) while the next one is being packed, and unpacked in order afterwards.
    Parameters without a gradient are skipped on every rank alike (the same model runs everywhere).  No-op for world 1."""
    params = [p for p in parameters if p.grad is not None]
    if not dist.is_initialized() or dist.get_world_size() == 1 or not params:
        return
    world = dist.get_world_size()
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.grad.numel() * 4
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    inflight = []
    for bucket in buckets:
        flat = torch.cat([p.grad.detach().reshape(-1).float() for p in bucket])
        inflight.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
    for bucket, flat, work in inflight:
        work.wait()
        if average:
            flat.div_(world)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
