"""Worker of tests/test_dist_gpu.py: one rank of an N-process data-parallel run whose ranks all sit on cuda:0.

Launched through `python -m torch.distributed.run --nproc-per-node N tests/dist_gpu_worker.py <config> <out.json>` with
the gloo backend (a 1-GPU box cannot run RCCL across ranks; the collective *calls* are the product's --
bcos_hip.dist.explain_sharded / all_gather_batch / OverlappedGather -- only the transport differs from the 8-GPU run).
Rank 0 writes the verdict as JSON.  TEST INFRASTRUCTURE: the oracle is imported here as the checker only.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (os.path.join(REPO, "b-cosification_amd"), REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bcos_hip import dist as bdist, engine, synth  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def free_others(rank):
    """ranks > 0 drop their device memory before rank 0 runs the whole batch unsharded"""
    dist.barrier()
    if rank != 0:
        torch.cuda.empty_cache()
    dist.barrier()


def replicate(net, world):
    """Data-parallel replicas carry rank 0's parameters (what loading one checkpoint / a broadcast gives in deployment); the
    ranks then prove it: a digest of every state-dict entry is exchanged and compared (bcos_hip.dist.replicate_parameters,
    the call bench.py makes for N > 1).
    Eight processes time-slicing ONE device are not the deployment mode, and independent per-rank calibrations were seen to
    diverge under it about once in a hundred processes: torch's multi-block `var` reduction returns wrong values for a few
    channels of an identical input (scripts/probe/layer3_stress2.py)."""
    return bdist.replicate_parameters(net)


def _oracle_pre_activations(sd, x_cpu):
    from oracle import bcos_oracle as O
    log = []
    with torch.no_grad():
        O.resnet_logits(sd, x_cpu, "resnet50", detach=True, gate_log=log)
    return log


def run_resnet50(rank, world, n_global):
    """BASELINE configs[4]: ResNet-50 explanation maps, global batch 1024 = 8 x 128, logits + maps gathered."""
    from oracle import bcos_oracle as O
    net = synth.build_bcosified_resnet("resnet50").to(DEV)
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to(DEV))
        replica_diff = replicate(net, world)
    eng = engine.attach(net)
    x = synth.synthetic_images(n_global, seed=4321).to(DEV)
    res = bdist.explain_sharded(eng, x, gather=("logits", "contribution_map", "prediction"), num_outputs=1000)
    lo, hi = res["shard"]
    # bench.py's form of the same exchange: one packed asynchronous all-gather per step, double buffered
    pipe = bdist.OverlappedGather(depth=2)
    mine = eng.explain(x[lo:hi], want_weights=False)
    pipe.submit({"logits": mine["logits"], "contribution_map": mine["contribution_map"]})
    packed = pipe.flush()[0]
    verdict = dict(config="resnet50", world=world, n_global=n_global, shard=[lo, hi], replicas_identical=not replica_diff,
                   gathered_shape=list(res["contribution_map"].shape),
                   overlapped_equals_gather=bool(torch.equal(packed["logits"], res["logits"])
                                                 and torch.equal(packed["contribution_map"], res["contribution_map"])),
                   overlapped_vs_gather=[rel(packed["logits"], res["logits"]), rel(packed["contribution_map"], res["contribution_map"]),
                                         (packed["contribution_map"] != res["contribution_map"]).flatten(1).any(1).nonzero().flatten().tolist()[:16]])
    del mine, packed, pipe
    free_others(rank)
    if rank == 0:
        full = eng.explain(x, want_weights=False)                          # unsharded: the whole global batch in one pass
        verdict.update(sharded_equals_unsharded=bool(torch.equal(full["logits"], res["logits"])
                                                     and torch.equal(full["contribution_map"], res["contribution_map"])
                                                     and torch.equal(full["prediction"], res["prediction"])),
                       rel_logits_vs_unsharded=rel(res["logits"], full["logits"]),
                       rel_maps_vs_unsharded=rel(res["contribution_map"], full["contribution_map"]))
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        idx = [0, n_global // 2 + 3, n_global // 2 + 3 + n_global // world, n_global - 1]     # images of four different ranks' shards
        ref = O.explain_batch(lambda xx, detach: O.resnet_logits(sd, xx, "resnet50", detach=detach), x[idx].cpu())
        verdict.update(rel_logits_vs_oracle=rel(res["logits"][idx], ref["logits"]),
                       classes_equal_oracle=bool(torch.equal(res["prediction"][idx].cpu(), ref["prediction"])),
                       rel_maps_vs_oracle=rel(res["contribution_map"][idx], ref["contribution_map"]))
        # the same images with the oracle's ReLU decisions replayed: W(x) and maps hold 1e-4; and a sub-batch
        # reproduces its rows of the gathered maps bit for bit
        gates = [(p > 0).float().permute(0, 2, 3, 1).contiguous().to(DEV) for p in _oracle_pre_activations(sd, x[idx].cpu())]
        pinned = eng.explain(x[idx], gates=gates)
        sub = eng.explain(x[idx[1]:idx[1] + 2], want_weights=False)
        verdict.update(rel_weights_pinned_vs_oracle=rel(pinned["dynamic_linear_weights"], ref["dynamic_linear_weights"]),
                       rel_maps_pinned_vs_oracle=rel(pinned["contribution_map"], ref["contribution_map"]),
                       sub_batch_maps_equal=bool(torch.equal(sub["contribution_map"], res["contribution_map"][idx[1]:idx[1] + 2])))
    return verdict


def run_clip(rank, world, n_global):
    """BASELINE configs[3]: CLIP RN50 image encoder, global batch 2048 = 8 x 256, embeddings gathered along the batch dim;
    plus the explanation of one embedding coordinate at the per-rank shard size."""
    from oracle import bcos_oracle as O
    from bcos_hip import clip_head
    net = synth.build_bcosified_clip_rn50().to(DEV)
    with torch.no_grad():
        synth.calibrate(net, synth.synthetic_images(8).to(DEV))
    with torch.no_grad():
        replica_diff = replicate(net, world)
    eng = engine.attach(net)
    x = synth.synthetic_images(n_global, seed=99).to(DEV)
    lo, hi = bdist.shard_bounds(n_global, rank, world)
    emb = eng.forward(x[lo:hi])
    gathered = bdist.all_gather_batch(emb, dim=0)                          # [n_global, 1024] on every rank
    wt = torch.randn(1024, 1000, generator=torch.Generator().manual_seed(5)).to(DEV)
    logits = clip_head.zeroshot_logits(gathered, wt)                       # the head after the gather
    # forward + explanation at the full per-rank shard (256): gradient of embedding coordinate 7 of every image
    tg = torch.full((hi - lo,), 7, dtype=torch.int64)
    expl = eng.explain(x[lo:hi], targets=tg, want_weights=False)
    maps = bdist.all_gather_batch(expl["contribution_map"], dim=0)
    verdict = dict(config="clip_rn50", world=world, n_global=n_global, shard=[lo, hi], gathered_shape=list(gathered.shape),
                   maps_shape=list(maps.shape), finite=bool(torch.isfinite(gathered).all() and torch.isfinite(maps).all()),
                   replicas_identical=not replica_diff, replica_diff=replica_diff[:6])
    del expl
    free_others(rank)
    # every rank: is its first forward (the one that built the lazily created weight images) what a repeat gives?
    again1 = eng.forward(x[lo:hi])
    again2 = eng.forward(x[lo:hi])
    verdict.update(first_forward_repeatable=bool(torch.equal(again1, emb)), later_forwards_repeatable=bool(torch.equal(again1, again2)),
                   first_forward_max_abs=float((again1 - emb).abs().max()))
    del again1, again2
    dist.barrier()
    if rank == 0:
        half = n_global // 2
        full = torch.cat([eng.forward(x[:half]), eng.forward(x[half:])])   # a different split of the same batch
        verdict.update(sharded_equals_unsharded=bool(torch.equal(full, gathered)), rel_vs_unsharded=rel(gathered, full))
        if not verdict["sharded_equals_unsharded"]:          # diagnostics: which images differ, and is a repeat of either side stable?
            bad = (full != gathered).any(1).nonzero().flatten().tolist()
            again = torch.cat([eng.forward(x[:half]), eng.forward(x[half:])])
            mine = eng.forward(x[lo:hi])
            verdict.update(mismatch_rows=len(bad), mismatch_first=bad[:8], unsharded_repeatable=bool(torch.equal(again, full)),
                           own_shard_repeatable=bool(torch.equal(mine, gathered[lo:hi])),
                           own_shard_equals_unsharded=bool(torch.equal(mine, full[lo:hi])),
                           max_abs=float((full - gathered).abs().max()))
        one = eng.explain(x[3 * (n_global // world) + 1: 3 * (n_global // world) + 3], targets=torch.tensor([7, 7]), want_weights=False)
        verdict.update(maps_equal_small_batch=bool(torch.equal(one["contribution_map"], maps[3 * (n_global // world) + 1: 3 * (n_global // world) + 3])))
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        idx = [1, n_global - 2]
        ref = O.clip_rn50_embed(sd, x[idx].cpu())
        verdict.update(rel_emb_vs_oracle=rel(gathered[idx], ref), rel_logits_vs_oracle=rel(logits[idx], O.zeroshot_logits(ref, wt.cpu())))
    return verdict


def run_unpool(rank, world, n_global):
    """The attn_unpool head returns [HW, N, D] (bcosattnpool.py:23-32): per-rank outputs are gathered along dim 1."""
    import numpy as np
    from test_host_cpu import _unpool_module
    m, sd, data = _unpool_module(os.path.join(HERE, "golden"))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n_global, *data["x"].shape[1:], generator=g).to(DEV)
    lo, hi = bdist.shard_bounds(n_global, rank, world)
    with torch.no_grad():
        y = m(x[lo:hi])                                                    # [HW, hi-lo, D]
        gathered = bdist.all_gather_batch(y, dim=1)
        full = m(x)
    from oracle import bcos_oracle as O
    ref = O.bcos_attention_unpool(sd, "", x.cpu())
    return dict(config="unpool", world=world, n_global=n_global, shard=[lo, hi], gathered_shape=list(gathered.shape),
                sharded_equals_unsharded=bool(torch.equal(gathered, full)), rel_vs_oracle=rel(gathered, ref))


def run_percalib(rank, world, n_global):
    """Every rank builds AND calibrates its own replica (no broadcast): the replicas agree bit for bit.  (Rounds 2-3: under
    8-process time-slicing of one device about one process in a hundred ended with different BatchNorm statistics -- torch's
    multi-block variance reduction, DESIGN.md section 6 / profiles/r04_var_triage.txt; calibrate now sums in a fixed order with
    this repo's own kernel.)"""
    net = synth.build_bcosified_resnet("resnet50").to(DEV)
    with torch.no_grad():
        for _ in range(max(1, n_global)):
            synth.calibrate(net, synth.synthetic_images(8).to(DEV))
    digest = bdist.state_digest(net)
    all_d = [None] * world
    dist.all_gather_object(all_d, digest)
    diff = [f"rank {r}: {k}" for r in range(world) for k in digest if all_d[r][k] != all_d[0][k]]
    return dict(config="percalib", world=world, shard=[0, 0], replicas_identical=not diff, replica_diff=diff[:8])


def run_percalib_stress(rank, world, n_global):
    """`n_global` ROUNDS of: every rank builds the network from its seed, calibrates it by itself and the ranks compare digests of
    every state-dict entry.  Since round 4 every statistic of synth.calibrate is summed in a fixed order by this repo's own kernel
    (ops.channel_moments_ordered -> bcos_colsum_ws: partials added in workgroup order), so every round must be clean; with torch's multi-block reductions about one process-calibration in a
    hundred differed under this contention (profiles/r04_var_triage.txt)."""
    bad = []
    for rnd in range(max(1, n_global)):
        net = synth.build_bcosified_resnet("resnet50").to(DEV)
        with torch.no_grad():
            synth.calibrate(net, synth.synthetic_images(8).to(DEV))
        digest = bdist.state_digest(net)
        all_d = [None] * world
        dist.all_gather_object(all_d, digest)
        diff = [f"round {rnd} rank {r}: {k}" for r in range(world) for k in digest if all_d[r][k] != all_d[0][k]]
        if diff:
            # the first entry (state-dict order) on which the ranks disagree, and how they group on it
            first = next(k for k in digest if any(all_d[r][k] != all_d[0][k] for r in range(world)))
            groups = {}
            for r in range(world):
                groups.setdefault(repr(all_d[r][first]), []).append(r)
            bad.append(dict(round=rnd, first_key=first, groups=sorted(groups.values()), n_keys=len({d_.split(": ")[1] for d_ in diff})))
        del net
    return dict(config="percalib_stress", world=world, shard=[0, 0], rounds=max(1, n_global), clean_rounds=max(1, n_global) - len(bad),
                replicas_identical=not bad, replica_diff=bad[:8])


def main():
    config, n_global, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, _, world = bdist.init(backend="gloo")
    torch.cuda.set_device(0)
    t0 = time.time()
    verdict = {"r50": run_resnet50, "clip": run_clip, "unpool": run_unpool, "percalib": run_percalib, "percalib_stress": run_percalib_stress}[config](rank, world, n_global)
    verdict["seconds"] = round(time.time() - t0, 1)
    all_v = [None] * world
    dist.all_gather_object(all_v, verdict)
    if rank == 0:
        verdict["shards"] = [v["shard"] for v in all_v]
        verdict["per_rank"] = [{k: v[k] for k in ("first_forward_repeatable", "later_forwards_repeatable", "first_forward_max_abs") if k in v}
                               for v in all_v]
        with open(out_path, "w") as f:
            json.dump(verdict, f, indent=1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
