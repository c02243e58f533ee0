"""`-m "not gpu"`: the data-parallel path with world_size 2 on the gloo backend (CPU), kernels emulated.
Checks that contiguous sharding + ONE all-gather reproduces the single-process result on every rank, for equal
and ragged shard sizes."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, q):
    repo = os.path.dirname(HERE)
    for p in (os.path.join(repo, "b-cosification_amd"), repo, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import cpu_emulation
    cpu_emulation.install_permanent()
    from bcos_hip import dist as bdist, engine, synth
    bdist.init(backend="gloo")
    net = synth.build_bcosified_resnet("resnet18")
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "linear") and isinstance(m.linear, torch.nn.Conv2d):
                m.linear.weight.mul_(3.0)
    eng = engine.ResNetEngine(net)
    x = synth.synthetic_images(n_images, size=32)
    res = bdist.explain_sharded(eng, x, gather=("logits", "contribution_map", "prediction"), num_outputs=1000)
    full = eng.explain(x)
    ok = (torch.allclose(res["logits"], full["logits"], rtol=1e-5, atol=1e-6)
          and torch.allclose(res["contribution_map"], full["contribution_map"], rtol=1e-4, atol=1e-7)
          and torch.equal(res["prediction"], full["prediction"])
          and res["logits"].shape[0] == n_images)
    # batch along dim 1 (the CLIP attn_unpool head's [HW, N, D]): gathered along that dimension
    lo, hi = res["shard"]
    t = torch.arange(5 * n_images * 3, dtype=torch.float32).view(5, n_images, 3)
    counts = [bdist.shard_bounds(n_images, r, world)[1] - bdist.shard_bounds(n_images, r, world)[0] for r in range(world)]
    ok = ok and torch.equal(bdist.all_gather_batch(t[:, lo:hi].contiguous(), dim=1, counts=counts), t)
    # the collectives issued depend on the arguments only, and bad requests are refused on EVERY rank before any collective
    # (a rank raising alone would leave the others hanging in all_gather_into_tensor) -- the empty-shard rank included
    w = bdist.explain_sharded(eng, x, gather=("dynamic_linear_weights", "contribution_map"), want_weights=True)
    ok = ok and torch.allclose(w["dynamic_linear_weights"], full["dynamic_linear_weights"], rtol=1e-4, atol=1e-7)
    ok = ok and w["dynamic_linear_weights"].shape[0] == n_images
    refused = 0
    for kw, exc in ((dict(gather=("dynamic_linear_weights",), want_weights=False), ValueError),
                    (dict(gather=("no_such_output",)), KeyError)) + (
                   ((dict(gather=("logits",)), ValueError),) if n_images < world else ()):
        try:
            bdist.explain_sharded(eng, x, **kw)
        except exc:
            refused += 1
    ok = ok and refused == (3 if n_images < world else 2)
    # replicas: rank 1 perturbs its parameters, replicate_parameters() restores rank 0's and the digests agree
    if rank == 1:
        with torch.no_grad():
            next(net.parameters()).add_(1.0)
    ok = ok and bdist.replicate_parameters(net) == []
    ref = [torch.zeros_like(next(net.parameters())) for _ in range(world)]
    dist.all_gather(ref, next(net.parameters()).detach())
    ok = ok and torch.equal(ref[0], ref[1])
    q.put((rank, bool(ok), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [4, 3, 1])          # equal shards, ragged shards, fewer images than ranks
def test_sharded_explanation_world2_gloo(n_images):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert all(ok for _, ok, _, _ in results), results
    assert results[0][2] == 0 and results[0][3] == results[1][2] and results[1][3] == n_images


def _overlap_worker(rank, world, port, q):
    repo = os.path.dirname(HERE)
    for p in (os.path.join(repo, "b-cosification_amd"), repo, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from bcos_hip import dist as bdist
    bdist.init(backend="gloo")
    pipe = bdist.OverlappedGather(depth=2)
    steps, got = 5, []
    for i in range(steps):            # rank r, step i contributes logits filled with 100 i + r and maps with -(100 i + r)
        out = {"logits": torch.full((3, 7), float(100 * i + rank)), "contribution_map": torch.full((3, 4, 5), -float(100 * i + rank))}
        done = pipe.submit(out)
        if done is not None:
            got.append(done)
    got += pipe.flush()
    ok = len(got) == steps
    for i, d in enumerate(got):
        exp_l = torch.cat([torch.full((3, 7), float(100 * i + r)) for r in range(world)])
        exp_m = torch.cat([torch.full((3, 4, 5), -float(100 * i + r)) for r in range(world)])
        ok = ok and torch.equal(d["logits"], exp_l) and torch.equal(d["contribution_map"], exp_m)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_world2_gloo():
    """bench.py's N > 1 collective: one packed asynchronous all-gather per step, double buffered, results in order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def _allreduce_worker(rank, world, port, q):
    repo = os.path.dirname(HERE)
    for p in (os.path.join(repo, "b-cosification_amd"), repo, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from bcos_hip import dist as bdist
    bdist.init(backend="gloo")
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 6, 7, 7), (64,), (128, 64, 3, 3), (1000, 512, 1, 1), (3,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes] + [torch.nn.Parameter(torch.zeros(4))]   # the last has no grad
    base = [torch.randn(s, generator=g) for s in shapes]
    for p, b in zip(params, base):
        p.grad = b * (rank + 1)                      # rank r holds (r + 1) * base
    bdist.allreduce_gradients(params, bucket_bytes=256 << 10)        # several buckets
    ok = params[-1].grad is None
    for p, b in zip(params, base):
        ok = ok and torch.allclose(p.grad, b * (sum(range(1, world + 1)) / world), rtol=1e-6, atol=1e-7)
    for p, b in zip(params, base):
        p.grad = b * (rank + 1)
    bdist.allreduce_gradients(params, average=False)                  # one bucket, plain sum
    for p, b in zip(params, base):
        ok = ok and torch.allclose(p.grad, b * sum(range(1, world + 1)), rtol=1e-6, atol=1e-7)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    """N4: bucketed asynchronous gradient all-reduce (the collective of data-parallel training)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def test_bench_self_launches_ranks_as_children():
    """`python bench.py --gpus 2` without a torchrun environment starts the ranks itself as a child process group and relays
    their exit code; on this GPU-less box the ranks stop at "needs a HIP device" -- loudly, in the children, while the parent
    (which never touches the GPU and never execs) reports the failure."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_dist_gpu.py::test_bench_two_ranks_self_launched")
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                           "--batch", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode != 0
    assert "needs a HIP device" in proc.stderr and "torch.distributed" in proc.stderr, proc.stderr[-2000:]
