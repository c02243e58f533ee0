"""`-m gpu`: the data-parallel path at the FULL global sizes of BASELINE.json configs[3] and configs[4], with 8 real
processes that all use the one device of the test box (gloo transport instead of RCCL: the product's sharding, packing
and gather calls are the ones bench.py uses for N > 1; what a 1-GPU box cannot show is the xGMI transport itself).
Each rank runs the fused engine on its contiguous shard in device memory; rank 0 checks the gathered result against the
unsharded run (bit for bit) and a sub-batch against the CPU oracle.  Workers are fresh child processes
(tests/dist_gpu_worker.py under torch.distributed.run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORLD = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(config, n_global, tmp_path, world=WORLD, timeout=1500):
    out = tmp_path / f"{config}.json"
    env = dict(os.environ, BCOS_DIST_BACKEND="gloo", OMP_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "dist_gpu_worker.py"), config, str(n_global), str(out)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if proc.returncode != 0 and ("address already in use" in (proc.stdout + proc.stderr).lower() or "EADDRINUSE" in proc.stderr):
        # the probed rendezvous port was taken between the probe and torchrun's bind: once more on a fresh port
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    return json.load(open(out))


@pytest.mark.gpu
def test_config5_resnet50_global1024_sharded_over_8_ranks(tmp_path):
    v = _launch("r50", 1024, tmp_path)
    assert v["gathered_shape"] == [1024, 224, 224] and v["shards"] == [[128 * r, 128 * (r + 1)] for r in range(8)]
    assert v["overlapped_equals_gather"] and v["replicas_identical"], json.dumps(v)
    assert v["sharded_equals_unsharded"], json.dumps(v)                       # same bits as one pass over all 1024 images
    assert v["rel_logits_vs_oracle"] <= 1e-4 and v["classes_equal_oracle"], v
    assert v["rel_maps_vs_oracle"] <= 3e-3, v                     # free ReLU gates: the ResNet-50 map floor (H1)
    assert v["rel_weights_pinned_vs_oracle"] <= 1e-4 and v["rel_maps_pinned_vs_oracle"] <= 1e-4, v     # the oracle's gates replayed
    assert v["sub_batch_maps_equal"], v


@pytest.mark.gpu
def test_config4_clip_rn50_global2048_sharded_over_8_ranks(tmp_path):
    v = _launch("clip", 2048, tmp_path)
    assert v["gathered_shape"] == [2048, 1024] and v["maps_shape"] == [2048, 224, 224] and v["finite"]
    assert v["shards"] == [[256 * r, 256 * (r + 1)] for r in range(8)]
    assert v["replicas_identical"], json.dumps(v)
    assert v["sharded_equals_unsharded"] and v["maps_equal_small_batch"], json.dumps(v)
    assert v["rel_emb_vs_oracle"] <= 1e-4 and v["rel_logits_vs_oracle"] <= 1e-4, v


@pytest.mark.gpu
def test_attn_unpool_outputs_gather_along_batch_dim(tmp_path):
    v = _launch("unpool", 19, tmp_path, world=4)                  # ragged: 5 + 5 + 5 + 4 images
    assert v["gathered_shape"][1] == 19 and v["shards"] == [[0, 5], [5, 10], [10, 15], [15, 19]]
    assert v["sharded_equals_unsharded"] and v["rel_vs_oracle"] <= 1e-5, v


@pytest.mark.gpu
def test_independently_calibrated_replicas_agree(tmp_path):
    """8 ranks time-slicing ONE device, each building and calibrating its own replica (3 passes, no parameter broadcast): bit-identical
    state dicts.  Rounds 2-3 carried this as a non-strict xfail: about one process in a hundred ended with different BatchNorm
    statistics.  Round 4 traced it to torch's multi-block `var` reduction under GPU time-slicing -- 26 grossly wrong results
    (16 channels each, 12-85 % off) in 425 600 reductions of bit-identical, settled inputs behind a device synchronisation, against
    0 of 212 800 for this repo's fixed-order kernel (profiles/r04_var_triage.txt) -- and synth.calibrate now derives every statistic
    with this repo's fixed-order reductions -- ops.channel_moments_ordered -> ops.colsum = bcos_colsum_ws: per-workgroup partials added in
    workgroup order, the kernel the 3 200 process-calibrations of profiles/r04_percalib_stress.json ran (the triage measured its
    single-pass sibling bcos_colsum_ordered) --: the test is strict."""
    v = _launch("percalib", 3, tmp_path)          # 8 ranks x 3 calibration passes each, no parameter broadcast
    assert v["replicas_identical"], json.dumps(v)


@pytest.mark.gpu
def test_bench_two_ranks_self_launched():
    """`python bench.py --gpus 2 ...` as the driver types it for N = 1 (no torchrun around it): the parent starts the two ranks
    as a child process group before touching the GPU, the ranks exchange rank 0's parameters + digests, run the step with
    the packed all-gather and rank 0's JSON line comes back through the parent.  Both ranks share the one device of the
    test box (BCOS_SINGLE_DEVICE, gloo transport)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BCOS_SINGLE_DEVICE="1", BCOS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--batch", "32"], env=env, capture_output=True, text=True, timeout=1200)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["ranks_seen"] == 2 and res["config"]["backend"] == "gloo"
    assert res["config"]["replicas_identical"] is True and res["config"]["global_batch"] == 64
    assert res["value"] > 0 and res["steps"] == 2 and res["cpu_baseline"] is None


@pytest.mark.gpu
def test_rccl_backend_calls_single_rank(tmp_path):
    """The collectives the product issues for N > 1 -- asynchronous all_gather_into_tensor on a flat fp32 buffer
    (OverlappedGather), asynchronous all_reduce SUM (allreduce_gradients), barrier, all_reduce MAX on a float64 scalar
    (bench.py's max-over-ranks timing) -- run through the "nccl" (= RCCL) backend in a fresh child process.  A 1-GPU box
    gives world size 1: this checks that RCCL initialises in this environment and accepts exactly these calls and dtypes,
    not the xGMI transport."""
    code = r'''
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
send = torch.arange(1000, device="cuda", dtype=torch.float32)
recv = torch.empty((1, 1000), device="cuda", dtype=torch.float32)
w = dist.all_gather_into_tensor(recv.view(-1), send, async_op=True); w.wait()
assert torch.equal(recv[0], send)
g = torch.ones(1 << 20, device="cuda"); w = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True); w.wait()
assert float(g.sum()) == float(1 << 20)
t = torch.tensor([1.25], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
assert float(t) == 1.25
dist.destroy_process_group()
print("rccl ok")
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-c", code, str(_free_port())], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "rccl ok" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]
