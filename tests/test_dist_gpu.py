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
    assert v["overlapped_equals_gather"]
    assert v["sharded_equals_unsharded"], v                       # same bits as one pass over all 1024 images
    assert v["rel_logits_vs_oracle"] <= 1e-4 and v["classes_equal_oracle"], v
    assert v["rel_maps_vs_oracle"] <= 3e-3, v                     # free ReLU gates: the ResNet-50 map floor (H1)


@pytest.mark.gpu
def test_config4_clip_rn50_global2048_sharded_over_8_ranks(tmp_path):
    v = _launch("clip", 2048, tmp_path)
    assert v["gathered_shape"] == [2048, 1024] and v["maps_shape"] == [2048, 224, 224] and v["finite"]
    assert v["shards"] == [[256 * r, 256 * (r + 1)] for r in range(8)]
    assert v["sharded_equals_unsharded"] and v["maps_equal_small_batch"], v
    assert v["rel_emb_vs_oracle"] <= 1e-4 and v["rel_logits_vs_oracle"] <= 1e-4, v


@pytest.mark.gpu
def test_attn_unpool_outputs_gather_along_batch_dim(tmp_path):
    v = _launch("unpool", 19, tmp_path, world=4)                  # ragged: 5 + 5 + 5 + 4 images
    assert v["gathered_shape"][1] == 19 and v["shards"] == [[0, 5], [5, 10], [10, 15], [15, 19]]
    assert v["sharded_equals_unsharded"] and v["rel_vs_oracle"] <= 1e-5, v
