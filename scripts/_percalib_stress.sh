#!/bin/bash
# 8 ranks on the one device, ROUNDS rounds of build + self-calibration + digest exchange (tests/dist_gpu_worker.py: percalib_stress)
ROUNDS=${ROUNDS:-60}
PORT=$((20000 + RANDOM % 20000))
BCOS_DIST_BACKEND=gloo OMP_NUM_THREADS=8 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout ${TMO:-2400} python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 \
  --master-addr 127.0.0.1 --master-port $PORT tests/dist_gpu_worker.py percalib_stress $ROUNDS gpurun_out/r4_percalib_stress.json > gpurun_out/r4_percalib_stress.log 2>&1
echo "rc $?" >> gpurun_out/r4_percalib_stress.log
