#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05/train_prof
python -m pytest tests/test_gpu_parity.py -x -q -k "training_step" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/train_prof -- python3 bench.py --train --arch resnet50 --steps 5 --warmup 2 > gpurun_out/r05/train_prof.json 2> gpurun_out/r05/train_prof.err
find gpurun_out/r05/train_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05/train_kernel_stats.csv
rm -rf gpurun_out/r05/train_prof
cat gpurun_out/r05/train_prof.json | head -c 300
