#!/bin/bash
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_batchnorm_training or training_step or train" 2>&1 | tail -5
for a in resnet50 resnet18; do
python bench.py --train --arch $a --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$a', r['value'], r['ms_per_step'], r['step_times'])"
done
