#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train or fused_batchnorm or clip_modified" 2>&1 | tail -3
python scripts/probe/train_host_probe.py vit_ti 2>&1 | grep issue
python scripts/probe/train_host_probe.py resnet50 2>&1 | grep issue
for rep in 1 2; do
for a in vit_ti resnet50 resnet18 clip_rn50; do
for s in 1 0; do
BCOS_TRAIN_SIDE_STREAM=$s python bench.py --train --arch $a --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$a side=$s', r['value'], r['ms_per_step'])"
done
done
done
