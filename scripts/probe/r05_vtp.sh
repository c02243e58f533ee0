#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train or fused_batchnorm or clip_modified or scale_derivative" 2>&1 | tail -3
for rep in 1 2 3; do
for a in resnet50 resnet18 clip_rn50; do
python bench.py --train --arch $a --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$a', r['value'], r['ms_per_step'])"
done
done
