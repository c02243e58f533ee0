#!/bin/bash
mkdir -p gpurun_out/r05
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for a in resnet50 resnet18 vit_ti; do
python bench.py --train --arch $a --steps 10 --warmup 3 2>/dev/null > gpurun_out/r05/bench_train_$a.json
python -c "import json; r=json.load(open('gpurun_out/r05/bench_train_$a.json')); print('$a', r['value'], r['ms_per_step'])"
done
