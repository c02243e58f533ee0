#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train or fused_batchnorm or clip_modified or layernorm or vit" 2>&1 | tail -3
python scripts/probe/train_host_probe.py vit_ti 2>&1 | grep issue
for rep in 1 2; do
for a in vit_ti resnet50 resnet18 clip_rn50; do
python bench.py --train --arch $a --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$a', r['value'], r['ms_per_step'])"
done
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('r50 infer', r['value'], r['ms_per_step'], r['roofline'].get('sclk_mhz_mean'))"
python bench.py --arch vit_ti --batch 512 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('vit infer', r['value'], r['ms_per_step'], r['roofline'].get('sclk_mhz_mean'))"
