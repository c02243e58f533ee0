#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "vit or gelu or layers_against_golden" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --arch vit_ti --batch 512 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('vit', r['value'], r['step_times']['all_steps']['median'], r['roofline'].get('sclk_mhz_mean'))"; done
