#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "attention_gradient or vit_training or vit_kernels or vit_ti" 2>&1 | tail -4
python bench.py --train --arch vit_ti --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('vit_ti train', r['value'], r['ms_per_step'], r['step_times'])"
