#!/bin/bash
for rep in 1 2; do
for pa in 0 1; do
BCOS_PUBLISH_ALWAYS=$pa python bench.py --train --arch vit_ti --steps 10 --warmup 3 --no-train-plan 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('vit per-layer publish_always=$pa', r['value'], r['ms_per_step'])"
done
done
