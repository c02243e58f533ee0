#!/bin/bash
# same-node A/B of an environment setting on the ViT-Ti bench: bash scripts/_ab_env_vit.sh NAME=VALUE [pairs]
PAIRS=${2:-3}
run() { env "$1" python bench.py --arch vit_ti --batch 512 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['step_times']['all_steps']['median'])"; }
for i in $(seq 1 $PAIRS); do
  run A=default; run "$1"
done
