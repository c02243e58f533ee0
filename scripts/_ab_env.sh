#!/bin/bash
# same-node A/B of an environment setting on the headline step: bash scripts/_ab_env.sh NAME=VALUE [pairs] [bench args]
PAIRS=${2:-3}
run() { env "$1" python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['step_times']['sub_batch_stream_steps']['median'])"; }
EXTRA="${@:3}"
for i in $(seq 1 $PAIRS); do
  run A=default; run "$1"
done
