#!/bin/bash
# same-node A/B of the headline step against another checkout of the repo (built in place): bash scripts/_ab_tree.sh _oldtree [pairs]
T=$1; PAIRS=${2:-3}
run() { (cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null) | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$2', r['value'], r['step_times']['sub_batch_stream_steps']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['mfma']['ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'])"; }
for i in $(seq 1 $PAIRS); do
  run . product; run $T other
done
