#!/bin/bash
# same-node A/B of two builds of the library on the headline step: bash scripts/probe/r05_ab_lib.sh <other.so> [pairs]
OTHER=$1; PAIRS=${2:-3}
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$2', r['value'], r['step_times']['sub_batch_stream_steps']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['mfma']['ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'], r['roofline'].get('sclk_mhz_mean'))"; }
for i in $(seq 1 $PAIRS); do
  run "BCOS_HIP_LIB=$OTHER" old
  run "BCOS_NOOP=1" new
done
