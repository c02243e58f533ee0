#!/bin/bash
# PROBE (timing only): the ResNet-50 step when conv3 of the first n blocks stores no multiplier and the gradient launch reads none
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['step_times']['sub_batch_stream_steps']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['mfma']['ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'], r['roofline'].get('sclk_mhz_mean'), r['roofline'].get('power_w_mean'))"; }
for i in 1 2; do
  run BCOS_NOOP=1
  run BCOS_PROBE_NO_T3=2
  run BCOS_PROBE_NO_T3=5
  run BCOS_PROBE_NO_T3=12
done
