#!/bin/bash
# same-node A/B: prologue DMA ahead of the operand-maxima loads (product) vs behind them (lib/variants/late.so = -DD_EARLY=0)
run() { BCOS_HIP_LIB=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$2', r['value'], r['step_times']['sub_batch_stream_steps']['median'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['mfma']['ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'], r['roofline']['by_bound']['hbm']['frac'])"; }
for i in 1 2 3; do
  run "" early
  run "$PWD/b-cosification_amd/lib/variants/late.so" late
done
