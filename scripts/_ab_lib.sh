#!/bin/bash
# same-node A/B of the headline step between the built library and another build of it: bash scripts/_ab_lib.sh <other.so> [pairs] [bench args]
# (BCOS_HIP_LIB selects the library bcos_hip/lib.py loads; the runs alternate so that node and clock drift hit both arms alike)
OTHER=$1; PAIRS=${2:-3}; shift; shift
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref "${@:3}" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('$2', r['value'], r['step_times']['sub_batch_stream_steps']['median'], rf['kernel_ms_per_step'], rf['by_bound']['mfma']['ms_per_step'], rf['by_bound']['hbm']['ms_per_step'], rf.get('sclk_mhz_mean'))"; }
for i in $(seq 1 $PAIRS); do
  run "BCOS_NOOP=1" product "$@"
  run "BCOS_HIP_LIB=$OTHER" other "$@"
done
