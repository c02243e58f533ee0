#!/bin/bash
# same-node A/B of the headline step between the built library and other builds of it (scripts/build_d_variants.sh):
#   bash scripts/_ab_lib.sh "<a.so> [b.so ...]" [rounds] [bench args]
# BCOS_HIP_LIB selects the library bcos_hip/lib.py loads; the arms alternate so that node and clock drift hit all of them alike.
# columns: arm, images/s, two-stream median ms, single-stream contraction ms (all | matrix-bound | bandwidth-bound), mean sclk MHz
OTHERS=$1; ROUNDS=${2:-3}; shift; shift
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref "${@:3}" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('$2', r['value'], r['step_times']['sub_batch_stream_steps']['median'], rf['kernel_ms_per_step'], rf['by_bound']['mfma']['ms_per_step'], rf['by_bound']['hbm']['ms_per_step'], rf.get('sclk_mhz_mean'))"; }
for i in $(seq 1 $ROUNDS); do
  run "BCOS_NOOP=1" product "$@"
  for o in $OTHERS; do run "BCOS_HIP_LIB=$o BCOS_ALLOW_DEV_BUILD=1" "$(basename $o .so)" "$@"; done
done
