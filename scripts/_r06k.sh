run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('$1', r['value'], r['step_times']['sub_batch_stream_steps']['median'], rf['kernel_ms_per_step'], rf['by_bound']['mfma']['ms_per_step'], rf['by_bound']['hbm']['ms_per_step'])"; }
for i in 1 2 3 4; do
  run BCOS_NOOP=1
  run BCOS_OPT_H2_WIDE_COST=5
  run BCOS_OPT_H2_WIDE_COST=6
done
