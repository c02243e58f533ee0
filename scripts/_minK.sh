for k in 256 128 64; do
  echo "== BCOS_F16X2_MIN_K=$k"
  BCOS_F16X2_MIN_K=$k timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['mfma']['ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'], r['roofline']['by_bound']['mfma']['frac_of_executing_pipe'], r['roofline']['by_bound']['hbm']['frac'])"
done
