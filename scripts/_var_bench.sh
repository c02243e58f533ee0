for v in ${VARIANTS:-default}; do
  if [ $v != default ]; then export BCOS_HIP_LIB=$GRAFT_REPO_ROOT/b-cosification_amd/lib/variants/$v.so; else unset BCOS_HIP_LIB; fi
  echo "== $v"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'], r['roofline']['by_bound']['hbm']['achieved_gbps'])"
done
