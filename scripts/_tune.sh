run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms_per_step'], r['roofline']['by_bound']['mfma']['ms_per_step'], r['roofline']['by_bound']['hbm']['ms_per_step'])"; }
run A=1
run A=1
run BCOS_OPT_H2_TILE=1
run BCOS_OPT_H2_TILE=2
run BCOS_OPT_H2_TALL=0
run BCOS_F16X2_MIN_K=128
run A=2
