timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "patch_loop" 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout 300 python bench.py --arch clip_rn50 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"; }
run A=1
run BCOS_OPT_PATCH=0
run A=1
run BCOS_OPT_PATCH=0
