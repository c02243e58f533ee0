# patch loop against the per-tap loop on the multi-tap ResNet-50 shapes (development aid)
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "patch_loop" 2>&1 | tail -5
PSHAPES=3 AB_OPT=patch=0 timeout 300 python scripts/d_bench.py 2>&1 | grep fwd
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms_per_step'])"; }
run A=1
run BCOS_OPT_PATCH=0
run A=1
run BCOS_OPT_PATCH=0
