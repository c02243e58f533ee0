run() { echo "== $*"; env "$@" timeout 300 python bench.py --arch vit_ti --batch 512 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms_per_step'])"; }
run A=1
run BCOS_VIT_F16X2=0
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "vit" 2>&1 | tail -3
