timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vit or attention" > gpurun_out/r4_attn_tests.txt 2>&1; tail -3 gpurun_out/r4_attn_tests.txt
run() { BCOS_HIP_LIB=$1 python bench.py --arch vit_ti --batch 512 --steps 10 --warmup 3 --no-cpu-baseline $3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$2', r['value'], r['step_times']['all_steps']['median'])"; }
for i in 1 2 3; do
  run "" new; run $PWD/b-cosification_amd/lib/variants/prev.so prev
done
run "" "new fwd" --forward-only; run $PWD/b-cosification_amd/lib/variants/prev.so "prev fwd" --forward-only
