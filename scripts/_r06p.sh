python -m pytest tests/test_gpu_parity.py -x -q -k "vit" 2>&1 | tail -3
for i in 1 2 3; do
python bench.py --arch vit_ti --batch 512 --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('vit', r['value'], r['step_times']['sub_batch_stream_steps']['median'])"
done
