#!/bin/bash
mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_full.json 2> gpurun_out/r05/bench_full.err
python -c "
import json; r=json.load(open('gpurun_out/r05/bench_full.json')); print(r['value'], r['ms_per_step']); print({k:v for k,v in r['roofline'].items() if 'clk' in k or 'power' in k or 'clock' in k}); c=r['cpu_baseline']; print({k:c[k] for k in ('value','cores','passes_images_per_s','thread_sweep_images_per_s','note','thread_binding')})"
tail -3 gpurun_out/r05/bench_full.err
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['step_times']['sub_batch_stream_steps']['median'], r['roofline'].get('sclk_mhz_mean'), r['roofline'].get('power_w_mean'))"; }
for i in 1 2; do
  run BCOS_SUBBATCH_STREAMS=2
  run BCOS_SUBBATCH_STREAMS=3
done
runv() { env $1 python bench.py --arch vit_ti --batch 512 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('vit $1', r['value'], r['step_times']['all_steps']['median'], r['roofline'].get('sclk_mhz_mean'), r['roofline'].get('power_w_mean'))"; }
for i in 1 2; do
  runv BCOS_VIT_SUBBATCH_STREAMS=3
  runv BCOS_VIT_SUBBATCH_STREAMS=4
  runv BCOS_VIT_SUBBATCH_STREAMS=6
done
