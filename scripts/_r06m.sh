run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$1', r['value'], r['step_times']['sub_batch_stream_steps']['median'])"; }
for i in 1 2; do
  for a in BCOS_SUBBATCH_SKEW=0 BCOS_SUBBATCH_SKEW=2 BCOS_SUBBATCH_SKEW=4 BCOS_SUBBATCH_SKEW=6 BCOS_SUBBATCH_SKEW=9 BCOS_SUBBATCH_SKEW=12 "BCOS_SUBBATCH_STREAMS=3 BCOS_SUBBATCH_SKEW=4" "BCOS_SUBBATCH_STREAMS=4 BCOS_SUBBATCH_SKEW=3"; do
    run "$a"
  done
done
