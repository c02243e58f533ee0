#!/bin/bash
# same-node A/B of the ViT plan with the LayerNorms folded into the contractions (default) against the round-3 plan (BCOS_VIT_LN_FUSED=0)
run() { BCOS_VIT_LN_FUSED=$1 python bench.py --arch vit_ti --batch 512 --steps 10 --warmup 3 --no-cpu-baseline $3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$2', r['value'], r['ms_per_step'], r['step_times']['sub_batch_stream_steps']['median'])"; }
for i in 1 2 3; do
  run 1 "fused   fwd+expl"; run 0 "unfused fwd+expl"
done
run 1 "fused   fwd" --forward-only; run 0 "unfused fwd" --forward-only
