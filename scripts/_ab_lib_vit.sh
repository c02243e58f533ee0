#!/bin/bash
# same-node A/B of a variant library on the ViT-Ti bench: bash scripts/_ab_lib_vit.sh b-cosification_amd/lib/variants/<name>.so [pairs]
LIBV=$PWD/$1; PAIRS=${2:-3}
run() { BCOS_HIP_LIB=$1 python bench.py --arch vit_ti --batch 512 --steps 10 --warmup 3 --no-cpu-baseline $3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$2', r['value'], r['step_times']['all_steps']['median'])"; }
for i in $(seq 1 $PAIRS); do
  run "" product; run "$LIBV" variant
done
run "" "product fwd" --forward-only; run "$LIBV" "variant fwd" --forward-only
