#!/bin/bash
# ViT-Ti batch 512 forward+explanation under a few existing switches (same node)
run() { env "$@" python bench.py --arch vit_ti --batch 512 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$*', r['value'], r['step_times']['all_steps']['median'])"; }
run A=default
run BCOS_SUBBATCH_STREAMS=1
run BCOS_SUBBATCH_STREAMS=3
run BCOS_SUBBATCH_STREAMS=4
run BCOS_OPT_H2_WIDE_COST=8
run BCOS_OPT_H2_TILE=1
run BCOS_OPT_TAIL_SPLIT=0
run BCOS_OPT_EPI_GENERIC=1
run A=default
