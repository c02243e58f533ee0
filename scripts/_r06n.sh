run() { env $1 python bench.py --train --arch $2 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-ref 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', '$2', r['value'], r['ms_per_step'])"; }
for i in 1 2 3; do
  run BCOS_TRAIN_PREP_AHEAD=2 resnet50
  run BCOS_TRAIN_PREP_AHEAD=0 resnet50
done
run BCOS_TRAIN_PREP_AHEAD=2 resnet18
run BCOS_TRAIN_PREP_AHEAD=0 resnet18
run BCOS_TRAIN_PREP_AHEAD=2 clip_rn50
run BCOS_TRAIN_PREP_AHEAD=0 clip_rn50
