run() { echo "== $*"; timeout 300 python bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"; }
run --arch resnet50
run --arch resnet50 --forward-only
run --arch resnet18
run --arch clip_rn50
run --arch clip_rn50 --forward-only
run --arch vit_ti --batch 512
run --arch vit_ti --batch 512 --forward-only
