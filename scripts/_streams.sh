# same-node A/B of the number of sub-batch streams (development aid)
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"; }
for rep in 1 2; do
for s in 1 2 3 4; do
run BCOS_SUBBATCH_STREAMS=$s
done
done
EXTRA="--arch vit_ti --batch 512"
for s in 2 3 4; do run BCOS_SUBBATCH_STREAMS=$s; done
