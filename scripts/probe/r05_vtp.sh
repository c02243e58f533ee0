#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "training or train or layernorm_gradient" 2>&1 | tail -3
for rep in 1 2; do
for a in resnet50 resnet18 clip_rn50 vit_ti; do
python bench.py --train --arch $a --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$a', r['value'], r['ms_per_step'])"
done
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python3 $GRAFT_REPO_ROOT/bench.py --train --arch resnet50 --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find /tmp/tp -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step', tot/7e6, 'launches/step', sum(int(r['Calls']) for r in rows)/7)
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls'])/7:7.1f} {float(r['TotalDurationNs'])/7e6:7.3f} {float(r['AverageNs'])/1e3:8.1f}")
PY
