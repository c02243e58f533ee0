#!/bin/bash
# instruction-cache behaviour of the contraction kernels (round 5): counters of one launch, and of the whole two-stream step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_TC_INST[A-Z_]*" | sort -u | tr '\n' ' '
echo
for spec in "CIN=256 COUT=1024 HH=14" "CIN=64 COUT=256 HH=56"; do
  for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=$R/gpurun_out/ic_$(echo $spec$c | tr ' =' '__' | cut -c1-40)
    env $spec rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/scripts/pmc_fwd.py > /dev/null 2>$d.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    echo "== $spec"
    python3 - "$f" <<'PY'
import csv,sys
agg={}
for r in csv.DictReader(open(sys.argv[1])):
    if 'tapconv' in r['Kernel_Name'] or 'tappatch' in r['Kernel_Name']:
        agg.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, v[-1])
PY
  done
done
# the whole step (two streams): sums over all contraction launches
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  d=$R/gpurun_out/ic_step
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-telemetry > /dev/null 2>$d.err
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== whole step, two streams"
  python3 - "$f" <<'PY'
import csv,sys
agg={}
for r in csv.DictReader(open(sys.argv[1])):
    if 'tapconv' in r['Kernel_Name'] or 'tappatch' in r['Kernel_Name']:
        agg[r['Counter_Name']]=agg.get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in agg.items(): print(k, v)
PY
done
