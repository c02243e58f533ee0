# kernel-level rocprofv3 stats of one bench invocation (development aid): bash scripts/prof_stats.sh <tag> [bench args]
TAG=${1:-stats}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:25]:
    print(f"{float(r['TotalDurationNs'])/1e6/5:9.3f} ms/step {int(r['Calls'])/5:6.1f} calls/step {float(r['AverageNs'])/1e3:9.1f} us avg  {r['Name'][:110]}")
print("total ms/step", tot/1e6/5)
PY
