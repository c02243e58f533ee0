#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash scripts/collect_profiles.sh r01'
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}/... and gpurun_out/profiles_<tag>/ (the summaries to copy into profiles/).
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
python3 bench.py > "$SUM/${TAG}_bench.json" 2> "$OUT/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$SUM/${TAG}_bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/write.err"
cd "$ROOT"
python3 scripts/summarise_profiles.py "$OUT" "$SUM" "$TAG"
ls -la "$SUM"
