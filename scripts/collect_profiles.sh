#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash scripts/collect_profiles.sh r02'
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,...}/ and gpurun_out/profiles_<tag>/ (the summaries to copy into profiles/).
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
# 1. the headline command as the driver runs it (CPU baseline included)
python3 bench.py --steps 20 --warmup 5 > "$SUM/${TAG}_bench.json" 2> "$OUT/bench.err"
# 2. kernel stats + HBM counters of the same workload (counters in their own passes, MI355X_MICROARCH.md)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$SUM/${TAG}_bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/write.err"
cd "$ROOT"
python3 scripts/summarise_profiles.py "$OUT" "$SUM" "$TAG"
# 3. per-launch table of the same step: TFLOP/s and GB/s per layer geometry, with the bound each launch sits on
LAYERS_CSV="$SUM/${TAG}_layers.csv" python3 scripts/layer_report.py > "$SUM/${TAG}_layers.txt" 2> "$OUT/layers.err"
# 4. the other BASELINE.json configurations: bench line + kernel stats each
for spec in "resnet50 --forward-only" "resnet18" "vit_ti --batch 512" "vit_ti --batch 512 --forward-only" "clip_rn50" "clip_rn50 --forward-only"; do
  name=$(echo $spec | tr -d '-' | tr ' ' '_')
  python3 bench.py --arch $spec --steps 10 --warmup 3 --no-cpu-baseline > "$SUM/${TAG}_bench_${name}.json" 2> "$OUT/bench_${name}.err"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_${name}" -- python3 "$ROOT/bench.py" --arch $spec $ARGS > /dev/null 2> "$OUT/stats_${name}.err"
  cd "$ROOT"
  f=$(find "$OUT/stats_${name}" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$SUM/${TAG}_kernel_stats_${name}.csv"
done
ls -la "$SUM"
