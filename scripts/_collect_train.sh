#!/bin/bash
# the training-step lines + kernel stats of scripts/collect_profiles.sh alone (step 4b): bash scripts/_collect_train.sh r04
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
for spec in "resnet50" "resnet18" "vit_ti"; do
  python3 bench.py --train --arch $spec --steps 10 --warmup 3 > "$SUM/${TAG}_bench_train_${spec}.json" 2> "$OUT/bench_train_${spec}.err"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_train_${spec}" -- python3 "$ROOT/bench.py" --train --arch $spec --steps 3 --warmup 2 > /dev/null 2> "$OUT/stats_train_${spec}.err"
  cd "$ROOT"
  f=$(find "$OUT/stats_train_${spec}" -name "*kernel_stats.csv" | sort | tail -1)
  [ -n "$f" ] && cp "$f" "$SUM/${TAG}_kernel_stats_train_${spec}.csv"
done
ls -la "$SUM" | grep train
