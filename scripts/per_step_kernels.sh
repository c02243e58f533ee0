#!/bin/bash
# Which kernels does ONE timed step launch, and for how long?  Two rocprofv3 kernel traces of bench.py that differ only in the number of
# timed steps (1 and 4, single stream so that a launch is a launch of the whole batch); the difference of the per-kernel totals / 3 is
# one step, free of model build, calibration, warm-up and the vendor / copy references.
#   bash scripts/per_step_kernels.sh <tag> [bench args]  ->  gpurun_out/<tag>_per_step_kernels.txt
TAG=${1:-r06}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/psk_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp BCOS_SUBBATCH_STREAMS=${BCOS_SUBBATCH_STREAMS:-1}
cd /tmp
for S in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s$S" -- python3 "$ROOT/bench.py" --steps $S --warmup 1 --no-cpu-baseline --no-vendor-ref --no-kernel-events --no-telemetry "$@" > /dev/null 2> "$OUT/s$S.err"
done
cd "$ROOT"
python3 - "$OUT" > "gpurun_out/${TAG}_per_step_kernels.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def load(d):
    f = glob.glob(f"{out}/{d}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("s1"), load("s4")
rows = []
for name, (c4, t4) in b.items():
    c1, t1 = a.get(name, (0, 0.0))
    dc, dt = (c4 - c1) / 3.0, (t4 - t1) / 3.0 / 1e3
    if dc > 0.01:
        rows.append((dt, dc, name))
rows.sort(reverse=True)
contr = sum(dt for dt, dc, n in rows if "tapconv" in n or "tappatch" in n or "skinny" in n)
other = sum(dt for dt, dc, n in rows if not ("tapconv" in n or "tappatch" in n or "skinny" in n))
print(f"# per timed step (difference of a 4-step and a 1-step trace / 3): contraction kernels {contr/1e3:.3f} ms, every other kernel {other/1e3:.3f} ms")
print(f"# {'us/step':>10s} {'calls/step':>10s} {'us/call':>9s}  kernel")
for dt, dc, n in rows:
    print(f"{dt:12.1f} {dc:10.1f} {dt/dc:9.1f}  {n[:150]}")
PY
