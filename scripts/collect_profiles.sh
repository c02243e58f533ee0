#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2700 -- 'bash scripts/collect_profiles.sh r06'
# Writes gpurun_out/prof_<tag>/{stats,fetch,write,...}/ and gpurun_out/profiles_<tag>/ (the summaries to copy into profiles/).
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-vendor-ref"
# 1. the headline command as the driver runs it (CPU baseline included)
python3 bench.py --steps 20 --warmup 5 > "$SUM/${TAG}_bench.json" 2> "$OUT/bench.err"
# 2. kernel stats + HBM counters of the same workload (counters in their own passes, MI355X_MICROARCH.md).  The profiled runs keep
#    every launch on ONE stream (BCOS_SUBBATCH_STREAMS=1: 116 contraction launches per step, each over the whole batch) -- the form
#    bench.py's own event-carrying steps use -- so that a launch in the trace is the launch bench.py's HIP events bracket; with the
#    default two sub-batch streams a step is 234 half-batch launches that overlap in time.
export BCOS_SUBBATCH_STREAMS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$SUM/${TAG}_bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/write.err"
cd "$ROOT"
unset BCOS_SUBBATCH_STREAMS
python3 scripts/summarise_profiles.py "$OUT" "$SUM" "$TAG"
# 3. per-launch table of the same step: TFLOP/s and GB/s per layer geometry, with the bound each launch sits on (single stream, as above:
#    an event pair around a launch must not time a neighbour on the other sub-batch stream)
BCOS_SUBBATCH_STREAMS=1 LAYERS_CSV="$SUM/${TAG}_layers.csv" python3 scripts/layer_report.py > "$SUM/${TAG}_layers.txt" 2> "$OUT/layers.err"
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
# 4b. the training-step diagnostic (bench.py --train: training plan for the ResNets, nn.Module path for the ViT; reference batch 64 per GPU) and its kernel stats
for spec in "resnet50" "resnet18" "vit_ti" "clip_rn50"; do
  python3 bench.py --train --arch $spec --steps 10 --warmup 3 > "$SUM/${TAG}_bench_train_${spec}.json" 2> "$OUT/bench_train_${spec}.err"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_train_${spec}" -- python3 "$ROOT/bench.py" --train --arch $spec --steps 3 --warmup 2 > /dev/null 2> "$OUT/stats_train_${spec}.err"
  cd "$ROOT"
  f=$(find "$OUT/stats_train_${spec}" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$SUM/${TAG}_kernel_stats_train_${spec}.csv"
done
# 5. SQ counters of four representative contraction launches (one counter group per pass): A = GEMM-shaped 3x3-class launch
#    M = 50176, K = 2304, N = 256; B = forward 64 -> 256 @56^2 (residual + ReLU + stored multiplier); C = forward 256 -> 1024 @14^2
{
  echo "SQ counters of four representative contraction launches (scripts/_pmc.sh: rocprofv3 --kernel-trace --pmc ..., one counter group per pass;"
  echo "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE and the LDS-array"
  echo "counters SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count cycles -- 3.9 per ds_read_b128).  derived: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /"
  echo "(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); LDS array busy = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)"
  echo "== A 3x3-class K=2304 (M=50176, N=256, 128x256 tiles, split-f16 loop with LDS-DMA staging)"
  MKN=50176,2304,256 bash scripts/_pmc.sh ${TAG}pmcA 2>/dev/null
  echo "== B fwd 64->256 @56^2 (residual + ReLU + stored multiplier)"
  PMC_SCRIPT=pmc_fwd.py CIN=64 COUT=256 HH=56 bash scripts/_pmc.sh ${TAG}pmcB 2>/dev/null
  echo "== C fwd 256->1024 @14^2"
  PMC_SCRIPT=pmc_fwd.py CIN=256 COUT=1024 HH=14 bash scripts/_pmc.sh ${TAG}pmcC 2>/dev/null
  echo "== D fwd 3x3 256->256 @14^2 (M=50176, K=2304, N=256: the input-patch loop, tile_body_p)"
  PMC_SCRIPT=pmc_conv.py bash scripts/_pmc.sh ${TAG}pmcD 2>/dev/null
} > "$SUM/${TAG}_pmc_raw.txt"
cd "$ROOT"
python3 - "$SUM/${TAG}_pmc_raw.txt" > "$SUM/${TAG}_pmc_summary.txt" <<'PY'
import sys
txt = open(sys.argv[1]).read()
print(txt.rstrip())
blocks = txt.split("== ")[1:]
print()
for b in blocks:
    lines = b.strip().splitlines()
    vals = {}
    for ln in lines[1:]:
        p = ln.split()
        if len(p) == 2:
            try: vals[p[0]] = float(p[1])
            except ValueError: pass
    durs = [float(ln.split()[1]) for ln in lines if ln.startswith("dur_us")]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * vals["GRBM_GUI_ACTIVE"] / 8)
        wc = vals.get("SQ_WAVE_CYCLES", 0) or 1
        print(f"derived [{lines[0][:40]}]: MFMA busy {100 * busy:.1f} % of SIMD cycles; duration under the profiler {min(durs) if durs else float('nan'):.1f} us; "
              f"wave cycles: issuing {100 * vals.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} % / issue-stalled {100 * vals.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} % / "
              f"waiting on a counter or barrier {100 * vals.get('SQ_WAIT_ANY', 0) / wc:.1f} %; VALU {vals.get('SQ_INSTS_VALU', 0):.0f} SALU {vals.get('SQ_INSTS_SALU', 0):.0f} "
              f"MFMA {vals.get('SQ_INSTS_MFMA', 0):.0f} LDS {vals.get('SQ_INSTS_LDS', 0):.0f} VMEM {vals.get('SQ_INSTS_VMEM', 0):.0f} instructions")
PY
# 6. which kernels ONE timed step launches, and for how long (difference of a 4-step and a 1-step trace): headline, CLIP forward, ViT-Ti, ResNet-50 training
cd "$ROOT"
bash scripts/per_step_kernels.sh ${TAG}_headline > /dev/null 2>&1; cp gpurun_out/${TAG}_headline_per_step_kernels.txt "$SUM/${TAG}_per_step_kernels.txt"
bash scripts/per_step_kernels.sh ${TAG}_clipfwd --arch clip_rn50 --forward-only > /dev/null 2>&1; cp gpurun_out/${TAG}_clipfwd_per_step_kernels.txt "$SUM/${TAG}_per_step_kernels_clip_rn50_forwardonly.txt"
BCOS_SUBBATCH_STREAMS=3 bash scripts/per_step_kernels.sh ${TAG}_vit --arch vit_ti --batch 512 > /dev/null 2>&1; cp gpurun_out/${TAG}_vit_per_step_kernels.txt "$SUM/${TAG}_per_step_kernels_vit_ti_batch_512.txt"
bash scripts/per_step_kernels.sh ${TAG}_train --train --arch resnet50 > /dev/null 2>&1; cp gpurun_out/${TAG}_train_per_step_kernels.txt "$SUM/${TAG}_per_step_kernels_train_resnet50.txt"
# 7. the weight-gradient launch on the ResNet-50 / ViT-Ti training shapes: ordered (round 6) against atomics (round 5)
{ for f in 1 0; do echo "== BCOS_WGRAD_ORDERED=$f (ResNet-50 shapes, batch 64)"; BCOS_WGRAD_ORDERED=$f python3 scripts/probe/wgrad_probe.py 2>/dev/null
  echo "== BCOS_WGRAD_ORDERED=$f (ViT-Ti token linears, 12 608 rows)"; BCOS_WGRAD_ORDERED=$f WGRAD_VIT=1 python3 scripts/probe/wgrad_probe.py 2>/dev/null; done; } > "$SUM/${TAG}_wgrad_probe.txt"
ls -la "$SUM"
