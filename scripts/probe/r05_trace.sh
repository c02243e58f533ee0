#!/bin/bash
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05/trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05/trace -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events --no-telemetry > gpurun_out/r05/trace_bench.json 2> gpurun_out/r05/trace.err
ls -la gpurun_out/r05/trace/*/ | head
python3 scripts/probe/trace_timeline.py gpurun_out/r05/trace | tee gpurun_out/r05/trace_timeline.txt
