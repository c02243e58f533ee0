#!/bin/bash
# 8 concurrent copies of scripts/probe/var_triage.py on the one device (the contention the divergence was seen under)
mkdir -p gpurun_out
for i in 0 1 2 3 4 5 6 7; do ITERS=${ITERS:-400} timeout ${TMO:-1500} python scripts/probe/var_triage.py $i 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_var_triage_$i.log & done
wait
cat gpurun_out/r4_var_triage_*.log | grep -c "mode" > gpurun_out/r4_var_triage_summary.txt
grep -h "done in" gpurun_out/r4_var_triage_*.log >> gpurun_out/r4_var_triage_summary.txt
