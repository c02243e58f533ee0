#!/bin/bash
# per-kernel totals of the headline step in two checkouts on the same node (single stream): bash scripts/_cmp_tree_kernels.sh _oldtree
T=$1; ROOT=$PWD
export TMPDIR=/tmp BCOS_SUBBATCH_STREAMS=1
for d in . $T; do
  name=$(basename $(cd $d && pwd))
  cd /tmp; rm -rf /tmp/cmpk_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cmpk_$name -- python3 $ROOT/$d/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  cd $ROOT
  cp $(find /tmp/cmpk_$name -name "*kernel_stats.csv" | head -1) gpurun_out/cmpk_$name.csv
done
