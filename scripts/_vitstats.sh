cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/vitstats
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vitstats -- python3 $GRAFT_REPO_ROOT/bench.py --arch vit_ti --batch 512 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/vitstats -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-150 | awk -F'"' '{print $2 "|" $3}' | cut -c1-170
