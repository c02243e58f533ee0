timeout 300 python scripts/layer_report.py > gpurun_out/r03_layers_dma4.txt 2>&1
BCOS_OPT_H2_LOOP=1 timeout 300 python scripts/layer_report.py > gpurun_out/r03_layers_regs4.txt 2>&1
tail -n 1 gpurun_out/r03_layers_dma4.txt gpurun_out/r03_layers_regs4.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r50stats
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r50stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
