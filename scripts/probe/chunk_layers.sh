#!/bin/bash
# Per-launch times of one forward+explanation pass at several batch sizes, single stream: does a 16-64-image chunk run the
# stage-1/2 launches faster PER IMAGE (operands found in the 256 MiB Infinity Cache) than the 256-image pass?
mkdir -p gpurun_out/chunk
for B in 256 16 32 64 128 256; do
  BCOS_SUBBATCH_STREAMS=1 B=$B python scripts/layer_report.py > gpurun_out/chunk/layers_B$B.txt 2>&1
  tail -1 gpurun_out/chunk/layers_B$B.txt
done
