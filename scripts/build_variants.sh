#!/bin/bash
# build kernel variants (development aid): scripts/build_variants.sh "name:-DFLAG=1 -DX=0" ...
set -e
cd "$(dirname "$0")/../b-cosification_amd"
mkdir -p lib/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -I../include -Icsrc $flags \
     csrc/bcos_tapconv.hip csrc/bcos_skinny.hip csrc/bcos_elementwise.hip csrc/bcos_vit.hip csrc/bcos_render.hip csrc/bcos_train.hip csrc/bcos_abi.hip -o lib/variants/$name.so &
done
wait
ls -la lib/variants
