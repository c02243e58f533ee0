#!/bin/bash
# variants of the patch kernels only (development aid): scripts/build_patch_variants.sh "name:-DP_KO=1" ...
# compiles slices 9 and 10 of bcos_tapconv.hip with the flags and links them with the product build's other objects
set -e
cd "$(dirname "$0")/../b-cosification_amd"
mkdir -p lib/variants /tmp/pvar
OTHERS=$(ls lib/obj/*.o | grep -v "_p9.o\|_p10.o")
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  for k in 9 10; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I../include -Icsrc $flags -DBCOS_TAPCONV_PART=$k -c csrc/bcos_tapconv.hip -o /tmp/pvar/${name}_p$k.o &
  done
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/pvar/${name}_p9.o /tmp/pvar/${name}_p10.o -o lib/variants/$name.so
done
ls -la lib/variants
