#!/bin/bash
# variants of the whole tapconv file (development aid): scripts/build_d_variants.sh "name:-DFLAG=1" ...  (all slices rebuilt, linked with the other objects)
# Knock-outs and the other fenced switches (csrc/bcos_internal.h: BCOS_DEV_SWITCH) need -DBCOS_DEV_BUILD among the flags; bcos_abi.hip is
# rebuilt with the variant's flags too, so that the variant's bcos_version() carries BCOS_VERSION_DEV_FLAG (load it with BCOS_ALLOW_DEV_BUILD=1).
set -e
cd "$(dirname "$0")/../b-cosification_amd"
mkdir -p lib/variants /tmp/dvar
PARTS=$(python3 -c "import re;print(re.search(r'TAPCONV_PARTS = (\d+)', open('bcos_hip/lib.py').read()).group(1))")
OTHERS=$(ls lib/obj/*.o | grep -v "bcos_tapconv_p\|bcos_abi")
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  for k in $(seq 0 $((PARTS - 1))); do
    echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I../include -Icsrc $flags -DBCOS_TAPCONV_PART=$k -c csrc/bcos_tapconv.hip -o /tmp/dvar/${name}_p$k.o"
  done
  echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I../include -Icsrc $flags -c csrc/bcos_abi.hip -o /tmp/dvar/${name}_abi.o"
done | xargs -P 8 -I{} bash -c "{}"
for spec in "$@"; do
  name="${spec%%:*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/dvar/${name}_abi.o /tmp/dvar/${name}_p*.o -o lib/variants/$name.so
done
ls -la lib/variants
