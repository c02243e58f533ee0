#!/bin/bash
# AddressSanitizer check of the C ABI's host side (no GPU needed): builds libbcos_hip with the HOST code sanitised
# (device code unsanitised: -fno-gpu-sanitize; GPU ASan is unavailable on the pool) and runs tests/asan/abi_validation.c
# against it.  Usage: scripts/asan_host_check.sh   (exit code 0 = clean)
# The objects are compiled in parallel, bcos_tapconv.hip in the same -DBCOS_TAPCONV_PART slices as the product build.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/b-cosification_amd/lib/asan"
mkdir -p "$OUT/obj"
SRC="$ROOT/b-cosification_amd/csrc"
LIB="$OUT/libbcos_hip_asan.so"
PARTS=$(python3 -c "import re;print(re.search(r'TAPCONV_PARTS = (\d+)', open('$ROOT/b-cosification_amd/bcos_hip/lib.py').read()).group(1))")
newest=$(ls -t "$SRC"/*.hip "$SRC"/*.h "$ROOT/include/bcos_hip.h" | head -1)
if [ ! -f "$LIB" ] || [ "$newest" -nt "$LIB" ]; then
  FLAGS="--offload-arch=gfx950 -O1 -g -std=c++20 -fPIC -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -I$ROOT/include -I$SRC"
  jobs=()
  for f in "$SRC"/bcos_*.hip; do
    b=$(basename "$f" .hip)
    if [ "$b" = bcos_tapconv ]; then
      for k in $(seq 0 $((PARTS - 1))); do jobs+=("/opt/rocm/bin/hipcc $FLAGS -DBCOS_TAPCONV_PART=$k -c $f -o $OUT/obj/${b}_p$k.o"); done
    else
      jobs+=("/opt/rocm/bin/hipcc $FLAGS -c $f -o $OUT/obj/$b.o")
    fi
  done
  rm -f "$OUT"/obj/*.o
  printf '%s\n' "${jobs[@]}" | xargs -P 8 -I{} bash -c "{}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address "$OUT"/obj/*.o -o "$LIB"
fi
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" \
   "$ROOT/tests/asan/abi_validation.c" -L"$OUT" -lbcos_hip_asan -Wl,-rpath,"$OUT" -o "$OUT/abi_validation"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 "$OUT/abi_validation"
