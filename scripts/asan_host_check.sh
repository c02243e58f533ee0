#!/bin/bash
# AddressSanitizer check of the C ABI's host side (no GPU needed): builds libbcos_hip with the HOST code sanitised
# (device code unsanitised: -fno-gpu-sanitize; GPU ASan is unavailable on the pool) and runs tests/asan/abi_validation.c
# against it.  Usage: scripts/asan_host_check.sh   (exit code 0 = clean)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/b-cosification_amd/lib/asan"
mkdir -p "$OUT"
SRC="$ROOT/b-cosification_amd/csrc"
LIB="$OUT/libbcos_hip_asan.so"
newest=$(ls -t "$SRC"/*.hip "$SRC"/*.h "$ROOT/include/bcos_hip.h" | head -1)
if [ ! -f "$LIB" ] || [ "$newest" -nt "$LIB" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++20 -fPIC -shared -fsanitize=address -fno-gpu-sanitize \
     -fno-omit-frame-pointer -I"$ROOT/include" -I"$SRC" "$SRC"/bcos_*.hip -o "$LIB"
fi
/opt/rocm/lib/llvm/bin/clang -O1 -g -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" \
   "$ROOT/tests/asan/abi_validation.c" -L"$OUT" -lbcos_hip_asan -Wl,-rpath,"$OUT" -o "$OUT/abi_validation"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 "$OUT/abi_validation"
