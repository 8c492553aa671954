#!/bin/bash
# Build the CPU emulation of libmonkeynet_hip.so (TEST INFRASTRUCTURE ONLY; see include/hip/hip_runtime.h here).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
SRC="$ROOT/monkey-net_amd/csrc"
OUT="$HERE/build"
mkdir -p "$OUT"
OBJS=""
NEWEST_H="$(ls -t "$SRC"/*.h "$ROOT/include/monkeynet_hip.h" "$HERE/include/hip/hip_runtime.h" | head -1)"
for f in "$SRC"/*.hip "$HERE/hipemu.cpp"; do
  o="$OUT/$(basename "$f").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$NEWEST_H" -nt "$o" ]; then
    g++ -O2 -g -std=c++17 -fPIC -x c++ -I"$HERE/include" -I"$ROOT/include" -I"$SRC" -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unused-variable -Wno-psabi -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
g++ -shared -o "$OUT/libmnk_emu.so" $OBJS -ldl
echo "$OUT/libmnk_emu.so"
