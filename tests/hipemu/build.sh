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
# relink only when an object is newer than the library (and atomically: a process may have the old file mapped)
NEED=0
[ -f "$OUT/libmnk_emu.so" ] || NEED=1
for o in $OBJS; do [ "$o" -nt "$OUT/libmnk_emu.so" ] && NEED=1; done
if [ "$NEED" = 1 ]; then
  g++ -shared -o "$OUT/libmnk_emu.so.tmp.$$" $OBJS -ldl -lrt && mv -f "$OUT/libmnk_emu.so.tmp.$$" "$OUT/libmnk_emu.so"
fi
echo "$OUT/libmnk_emu.so"
