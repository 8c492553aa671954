#!/bin/bash
# Build libmonkeynet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
# MNK_BUILD_TAG=_x MNK_EXTRA_FLAGS=-D... builds an experiment variant next to the product library
TAG="${MNK_BUILD_TAG:-}"
OUT="$HERE/build$TAG"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE -Wall -Wno-unused-function -Wno-unused-variable ${MNK_EXTRA_FLAGS}"
OBJS=""
pids=""
NEWEST_H="$(ls -t "$HERE"/*.h "$ROOT/include/monkeynet_hip.h" | head -1)"
for f in "$HERE"/*.hip; do
  o="$OUT/$(basename "$f").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$NEWEST_H" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids="$pids $!"
  fi
  OBJS="$OBJS $o"
done
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$HERE/../libmonkeynet_hip$TAG.so" $OBJS -ldl
echo "$HERE/../libmonkeynet_hip$TAG.so"
# the CPython binding of the C-ABI, generated from the header (mnk/_lib.py binds it to whichever library it loads; without it
# every call goes through ctypes)
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
PYEXT="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
FAST="$HERE/../_mnkfast$PYEXT"
if [ ! -f "$FAST" ] || [ "$ROOT/include/monkeynet_hip.h" -nt "$FAST" ] || [ "$HERE/gen_fastcall.py" -nt "$FAST" ]; then
  # (not fatal: without the module every call goes through ctypes -- slower on the host, same results)
  ( python3 "$HERE/gen_fastcall.py" "$ROOT/include/monkeynet_hip.h" "$OUT/_mnkfast.c" > /dev/null &&
    gcc -O2 -shared -fPIC -Wall -Wno-unused-function -I"$PYINC" -o "$FAST.tmp.$$" "$OUT/_mnkfast.c" && mv -f "$FAST.tmp.$$" "$FAST" ) ||
    { rm -f "$FAST.tmp.$$"; echo "warning: the _mnkfast binding was not built (python3 headers / gcc?): calls will go through ctypes" >&2; }
fi
[ -f "$FAST" ] && echo "$FAST"
exit 0
