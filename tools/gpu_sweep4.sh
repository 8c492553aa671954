#!/bin/bash
TAG="${1:-sweep4}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/summary.txt"
run() {
  cfg=$1; shift
  echo "== $cfg $*" | tee -a "$OUT/summary.txt"
  f="$OUT/cb_${cfg}_$(echo "$*" | tr ' =' '__').txt"
  env "$@" timeout 300 python tools/conv_bench.py --config $cfg --batch 32 > "$f" 2>&1
  grep "TOTAL" "$f" | tee -a "$OUT/summary.txt"
}
run taichi MNK_MFMA16=0
run taichi MNK_MFMA16=1
run moving-gif MNK_MFMA16=0
run moving-gif MNK_MFMA16=1
cp "$OUT/cb_taichi_MNK_MFMA16_1.txt" "$OUT/conv_bench_taichi.txt"
