#!/bin/bash
TAG="${1:-sweep3}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/summary.txt"
run() {
  cfg=$1; shift
  echo "== $cfg $*" | tee -a "$OUT/summary.txt"
  f="$OUT/cb_${cfg}_$(echo "$*" | tr ' =' '__').txt"
  env "$@" timeout 300 python tools/conv_bench.py --config $cfg --batch 32 > "$f" 2>&1
  grep "TOTAL wgrad" "$f" | tee -a "$OUT/summary.txt"
}
run taichi MNK_WGRAD_HALO=0
run taichi MNK_WHALO_TARGET=768 MNK_WHALO_MINTILES=8
run taichi MNK_WHALO_TARGET=512 MNK_WHALO_MINTILES=8
run taichi MNK_WHALO_TARGET=1536 MNK_WHALO_MINTILES=8
run taichi MNK_WHALO_TARGET=768 MNK_WHALO_MINTILES=4
run taichi MNK_WHALO_TARGET=768 MNK_WHALO_MINTILES=16
run taichi MNK_WHALO_TARGET=1536 MNK_WHALO_MINTILES=4
run moving-gif MNK_WGRAD_HALO=0
run moving-gif MNK_WHALO_TARGET=768 MNK_WHALO_MINTILES=8
cp "$OUT/cb_taichi_MNK_WHALO_TARGET_768_MNK_WHALO_MINTILES_8.txt" "$OUT/conv_bench_taichi.txt"
