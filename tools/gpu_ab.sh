#!/bin/bash
# A/B of library variants on the per-layer conv benchmark: gpu_ab.sh TAG CFG variant1 variant2 ...  ("" = product lib)
TAG="$1"; CFG="$2"; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for v in "$@"; do
  [ "$v" = "base" ] && L="$PWD/monkey-net_amd/libmonkeynet_hip.so" || L="$PWD/monkey-net_amd/libmonkeynet_hip_$v.so"
  echo "== $v" | tee -a "$OUT/summary.txt"
  MNK_LIBRARY="$L" timeout 300 python tools/conv_bench.py --config $CFG --batch 32 > "$OUT/conv_${CFG}_$v.txt" 2>&1
  grep TOTAL "$OUT/conv_${CFG}_$v.txt" | tee -a "$OUT/summary.txt"
done
