#!/bin/bash
# A/B of library variants on the per-layer conv benchmark: gpu_ab.sh TAG CFG variant1 variant2 ...  ("" = product lib)
TAG="$1"; CFG="$2"; shift 2
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for v in "$@"; do
  # variant = libtag[:ENV=VAL[,ENV=VAL...]]
  lib="${v%%:*}"; envs=""; [ "$lib" != "$v" ] && envs="$(echo "${v#*:}" | tr ',' ' ')"
  [ "$lib" = "base" ] && L="$PWD/monkey-net_amd/libmonkeynet_hip.so" || L="$PWD/monkey-net_amd/libmonkeynet_hip_$lib.so"
  echo "== $v" | tee -a "$OUT/summary.txt"
  env $envs MNK_LIBRARY="$L" timeout 300 python tools/conv_bench.py --config $CFG --batch 32 > "$OUT/conv_${CFG}_$(echo $v | tr ":=," "___").txt" 2>&1
  grep TOTAL "$OUT/conv_${CFG}_$(echo $v | tr ":=," "___").txt" | tee -a "$OUT/summary.txt"
done
