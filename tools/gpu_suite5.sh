#!/bin/bash
# The whole GPU suite N times back to back in ONE box visit (verdict r4 item 1d: a suite that is green once may be a coin flip).
# Usage: gpu_suite5.sh [TAG] [N]  -> gpurun_out/TAG/run_k.txt (the tail of every run) + summary.txt
TAG="${1:-r05suite}"; N="${2:-5}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
S="$OUT/summary.txt"; echo "commit $(cat .gpurun_commit 2>/dev/null || echo unknown); pytest tests -m gpu -x -q, $N runs in one visit" > "$S"
for k in $(seq 1 "$N"); do
  timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > "$OUT/run_$k.log" 2>&1; rc=$?
  tail -3 "$OUT/run_$k.log" | cut -c1-200 > "$OUT/run_$k.txt"
  echo "run $k: rc=$rc  $(tail -1 "$OUT/run_$k.log" | cut -c1-160)" | tee -a "$S"
done
