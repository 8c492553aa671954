#!/bin/bash
# A/B of arbitrary environment settings on the whole training iteration inside ONE box visit:
#   gpu_env_ab.sh TAG "" "VAR=1" "VAR=1 OTHER=2" ...      ("" = defaults); REPS interleaved runs each
TAG="$1"; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for rep in $(seq 1 ${REPS:-2}); do
  k=0
  for v in "$@"; do
    k=$((k+1)); f="$OUT/bench_${k}_$rep.json"
    env $v timeout 200 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-profile --dropin 0 ${BENCH_ARGS:-} > "$f" 2> "$f.err"
    echo "rep=$rep [${v:-default}]: $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"capture_failed": [a-z]*' "$f")" | tee -a "$OUT/summary.txt"
  done
done
