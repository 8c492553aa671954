#!/bin/bash
# A/B of MNK_TUNING values on the whole training iteration inside ONE box visit (values contain commas, so one argument each):
#   gpu_tuning_ab.sh TAG "" "wgroup_chunk=512" "wgroup_chunk=512,wgroup_long=2" ...      ("" = defaults); REPS interleaved runs each
TAG="$1"; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    f="$OUT/bench_$(echo "${v:-default}" | tr '=,' '__')_$rep.json"
    MNK_TUNING="$v" timeout 200 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-profile --dropin 0 ${BENCH_ARGS:-} > "$f" 2> "$f.err"
    echo "rep=$rep ${v:-default}: $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"capture_failed": [a-z]*' "$f")" | tee -a "$OUT/summary.txt"
  done
done
