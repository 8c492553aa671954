#!/bin/bash
# A/B of environment knobs on the whole training iteration inside ONE box visit (boxes differ by up to 10 %):
#   gpu_knob_ab.sh TAG "" "MNK_SPLIT_TILES=384" "MNK_SPLIT_TILES=384,MNK_SPLIT_TARGET=768" ...   ("" = defaults)
# every variant runs REPS times (default 2), interleaved; prints ms/step per run.
TAG="$1"; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/summary.txt"
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    envs="$(echo "$v" | tr ',' ' ')"
    f="$OUT/bench_$(echo "${v:-default}" | tr '=,' '__')_$rep.json"
    env $envs timeout 200 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-profile --dropin 0 ${BENCH_ARGS:-} > "$f" 2> "$f.err"
    echo "rep=$rep ${v:-default}: $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1)" | tee -a "$OUT/summary.txt"
  done
done
