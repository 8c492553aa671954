#!/bin/bash
# visit 15: split-depth knobs of the 64x64 plan, tap-major weight-gradient tile height, chunk
OUT=gpurun_out/r02v15; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v15/ab "" "MNK_SPLIT64_DEEP=16" "MNK_SPLIT64_DEEP=0" "MNK_SPLIT64_DEEP=16,MNK_SPLIT64_MINSTEPS=9" "MNK_SPLIT64_TILES=512" "MNK_WTAP_BM_MAX=64" "MNK_WTAP_BM_MAX=64,MNK_WGROUP_CHUNK=512" "MNK_PLAN_TABLE=0" 2>&1 | tee "$OUT/summary.txt"
BENCH_ARGS="--config taichi" REPS=1 STEPS=30 bash tools/gpu_knob_ab.sh r02v15/ab_taichi "" "MNK_SPLIT64_DEEP=16" "MNK_WTAP_BM_MAX=64" 2>&1 | tee -a "$OUT/summary.txt"
