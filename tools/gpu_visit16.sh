#!/bin/bash
# visit 16: row tiles per block of the many-layer split reduction
OUT=gpurun_out/r02v16; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v16/ab "" "MNK_REDUCE_RPT=2" "MNK_REDUCE_RPT=4" "MNK_REDUCE_RPT=8" "MNK_REDUCE_RPT=16" 2>&1 | tee "$OUT/summary.txt"
