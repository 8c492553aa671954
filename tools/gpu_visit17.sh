#!/bin/bash
# visit 17: split count from which 16 thread groups share a row in the many-layer split reduction
OUT=gpurun_out/r02v17; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v17/ab "" "MNK_REDUCE_TW16=16" "MNK_REDUCE_TW16=8" "MNK_REDUCE_TW16=4" "MNK_REDUCE_TW16=64" 2>&1 | tee "$OUT/summary.txt"
