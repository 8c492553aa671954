#!/bin/bash
# visit 26: longer pixel chunks for the long layers of the grouped tap-major launches
OUT=gpurun_out/r02v26; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v26/ab "" "MNK_WGROUP_LONG=2" "MNK_WGROUP_LONG=4" "MNK_WGROUP_LONG=2,MNK_WGROUP_LONG_FROM=32" "MNK_WGROUP_LONG=2,MNK_WGROUP_LONG_FROM=512" 2>&1 | tee "$OUT/summary.txt"
