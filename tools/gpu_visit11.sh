#!/bin/bash
# visit 11: launch-plan sweep of the forward / data-gradient GEMM (tools/plan_tune.py), n16 weight-gradient split knobs
OUT=gpurun_out/r02v11; mkdir -p "$OUT"; export TMPDIR=/tmp
for c in moving-gif taichi; do
  timeout 900 python tools/plan_tune.py --config $c --batch 32 > "$OUT/plan_tune_$c.txt" 2> "$OUT/plan_tune_$c.err"; echo "plan_tune $c rc=$?"
  grep "^# rows" "$OUT/plan_tune_$c.txt"
done
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v11/ab "" "MNK_WN16_TARGET=256" "MNK_WN16_TARGET=1024" "MNK_WN16_MINTILES=4" "MNK_WN16_MINTILES=8" "MNK_WHALO_TARGET=384" 2>&1 | tee "$OUT/summary.txt"
