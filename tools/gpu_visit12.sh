#!/bin/bash
# visit 12: 64x64-tile rule of the forward / data-gradient GEMM against the previous rule; residual of the plan sweep
OUT=gpurun_out/r02v12; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v12/ab "" "MNK_BN128_KWORK=-1" "MNK_BN128_KWORK=16800" 2>&1 | tee "$OUT/summary.txt"
BENCH_ARGS="--config taichi" REPS=1 STEPS=30 bash tools/gpu_knob_ab.sh r02v12/ab_taichi "" "MNK_BN128_KWORK=-1" 2>&1 | tee -a "$OUT/summary.txt"
for c in moving-gif taichi; do
  timeout 900 python tools/plan_tune.py --config $c --batch 32 > "$OUT/plan_tune_$c.txt" 2> "$OUT/plan_tune_$c.err"; echo "plan_tune $c rc=$?"
  grep "^# rows" "$OUT/plan_tune_$c.txt" | tee -a "$OUT/summary.txt"
done
for c in moving-gif taichi; do timeout 300 python tools/conv_bench.py --config $c --batch 32 > "$OUT/conv_bench_${c}_b32.txt" 2>&1; grep TOTAL "$OUT/conv_bench_${c}_b32.txt" | tee -a "$OUT/summary.txt"; done
