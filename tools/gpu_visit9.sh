#!/bin/bash
# round-2 visit 9: small-layer BatchNorm forms / zero bias gradients: tests + A/B on the whole iteration
OUT=gpurun_out/r02v9; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=3 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/summary.txt"
tail -6 "$OUT/pytest_gpu.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v9/ab "" "MNK_BN_SMALL=0" "MNK_BN_ZERO_BIAS_GRAD=0,MNK_BN_SMALL=0" 2>&1 | tee -a "$OUT/summary.txt"
REPS=1 STEPS=30 BENCH_ARGS="--config taichi" bash tools/gpu_knob_ab.sh r02v9/ab_taichi "" "MNK_BN_SMALL=0" 2>&1 | tee -a "$OUT/summary.txt"
