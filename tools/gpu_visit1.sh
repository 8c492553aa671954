#!/bin/bash
# round-2 visit 1: every -m gpu test (new full-size oracle tests included), then the fused feature-matching-loss A/B
OUT=gpurun_out/r02v1; mkdir -p "$OUT"
timeout 1100 python -m pytest tests -q -m gpu -x --durations=15 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/summary.txt"
tail -25 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v1/ab "" "MNK_FUSED_FM_LOSS=1" 2>&1 | tee -a "$OUT/summary.txt"
