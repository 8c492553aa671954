#!/bin/bash
# visit 22: capture warm-up undone (first graph step = one update): replay test, bench
OUT=gpurun_out/r02v22; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fullsize.py tests/test_step.py -q -m gpu 2>&1 | tail -5
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v22/ab "" 2>&1 | tee "$OUT/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
