#!/bin/bash
# visit 24 (experiment): appearance encoder on a second stream next to the dense-motion network
OUT=gpurun_out/r02v24; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v24/ab "" "MNK_TWO_STREAMS=1" 2>&1 | tee "$OUT/summary.txt"
MNK_TWO_STREAMS=1 timeout 600 python -m pytest tests/test_fullsize_oracle.py tests/test_fullsize.py -q -m gpu -x 2>&1 | tail -3
grep -h "capture\|Error\|error" $OUT/ab/*TWO_STREAMS*err | head -5
