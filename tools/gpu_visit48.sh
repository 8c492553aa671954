#!/bin/bash
# visit 48: every split-K reduction as float4 rows (the statistics kernel without its statistics) against the 64 x 4-group kernel
timeout 300 python -m pytest tests/test_kernels_conv.py -m gpu -x -q 2>&1 | tail -1
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v48 "" "MNK_REDUCE_V4=0" "MNK_RS_RPT=4" "MNK_RS_RPT=1"
