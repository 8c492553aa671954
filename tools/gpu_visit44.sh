#!/bin/bash
# visit 44: backward of the warps with one lane per channel (atomic adds to consecutive floats) against one lane per channel quad
timeout 200 python -m pytest tests/test_kernels_motion.py -m gpu -x -q 2>&1 | tail -1
REPS=3 STEPS=40 bash tools/gpu_knob_ab.sh v44 "" "MNK_DEFORM_BWD_CHAN=0"
