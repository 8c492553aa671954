#!/bin/bash
# visit 46: channel slices for the backward of the warps on few-pixel maps (blocks wanted per launch; 0 = no slices)
timeout 200 python -m pytest tests/test_kernels_motion.py -m gpu -x -q 2>&1 | tail -1
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v46 "" "MNK_DEFORM_BWD_BLOCKS=${A:-0}" "MNK_DEFORM_BWD_BLOCKS=${B:-2048}"
