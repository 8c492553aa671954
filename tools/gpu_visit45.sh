#!/bin/bash
# visit 45: key-point channels per block of the soft-argmax kernels (16 = a block per frame: 64 blocks of one wave per SIMD)
timeout 200 python -m pytest tests/test_kernels_keypoints.py -m gpu -x -q 2>&1 | tail -1
MNK_KP_GROUP=2 timeout 200 python -m pytest tests/test_kernels_keypoints.py -m gpu -x -q -k softmax 2>&1 | tail -1
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v45 "" "MNK_KP_GROUP=5" "MNK_KP_GROUP=2" "MNK_KP_GROUP=1"
