#!/bin/bash
# visit 50: BatchNorm kernels with exactly nv lanes per row when the quad count is not a power of two (45 channels: 12 lanes, 192-thread
# blocks, instead of 16 lanes of which 4 idle) -- MNK_BN_EXACT_TX=0 is the previous map
timeout 300 python -m pytest tests/test_kernels_bn.py tests/test_modules.py tests/test_fullsize_oracle.py tests/test_train_sanity.py -m gpu -x -q 2>&1 | tail -1
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v50 "" "MNK_BN_EXACT_TX=0"
