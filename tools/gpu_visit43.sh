#!/bin/bash
# visit 43: the small-layer launch shape for the backward kernel too (MNK_BN_SMALL_BWD_SHAPE), rows per thread of the split-K
# reduction with statistics (MNK_RS_RPT)
timeout 200 python -m pytest tests/test_kernels_bn.py -m gpu -x -q 2>&1 | tail -1
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v43 "" "MNK_BN_SMALL_BWD_SHAPE=0" "MNK_RS_RPT=2" "MNK_RS_RPT=1" "MNK_RS_RPT=8"
