#!/bin/bash
# visit 38: split-K reduction that leaves the BatchNorm statistics (MNK_SPLITK_STATS) and the last-block second stage of the
# column sums (MNK_BN_TAIL: bit 0 statistics kernels, bit 1 the dy pass): kernel / module / full-size parity tests, then the A/B
mkdir -p gpurun_out/v38
timeout 400 python -m pytest tests/test_kernels_bn.py tests/test_kernels_conv.py tests/test_modules.py tests/test_fullsize.py tests/test_fullsize_oracle.py -m gpu -x -q > gpurun_out/v38/pytest.log 2>&1
tail -4 gpurun_out/v38/pytest.log
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v38 "" "MNK_BN_TAIL=0" "MNK_BN_TAIL=1" "MNK_SPLITK_STATS=0" "MNK_SPLITK_STATS=0,MNK_BN_TAIL=0"
