#!/bin/bash
# visit 39: last-block second stage again, with one device-scope release per block (by the ticket thread) and the acquire in the
# last block only -- visit 38's __threadfence() per thread cost +3.2 ms per iteration
mkdir -p gpurun_out/v39
timeout 300 python -m pytest tests/test_kernels_bn.py tests/test_modules.py tests/test_fullsize_oracle.py -m gpu -x -q > gpurun_out/v39/pytest.log 2>&1
tail -3 gpurun_out/v39/pytest.log
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v39 "" "MNK_BN_TAIL=0" "MNK_BN_TAIL=1" "MNK_BN_TAIL=2"
