#!/bin/bash
# visit 42: one-launch BatchNorm forward of the small layers: threads per block x channel quads per block (the strided reads
# of one quad per block use 16 of every 128 bytes they touch; more quads per block = fewer blocks)
T=MNK_BN_SMALL_FWD_THREADS; X=MNK_BN_SMALL_FWD_TXN
timeout 200 python -m pytest tests/test_kernels_bn.py -m gpu -x -q -k small 2>&1 | tail -1
$T=1024 $X=8 timeout 200 python -m pytest tests/test_kernels_bn.py -m gpu -x -q -k small 2>&1 | tail -1
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v42 "" "$T=1024,$X=8" "$T=1024,$X=4" "$T=1024,$X=2" "$T=512,$X=4" "$T=1024,$X=1"
