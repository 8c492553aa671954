#!/bin/bash
# visit 23: rows per thread / block cap of the two-stage BatchNorm kernels
OUT=gpurun_out/r02v23; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v23/ab "" "MNK_BN_RPT=4" "MNK_BN_RPT=2" "MNK_BN_RPT=1" "MNK_BN_RPT=4,MNK_BN_BLOCKS=2048" "MNK_BN_RPT=2,MNK_BN_BLOCKS=2048" "MNK_BN_RPT=16" 2>&1 | tee "$OUT/summary.txt"
