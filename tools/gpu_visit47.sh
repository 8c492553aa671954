#!/bin/bash
# visit 47: row limit of the one-launch BatchNorm forms again, now that the forward kernel runs 1024 threads over full lines
REPS=2 STEPS=40 bash tools/gpu_knob_ab.sh v47 "" "MNK_BN_SMALL_ROWS=1024" "MNK_BN_SMALL_ROWS=2048" "MNK_BN_SMALL_ROWS=1024,MNK_BN_SMALL_BWD_SHAPE=1"
