#!/bin/bash
# visit 28: where a fixed-batch run goes non-finite, under the switches that change the arithmetic
for v in "" "MNK_HAND_ADAM=0" "MNK_UP_SUBPIXEL=0" "MNK_BN_SMALL=0,MNK_BN_ZERO_BIAS_GRAD=0" "MNK_FUSED_FM_LOSS=0"; do
  echo "=== ${v:-default} (eager)"
  env $(echo "$v" | tr ',' ' ') python tools/train_sanity.py --steps 60 --graph 0 --every 4 2>&1 | grep "iteration\|reconstruction\|Error" | cut -c1-150
done
