#!/bin/bash
for w in none sync materialize check; do echo "=== wrap $w"; python tools/nan_diag.py --seed 11 --steps 7 --wrap $w 2>&1 | grep "losses" | cut -c1-110; done
echo "=== wrap none, MNK_HAND_ADAM=0"; MNK_HAND_ADAM=0 python tools/nan_diag.py --seed 11 --steps 7 --wrap none 2>&1 | grep "losses" | cut -c1-110
echo "=== wrap none, MNK_WGRAD_GROUPED=0"; MNK_WGRAD_GROUPED=0 python tools/nan_diag.py --seed 11 --steps 7 --wrap none 2>&1 | grep "losses" | cut -c1-110
