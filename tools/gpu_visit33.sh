#!/bin/bash
# visit 33: after the stable sigma_min: the seed that went non-finite, 24 seeds (graph replay without host syncs), kernel tests
timeout 600 python -m pytest tests/test_kernels_keypoints.py tests/test_fullsize_oracle.py tests/test_modules.py -q -m gpu 2>&1 | tail -2
for w in none; do python tools/nan_diag.py --seed 11 --steps 8 --wrap $w 2>&1 | grep "losses" | cut -c1-100; done
bad=0
for seed in $(seq 1 24); do
  out=$(python tools/train_sanity.py --steps 120 --graph 1 --every 1000 --seed $seed 2>&1 | tail -2 | tr '\n' ' ')
  case "$out" in *ok*) ;; *) bad=$((bad+1)); echo "seed $seed graph: $(echo $out | cut -c1-300)";; esac
done
echo "graph unsynced, 120 iterations: $bad of 24 seeds non-finite"
