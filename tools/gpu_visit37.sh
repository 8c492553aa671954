#!/bin/bash
# visit 37: every shipped configuration at its native frame size, short fixed-batch run + evaluation forward
for args in "--config moving-gif --size 128 --batch 8" "--config shapes --size 64 --batch 16" "--config taichi --size 64 --batch 16" "--config bair --size 64 --batch 16" "--config vox --size 256 --batch 2"; do
  echo "=== $args"; python tools/train_sanity.py $args --steps 50 --graph 1 --every 25 --seed 6 2>&1 | grep "iteration\|reconstruction\|evaluation\|^ok\|Error\|error" | cut -c1-170
done
