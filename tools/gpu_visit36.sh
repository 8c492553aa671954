#!/bin/bash
# visit 36: frame sizes / batch sizes off the beaten path (odd bottleneck maps, maps that are no multiple of 8)
for args in "--size 96 --batch 8" "--size 160 --batch 3" "--size 64 --batch 5" "--size 128 --batch 7 --config taichi" "--size 32 --batch 16"; do
  echo "=== $args"; python tools/train_sanity.py $args --steps 60 --graph 1 --every 30 --seed 4 2>&1 | grep "iteration\|reconstruction\|evaluation\|^ok\|Error\|error" | cut -c1-170
done
