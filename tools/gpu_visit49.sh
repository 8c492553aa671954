#!/bin/bash
# visit 49: the final build on a fixed batch for 400 captured iterations (two seeds) and the taichi configuration for 100
for args in "--seed 11 --steps 400" "--seed 3 --steps 400" "--config taichi --size 64 --batch 32 --seed 5 --steps 100"; do
  echo "=== $args"; timeout 120 python tools/train_sanity.py $args --graph 1 --every 100 2>&1 | grep "iteration\|reconstruction\|evaluation\|^ok\|Error\|error\|nan" | cut -c1-170
done
