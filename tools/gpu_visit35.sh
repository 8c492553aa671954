#!/bin/bash
# visit 35: fixed-batch runs of the other configurations + evaluation forward with the trained weights
for c in moving-gif taichi bair vox; do
  echo "=== $c"; python tools/train_sanity.py --config $c --steps 150 --graph 1 --every 75 --seed 2 2>&1 | grep "iteration\|reconstruction\|evaluation\|ok\|Error\|error" | cut -c1-150
done
echo "=== vox 256 batch 4"; python tools/train_sanity.py --config vox --size 256 --batch 4 --steps 40 --graph 0 --every 20 --seed 2 2>&1 | grep "iteration\|reconstruction\|evaluation\|ok\|Error\|error" | cut -c1-150
