#!/bin/bash
for ev in 50 50 25 10; do echo "=== graph every $ev"; python tools/train_sanity.py --steps 101 --graph 1 --every $ev 2>&1 | grep "iteration\|Error\|ok" | cut -c1-150; done
echo "=== eager every 50"; python tools/train_sanity.py --steps 101 --graph 0 --every 50 2>&1 | grep "iteration\|Error\|ok" | cut -c1-150
echo "=== graph every 50, torch adam"; MNK_HAND_ADAM=0 python tools/train_sanity.py --steps 101 --graph 1 --every 50 2>&1 | grep "iteration\|Error\|ok" | cut -c1-150
