#!/bin/bash
timeout 600 python -m pytest tests/test_train_sanity.py -q -m gpu 2>&1 | tail -2
for seed in 11 5; do python tools/train_sanity.py --steps 1500 --graph 1 --every 500 --seed $seed 2>&1 | grep "iteration\|reconstruction\|ok\|Error" | cut -c1-140; done
