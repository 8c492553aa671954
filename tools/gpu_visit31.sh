#!/bin/bash
# visit 31: watch for the first non-finite tensor over many seeds (eager, checked every iteration), then graph runs without syncs
bad=0
for seed in $(seq 1 24); do
  out=$(python tools/train_sanity.py --steps 80 --graph 0 --every 1000 --watch 1 --seed $seed 2>&1 | tail -3 | tr '\n' ' ')
  case "$out" in *ok*) ;; *) bad=$((bad+1)); echo "seed $seed eager: $(echo $out | cut -c1-600)";; esac
done
echo "eager watched: $bad of 24 seeds non-finite"
bad=0
for seed in $(seq 1 24); do
  out=$(python tools/train_sanity.py --steps 80 --graph 1 --every 1000 --seed $seed 2>&1 | tail -2 | tr '\n' ' ')
  case "$out" in *ok*) ;; *) bad=$((bad+1)); echo "seed $seed graph: $(echo $out | cut -c1-300)";; esac
done
echo "graph unsynced: $bad of 24 seeds non-finite"
