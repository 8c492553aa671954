#!/bin/bash
# visit 30: how often does a 200-iteration fixed-batch run go non-finite? graph replay without host syncs vs eager
for mode in 1 0; do
  bad=0
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    out=$(python tools/train_sanity.py --steps 200 --graph $mode --every 1000 2>&1 | tail -2 | tr '\n' ' ')
    case "$out" in *ok*) ;; *) bad=$((bad+1)); echo "run $i graph=$mode: $(echo $out | cut -c1-200)";; esac
  done
  echo "graph=$mode: $bad of 12 runs non-finite"
done
