#!/bin/bash
# FETCH_SIZE pass of the eager iteration with and without the XCD-aware tile order: gpu_fetch_ab.sh TAG
TAG="${1:-fetchab}"; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 2 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
for x in 0 1; do
  ( cd /tmp && MNK_XCD_REMAP=$x timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$OUT/f$x" -o f -- $CMD > "$OLDPWD/$OUT/f$x.log" 2>&1 )
  ( cd /tmp && MNK_XCD_REMAP=$x timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/$OUT/w$x" -o w -- $CMD > "$OLDPWD/$OUT/w$x.log" 2>&1 )
  echo "== MNK_XCD_REMAP=$x" | tee -a "$OUT/summary.txt"
  python tools/pmc_summarize.py "$OUT/f$x" "$OUT/w$x" "$OUT/pmc_$x.json" 2>&1 | grep "conv3x3\|pack\|colsum\|bn_" | tee -a "$OUT/summary.txt"
  find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*counter_collection*" -size +8M -delete
done
