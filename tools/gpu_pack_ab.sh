#!/bin/bash
# A/B of the per-iteration weight packing inside ONE box: one pack_multi launch vs one pack_all launch per layer
OUT=gpurun_out/${1:-packab}; mkdir -p "$OUT"
timeout 200 python -m pytest tests/test_kernels_conv.py tests/test_modules.py -x -q -m gpu -k "pack or version_bump" > "$OUT/tests.log" 2>&1; tail -2 "$OUT/tests.log"
for rep in 1 2; do
  for m in 1 0; do
    MNK_PACK_MULTI=$m timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/bench_multi${m}_$rep.json" 2> "$OUT/bench_multi${m}_$rep.err"
    echo "multi=$m rep=$rep $(grep -o '"ms_per_step": [0-9.]*' "$OUT/bench_multi${m}_$rep.json")"
  done
done
