#!/bin/bash
TAG="${1:-sweep}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
run() {
  echo "== $*" | tee -a "$OUT/summary.txt"
  env "$@" timeout 300 python tools/conv_bench.py --config ${CFG:-taichi} --batch 32 > "$OUT/cb_$(echo "$*" | tr ' =' '__').txt" 2>&1
  grep TOTAL "$OUT/cb_$(echo "$*" | tr ' =' '__').txt" | tee -a "$OUT/summary.txt"
}
: > "$OUT/summary.txt"
run A=default
run MNK_SPLIT_TILES=256 MNK_SPLIT_TARGET=768
run MNK_SPLIT_TILES=384 MNK_SPLIT_TARGET=1024
run MNK_SPLIT_TILES=192 MNK_SPLIT_TARGET=512 MNK_SPLIT_MINSTEPS=12
run MNK_WSPLIT_TILES=256 MNK_WSPLIT_TARGET=512 MNK_WSPLIT_MINSTEPS=16
run MNK_WSPLIT_TILES=1024 MNK_WSPLIT_TARGET=2048 MNK_WSPLIT_MINSTEPS=8
run MNK_WSPLIT_TILES=512 MNK_WSPLIT_TARGET=768 MNK_WSPLIT_MINSTEPS=8
cp "$OUT/cb_A_default.txt" "$OUT/conv_bench_default.txt"
echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; cat "$OUT/bench.json" | cut -c1-330 | tee -a "$OUT/summary.txt"
