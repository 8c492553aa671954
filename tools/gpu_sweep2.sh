#!/bin/bash
TAG="${1:-sweep2}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/summary.txt"
run() {
  cfg=$1; shift
  echo "== $cfg $*" | tee -a "$OUT/summary.txt"
  f="$OUT/cb_${cfg}_$(echo "$*" | tr ' =' '__').txt"
  env "$@" timeout 300 python tools/conv_bench.py --config $cfg --batch 32 > "$f" 2>&1
  grep TOTAL "$f" | tee -a "$OUT/summary.txt"
}
for cfg in taichi moving-gif; do
run $cfg MNK_BM64_TILES=0
run $cfg MNK_BM64_TILES=512
run $cfg MNK_BM64_TILES=1024
run $cfg MNK_BM64_TILES=1024 MNK_SPLIT_TILES=256
done
for a in 0 1; do
echo "== bench MNK_WGRAD_ATOMIC=$a" | tee -a "$OUT/summary.txt"
MNK_WGRAD_ATOMIC=$a timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_atomic$a.json" 2>/dev/null
python -c "
import json;d=json.load(open('$OUT/bench_atomic$a.json'));print(d['ms_per_step'], {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items() if 'conv' in k})" | tee -a "$OUT/summary.txt"
done
