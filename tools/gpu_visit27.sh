#!/bin/bash
# visit 27: BASELINE config 3's per-GPU share on one GPU (vox 256x256, batch 8): step time, roofline, per-layer conv bench
OUT=gpurun_out/r02v27; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python bench.py --config vox --size 256 --batch 8 --no-cpu-baseline > "$OUT/bench_vox256_b8.json" 2> "$OUT/bench_vox.err"; echo "rc=$?"; cut -c1-700 "$OUT/bench_vox256_b8.json"; tail -3 "$OUT/bench_vox.err" | cut -c1-300
timeout 300 python tools/conv_bench.py --config vox --batch 8 --size 256 > "$OUT/conv_bench_vox256_b8.txt" 2>&1; grep TOTAL "$OUT/conv_bench_vox256_b8.txt"
