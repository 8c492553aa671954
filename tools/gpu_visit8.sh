#!/bin/bash
# round-2 visit 6: sub-pixel up-sampled convolutions: tests, per-layer conv bench on / off, whole-iteration A/B
OUT=gpurun_out/r02v8; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=3 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/summary.txt"
tail -8 "$OUT/pytest_gpu.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
for v in 1 0; do
  MNK_UP_SUBPIXEL=$v timeout 300 python tools/conv_bench.py --config moving-gif --batch 32 > "$OUT/conv_bench_moving-gif_subpixel$v.txt" 2>&1
  grep -E "dec|TOTAL" "$OUT/conv_bench_moving-gif_subpixel$v.txt" | tee -a "$OUT/summary.txt"
done
MNK_UP_SUBPIXEL=1 timeout 300 python tools/conv_bench.py --config taichi --batch 32 > "$OUT/conv_bench_taichi_subpixel1.txt" 2>&1; grep TOTAL "$OUT/conv_bench_taichi_subpixel1.txt" | tee -a "$OUT/summary.txt"
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v8/ab "" "MNK_UP_SUBPIXEL=0" "MNK_WGROUP_CHUNK=128" "MNK_WGROUP_CHUNK=512" 2>&1 | tee -a "$OUT/summary.txt"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; python - <<'P' | tee -a "$OUT/summary.txt"
import json
try:
    r = json.load(open("gpurun_out/r02v8/bench.json"))
    print({k: r[k] for k in ("value", "ms_per_step", "hot_path_only_ms", "hot_path_only")}, r["roofline"]["frac"], r["roofline"]["achieved"])
    for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]): print("  %-22s %6.1f launches %7.3f ms" % (k, v["launches_per_step"], v["ms_per_step"]))
except Exception as e:
    print("bench.json:", e)
P
tail -3 "$OUT/bench.err" | cut -c1-300 | tee -a "$OUT/summary.txt"
