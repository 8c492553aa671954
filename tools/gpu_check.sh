#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench (with roofline + cpu_baseline), rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box):  tools/gpu_check.sh [tag]
TAG="${1:-r01}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== smoke" | tee "$OUT/summary.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -40 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
echo "== bench" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --config taichi --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_taichi.json" 2> "$OUT/bench_taichi.err"
cat "$OUT/bench_taichi.json" | tee -a "$OUT/summary.txt"
echo "== rocprofv3 kernel stats" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -3 | tee -a "$OUT/summary.txt"
f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200 | tee -a "$OUT/summary.txt"
# keep the merged-back payload small: drop the raw per-dispatch trace, keep the stats
find "$OUT/prof" -name "*kernel_trace*" -size +8M -delete
