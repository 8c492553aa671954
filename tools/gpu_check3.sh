#!/bin/bash
TAG="${1:-r01c}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== conv_bench taichi" | tee "$OUT/summary.txt"
timeout 600 python tools/conv_bench.py --config taichi --batch 32 > "$OUT/conv_bench_taichi.txt" 2>&1; echo "rc=$?" | tee -a "$OUT/summary.txt"
grep -v amdgpu.ids "$OUT/conv_bench_taichi.txt" | tee -a "$OUT/summary.txt"
echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -4 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
echo "== bench moving-gif" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"
echo "== bench taichi" | tee -a "$OUT/summary.txt"
timeout 400 python bench.py --config taichi --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_taichi.json" 2> "$OUT/bench_taichi.err"
cat "$OUT/bench_taichi.json" | tee -a "$OUT/summary.txt"
if [ "${ROCPROF:-1}" = "1" ]; then
echo "== rocprofv3 kernel stats (bench.py --graph 0, moving-gif)" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -type f | head | tee -a "$OUT/summary.txt"
f=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -30 "$f" | cut -c1-220 | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_trace*" -size +4M -delete
fi
