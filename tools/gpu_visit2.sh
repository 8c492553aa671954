#!/bin/bash
# round-2 visit 2: all -m gpu tests, A/B of the hand-written optimiser pipeline, rocprofv3 steady-state kernel stats
OUT=gpurun_out/r02v2; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/summary.txt"
tail -30 "$OUT/pytest_gpu.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v2/ab "" "MNK_HAND_ADAM=0" 2>&1 | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/steady_kernel_stats.csv" > "$OUT/steady_groups.txt" 2>&1
head -70 "$OUT/steady_groups.txt" | cut -c1-150 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -size +4M -delete
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-900 "$OUT/bench.json" | tee -a "$OUT/summary.txt"
