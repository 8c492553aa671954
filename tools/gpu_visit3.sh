#!/bin/bash
# round-2 visit 3: all -m gpu tests, A/B grouped wgrad / hand-written pipeline / round-1 pipeline, steady-state kernel trace
OUT=gpurun_out/r02v3; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=5 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/summary.txt"
tail -14 "$OUT/pytest_gpu.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v3/ab "" "MNK_WGRAD_GROUPED=0" "MNK_HAND_ADAM=0" "MNK_WGROUP_CHUNK=2048" "MNK_WGROUP_CHUNK=512" 2>&1 | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/steady_kernel_stats.csv" > "$OUT/steady_groups.txt" 2>&1
head -75 "$OUT/steady_groups.txt" | cut -c1-150 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -size +4M -delete
