#!/bin/bash
# GPU visit: rocprofv3 kernel stats of the eager training iteration (+ kernel trace kept small) and a bench line
TAG="${1:-prof}"
CFG="${2:-moving-gif}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/summary.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --config $CFG --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$CFG.csv" && head -30 "$f" | cut -c1-160 | tee -a "$OUT/summary.txt"
# per-kernel-per-shape detail: group the trace by (kernel, grid) to find slow launches
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/steady_kernel_stats_$CFG.csv" > "$OUT/trace_groups_$CFG.txt" 2>&1
find "$OUT" -name "*kernel_trace*" -size +4M -delete
timeout 600 python bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_$CFG.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
cut -c1-400 "$OUT/bench_$CFG.json" | tee -a "$OUT/summary.txt"
