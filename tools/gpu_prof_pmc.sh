#!/bin/bash
# Short box visit: rocprofv3 kernel stats (+ steady-state window) and the two PMC traffic passes, each under its own
# timeout.  Usage: gpu_prof_pmc.sh TAG
TAG="${1:-r01prof}"; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; S="$OUT/summary.txt"; : > "$S"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$S"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/moving-gif_b32_eager_kernel_stats.csv"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/moving-gif_b32_steady_kernel_stats.csv" > "$OUT/moving-gif_b32_steady_groups.txt" 2>&1
head -12 "$OUT/moving-gif_b32_steady_groups.txt" | cut -c1-130 | tee -a "$S"
find "$OUT" -name "*kernel_trace*" -size +4M -delete
CMD="python $PWD/bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout ${PMC_TIMEOUT:-80} rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_fetch" -o f -- $CMD > "$OLDPWD/$OUT/pmc_fetch.log" 2>&1 ); echo "fetch rc=$?" | tee -a "$S"
( cd /tmp && timeout ${PMC_TIMEOUT:-80} rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_write" -o w -- $CMD > "$OLDPWD/$OUT/pmc_write.log" 2>&1 ); echo "write rc=$?" | tee -a "$S"
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic_moving-gif_b32.json" 2>&1 | tail -15 | tee -a "$S"
find "$OUT" -name "*kernel_trace*" -size +4M -delete; find "$OUT" -name "*counter_collection*" -size +8M -delete
