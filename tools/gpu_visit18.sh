#!/bin/bash
# visit 18: many-layer split reduction with every element of the parameter-major form in flight: step time + kernel trace
OUT=gpurun_out/r02v18; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v18/ab "" 2>&1 | tee "$OUT/summary.txt"
CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python tools/trace_groups.py "$t" --csv "$OUT/steady.csv" > "$OUT/steady_groups.txt" 2>&1
rm -rf "$OUT/prof"; head -40 "$OUT/steady_groups.txt" | cut -c1-120
