#!/bin/bash
# visit 21: loss tail as two custom functions (image L1, LSGAN terms) + batch means in one reduction
OUT=gpurun_out/r02v21; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_losses.py tests/test_step.py tests/test_fullsize_oracle.py -q -m gpu 2>&1 | tail -3
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v21/ab "" 2>&1 | tee "$OUT/summary.txt"
CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python tools/trace_groups.py "$t" --csv "$OUT/steady.csv" > "$OUT/steady_groups.txt" 2>&1
rm -rf "$OUT/prof"; grep "at::\|kernel time\|rocclr" "$OUT/steady_groups.txt" | head -14 | cut -c1-150
