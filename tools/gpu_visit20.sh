#!/bin/bash
# visit 20: wider blocks for the per-frame reduction kernels (key-point soft-argmax, motion field backward, pair L1)
OUT=gpurun_out/r02v20; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_motion.py tests/test_kernels_keypoints.py tests/test_kernels_losses.py -q -m gpu 2>&1 | tail -2
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v20/ab "" 2>&1 | tee "$OUT/summary.txt"
CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python tools/trace_groups.py "$t" --csv "$OUT/steady.csv" > "$OUT/steady_groups.txt" 2>&1
rm -rf "$OUT/prof"; grep "softmax_kp\|motion_field\|pair_l1\|kernel time" "$OUT/steady_groups.txt" | head -12 | cut -c1-120
