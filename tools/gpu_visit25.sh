#!/bin/bash
# visit 25: nine-tap 16x16 weight-gradient kernel grouped over the narrow layers (blocks per layer), against per-layer launches
OUT=gpurun_out/r02v25; mkdir -p "$OUT"; export TMPDIR=/tmp

REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v25/ab "" "MNK_WN16_GROUP_TARGET=128,MNK_WN16_GROUP_FEW=512" "MNK_WN16_GROUP_TARGET=192,MNK_WN16_GROUP_FEW=512" "MNK_WN16_GROUP_TARGET=256,MNK_WN16_GROUP_FEW=512" "MNK_WN16_GROUP_TARGET=128,MNK_WN16_GROUP_FEW=256" "MNK_WN16_GROUP_TARGET=256,MNK_WN16_GROUP_FEW=384" 2>&1 | tee "$OUT/summary.txt"
CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python tools/trace_groups.py "$t" --csv "$OUT/steady.csv" > "$OUT/steady_groups.txt" 2>&1
rm -rf "$OUT/prof"; grep "n16\|reduce_multi\|kernel time" "$OUT/steady_groups.txt" | head -12 | cut -c1-130
