#!/bin/bash
# visit 14: 64x64-tile rule + its split rule against the previous rule (whole step), residual of the plan sweep, in-situ trace
OUT=gpurun_out/r02v14; mkdir -p "$OUT"; export TMPDIR=/tmp
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v14/ab "" "MNK_BN128_KWORK=-1" "MNK_SPLIT64_TARGET=1536" "MNK_SPLIT64_TARGET=768" 2>&1 | tee "$OUT/summary.txt"
BENCH_ARGS="--config taichi" REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v14/ab_taichi "" "MNK_BN128_KWORK=-1" 2>&1 | tee -a "$OUT/summary.txt"
for c in moving-gif taichi; do
  timeout 900 python tools/plan_tune.py --config $c --batch 32 > "$OUT/plan_tune_$c.txt" 2> "$OUT/plan_tune_$c.err"; echo "plan_tune $c rc=$?"
  grep "^# rows" "$OUT/plan_tune_$c.txt" | tee -a "$OUT/summary.txt"
done
for c in moving-gif taichi; do timeout 300 python tools/conv_bench.py --config $c --batch 32 > "$OUT/conv_bench_${c}_b32.txt" 2>&1; grep TOTAL "$OUT/conv_bench_${c}_b32.txt" | tee -a "$OUT/summary.txt"; done
CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python tools/trace_groups.py "$t" --csv "$OUT/steady.csv" > "$OUT/steady_groups.txt" 2>&1
python - "$t" "$OUT/trace.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    for r in rows:
        f.write("%s,%s,%s,%s,%s,%d\n" % (r["Kernel_Name"].replace(",", ";")[:90], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"],
                                     r["Start_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
P
rm -rf "$OUT/prof"; head -3 "$OUT/steady_groups.txt"
