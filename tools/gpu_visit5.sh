#!/bin/bash
# round-2 visit 4: all -m gpu tests, pipeline A/B, single-rank RCCL exercise (direct communicator vs torch.distributed), trace
OUT=gpurun_out/r02v5; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=5 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/summary.txt"
tail -14 "$OUT/pytest_gpu.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
REPS=2 STEPS=30 bash tools/gpu_knob_ab.sh r02v5/ab "" "MNK_WGRAD_GROUPED=0" "MNK_HAND_ADAM=0" "MNK_WGROUP_CHUNK=256" "MNK_WGROUP_CHUNK=1024" 2>&1 | tee -a "$OUT/summary.txt"
for v in "MNK_RCCL_DIRECT=1" "MNK_RCCL_DIRECT=0"; do
  env $v MNK_DIST_FORCE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > "$OUT/bench_dist1_$v.json" 2> "$OUT/bench_dist1_$v.err"; echo "dist1 $v rc=$?" | tee -a "$OUT/summary.txt"
  cut -c1-420 "$OUT/bench_dist1_$v.json" | tee -a "$OUT/summary.txt"; grep -h "mnk.dist\|Error\|error" "$OUT/bench_dist1_$v.err" | head -5 | cut -c1-200 | tee -a "$OUT/summary.txt"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 6 --warmup 3 --graph 1 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$OUT/summary.txt"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/steady_kernel_stats.csv" > "$OUT/steady_groups.txt" 2>&1
head -75 "$OUT/steady_groups.txt" | cut -c1-150 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -size +4M -delete
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; python - <<'P' | tee -a "$OUT/summary.txt"
import json
try:
    r = json.load(open("gpurun_out/r02v5/bench.json"))
    print({k: r[k] for k in ("value", "ms_per_step", "hot_path_only_ms", "hot_path_only")}, r["roofline"]["frac"])
    for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]): print("  %-22s %6.1f launches %7.3f ms" % (k, v["launches_per_step"], v["ms_per_step"]))
except Exception as e:
    print("bench.json:", e)
P
tail -5 "$OUT/bench.err" | cut -c1-300 | tee -a "$OUT/summary.txt"
