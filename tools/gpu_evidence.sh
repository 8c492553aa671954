#!/bin/bash
# One GPU-box visit that collects everything profiles/ cites: PMC traffic, bench lines (moving-gif with roofline +
# cpu_baseline, taichi), rocprofv3 kernel stats (+ steady-state window), per-layer conv bench, batched inference,
# single-rank RCCL exercise of the distributed path.  Usage: gpu_evidence.sh TAG
TAG="${1:-r01final}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
S="$OUT/summary.txt"; : > "$S"
CMD="python $PWD/bench.py --steps 2 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
if [ "${SKIP_PMC:-0}" != "1" ]; then
echo "== pmc FETCH_SIZE / WRITE_SIZE (separate passes)" | tee -a "$S"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_fetch" -o f -- $CMD > "$OLDPWD/$OUT/pmc_fetch.log" 2>&1 ); echo "fetch rc=$?" | tee -a "$S"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_write" -o w -- $CMD > "$OLDPWD/$OUT/pmc_write.log" 2>&1 ); echo "write rc=$?" | tee -a "$S"
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic_moving-gif_b32.json" 2>&1 | tee -a "$S"
[ -s "$OUT/pmc_traffic_moving-gif_b32.json" ] && cp "$OUT/pmc_traffic_moving-gif_b32.json" profiles/r01_pmc_traffic_moving-gif_b32.json
find "$OUT" -name "*kernel_trace*" -size +4M -delete; find "$OUT" -name "*counter_collection*" -size +8M -delete
fi
if [ "${RUN_TESTS:-0}" = "1" ]; then
echo "== pytest -m gpu" | tee -a "$S"
timeout 600 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" | tee -a "$S"; tail -2 "$OUT/pytest_gpu.log" | tee -a "$S"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a "$S"
fi
echo "== bench (default: moving-gif, roofline + cpu_baseline)" | tee -a "$S"
timeout 900 python bench.py > "$OUT/bench_moving-gif_b32.json" 2> "$OUT/bench.err"; echo "rc=$?" | tee -a "$S"
cut -c1-700 "$OUT/bench_moving-gif_b32.json" | tee -a "$S"
timeout 400 python bench.py --config taichi --no-cpu-baseline > "$OUT/bench_taichi_b32.json" 2> "$OUT/bench_taichi.err"; echo "taichi rc=$?" | tee -a "$S"
cut -c1-300 "$OUT/bench_taichi_b32.json" | tee -a "$S"
echo "== rocprofv3 kernel stats (eager iteration)" | tee -a "$S"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$S"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/moving-gif_b32_eager_kernel_stats.csv"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/moving-gif_b32_steady_kernel_stats.csv" > "$OUT/moving-gif_b32_steady_groups.txt" 2>&1
head -40 "$OUT/moving-gif_b32_steady_groups.txt" | cut -c1-130 | tee -a "$S"
find "$OUT" -name "*kernel_trace*" -size +4M -delete
echo "== per-layer conv bench" | tee -a "$S"
for c in moving-gif taichi; do timeout 300 python tools/conv_bench.py --config $c --batch 32 > "$OUT/conv_bench_${c}_b32.txt" 2>&1; grep TOTAL "$OUT/conv_bench_${c}_b32.txt" | tee -a "$S"; done
echo "== batched inference (bair, B=512)" | tee -a "$S"
timeout 300 python tools/infer_bench.py > "$OUT/infer_bair_b512.json" 2> "$OUT/infer.err"; cat "$OUT/infer_bair_b512.json" | tee -a "$S"
echo "== single-rank RCCL exercise (MNK_DIST_FORCE=1, torch.distributed.run)" | tee -a "$S"
MNK_DIST_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_dist1.json" 2> "$OUT/bench_dist1.err"; echo "rc=$?" | tee -a "$S"
cut -c1-330 "$OUT/bench_dist1.json" | tee -a "$S"; tail -3 "$OUT/bench_dist1.err" | cut -c1-200 | tee -a "$S"
echo "== the same with the graph phase's deadline forced to expire: the eager line must still come out, rc 0" | tee -a "$S"
MNK_GRAPH_DEADLINE_S=0.05 MNK_DIST_FORCE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile > "$OUT/bench_dist1_deadline.json" 2> "$OUT/bench_dist1_deadline.err"; echo "rc=$?" | tee -a "$S"
cut -c1-330 "$OUT/bench_dist1_deadline.json" | tee -a "$S"; grep -h "deadline" "$OUT/bench_dist1_deadline.err" | cut -c1-200 | tee -a "$S"
