#!/bin/bash
TAG="${1:-r01pmc}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 2 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
echo "== single-rank RCCL exercise (MNK_DIST_FORCE=1)" | tee "$OUT/summary.txt"
MNK_DIST_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > "$OUT/bench_dist1.json" 2> "$OUT/bench_dist1.err"; echo "rc=$?" | tee -a "$OUT/summary.txt"
cut -c1-330 "$OUT/bench_dist1.json" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/bench_dist1.err" | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== pmc FETCH_SIZE" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_fetch" -o f -- $CMD > "$OLDPWD/$OUT/pmc_fetch.log" 2>&1 ); echo "rc=$?" | tee -a "$OUT/summary.txt"
echo "== pmc WRITE_SIZE" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_write" -o w -- $CMD > "$OLDPWD/$OUT/pmc_write.log" 2>&1 ); echo "rc=$?" | tee -a "$OUT/summary.txt"
find "$OUT" -name "*.csv" | head | tee -a "$OUT/summary.txt"
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json" 2>&1 | tee -a "$OUT/summary.txt"
f=$(find "$OUT/pmc_fetch" -name "*counter_collection*.csv" | head -1); [ -n "$f" ] && head -3 "$f" | cut -c1-400 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -size +4M -delete; find "$OUT" -name "*counter_collection*" -size +20M -delete
