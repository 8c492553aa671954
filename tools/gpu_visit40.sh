#!/bin/bash
# visit 40 / 41: second-stage column sums with four / eight partials in flight; split partials fetched ahead in bn_small_fwd: BN kernel tests, the step, per-kernel times
OUT=gpurun_out/v40; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_bn.py tests/test_kernels_conv.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-profile 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/steady_kernel_stats.csv" > "$OUT/steady_groups.txt" 2>&1
head -4 "$OUT/steady_groups.txt"; grep "colsum2\|bn_final\|splitk_reduce\|bn_act\|bn_small" "$OUT/steady_groups.txt" | cut -c1-130
find "$OUT" -name "*kernel_trace*" -size +4M -delete
