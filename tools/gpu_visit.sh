#!/bin/bash
# One short MI355X box visit, parameterised (replaces the per-visit scripts of rounds 1-2; their results are in
# profiles/r0*_knob_ab_log.txt).  Stages run in this order; each is optional:
#   TESTS="tests/test_kernels_bn.py ..."   pytest -m gpu -x -q on these files ("all" = the whole tests/ directory)
#   AB="|MNK_X=0|MNK_X=0,MNK_Y=2"          whole-iteration knob A/B (tools/gpu_knob_ab.sh; variants separated by '|', "" = defaults)
#   BENCH="--config taichi"                extra bench.py line(s), separated by ';'
#   TRACE=1                                rocprofv3 kernel trace of an eager iteration -> steady-state groups (tools/trace_groups.py)
#   CONVBENCH="moving-gif taichi"          per-layer conv bench of these configurations
#   CMD="python tools/foo.py"              any other command, output to $OUT/cmd.log
# Usage: [VARS] gpu_visit.sh TAG      -> everything lands in gpurun_out/TAG/, a digest in gpurun_out/TAG/summary.txt
TAG="${1:-visit}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
S="$OUT/summary.txt"; : > "$S"
echo "commit $(cat .gpurun_commit 2>/dev/null || echo unknown)" | tee -a "$S"
if [ -n "$TESTS" ]; then
  [ "$TESTS" = all ] && TESTS=tests
  echo "== pytest -m gpu: $TESTS" | tee -a "$S"
  timeout ${TEST_TIMEOUT:-900} python -m pytest $TESTS -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "rc=$?" | tee -a "$S"
  tail -4 "$OUT/pytest.log" | cut -c1-220 | tee -a "$S"
fi
if [ -n "${AB+x}" ] && [ -n "$AB" -o "${AB_DEFAULT_ONLY:-0}" = 1 ]; then
  echo "== knob A/B: $AB" | tee -a "$S"
  IFS='|' read -r -a variants <<< "$AB"
  REPS=${REPS:-2} STEPS=${STEPS:-40} bash tools/gpu_knob_ab.sh "$TAG/ab" "${variants[@]}" | tee -a "$S"
fi
if [ -n "$BENCH" ]; then
  IFS=';' read -r -a lines <<< "$BENCH"
  i=0
  for b in "${lines[@]}"; do
    i=$((i + 1))
    echo "== bench.py $b" | tee -a "$S"
    timeout 600 python bench.py $b > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"; echo "rc=$?" | tee -a "$S"
    cut -c1-1200 "$OUT/bench_$i.json" | tee -a "$S"
  done
fi
if [ "${TRACE:-0}" = 1 ]; then
  echo "== rocprofv3 kernel trace (eager iteration${TRACE_ARGS:+, $TRACE_ARGS})" | tee -a "$S"
  CMDT="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile --dropin 0 ${TRACE_ARGS:-}"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMDT > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$S"
  f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/eager_kernel_stats.csv"
  t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/steady_kernel_stats.csv" > "$OUT/steady_groups.txt" 2>&1
  head -${TRACE_HEAD:-60} "$OUT/steady_groups.txt" | cut -c1-150 | tee -a "$S"
  find "$OUT" -name "*kernel_trace*" -size +4M -delete
fi
for c in $CONVBENCH; do
  echo "== per-layer conv bench: $c" | tee -a "$S"
  timeout 300 python tools/conv_bench.py --config $c --batch ${CONVBENCH_BATCH:-32} ${CONVBENCH_ARGS:-} > "$OUT/conv_bench_$c.txt" 2>&1
  grep TOTAL "$OUT/conv_bench_$c.txt" | tee -a "$S"
done
if [ -n "$CMD" ]; then
  echo "== $CMD" | tee -a "$S"
  timeout ${CMD_TIMEOUT:-600} bash -c "$CMD" > "$OUT/cmd.log" 2>&1; echo "rc=$?" | tee -a "$S"
  tail -${CMD_TAIL:-40} "$OUT/cmd.log" | cut -c1-240 | tee -a "$S"
fi
