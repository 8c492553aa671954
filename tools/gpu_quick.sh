#!/bin/bash
# quick GPU visit: tests, conv bench (optional), moving-gif + taichi bench
TAG="${1:-q}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
: > "$OUT/summary.txt"
if [ "${CB:-1}" = "1" ]; then
timeout 300 python tools/conv_bench.py --config taichi --batch 32 > "$OUT/conv_bench_taichi.txt" 2>&1
grep TOTAL "$OUT/conv_bench_taichi.txt" | tee -a "$OUT/summary.txt"
fi
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"; grep -E "^E  |FAILED" "$OUT/pytest_gpu.log" | head -10 | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
timeout 300 python tools/infer_bench.py > "$OUT/infer_bench.json" 2> "$OUT/infer_bench.err"; cat "$OUT/infer_bench.json" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/infer_bench.err" | cut -c1-200 | tee -a "$OUT/summary.txt"
timeout 400 python bench.py --config taichi --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_taichi.json" 2> "$OUT/bench_taichi.err"
python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for f in ("bench.json", "bench_taichi.json"):
    try:
        d = json.load(open(sys.argv[1] + "/" + f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], "frames/s", d["ms_per_step"], "ms", d["config"]["launch"], "roofline", d["roofline"] and (d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"]), "cpu", d.get("cpu_baseline") and round(d["cpu_baseline"]["value"], 2))
    print("   ", {k: round(v["ms_per_step"], 2) for k, v in d["kernels"].items()}, "sum", round(sum(v["ms_per_step"] for v in d["kernels"].values()), 2))
PY
tail -3 "$OUT/bench.err" | cut -c1-300 | tee -a "$OUT/summary.txt"
