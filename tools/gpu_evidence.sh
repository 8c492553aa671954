#!/bin/bash
# One GPU-box visit that collects everything profiles/ cites for the final build of a round.  Usage: gpu_evidence.sh TAG
#   tests + smoke, the bench lines (default = BASELINE configs[1] with roofline / cpu_baseline / hot_path_only), rocprofv3
#   kernel stats (+ steady-state window), FETCH_SIZE / WRITE_SIZE PMC passes (separate runs), an SQ pass over the per-layer
#   conv bench (MFMA busy, GRBM clock), per-layer conv bench of both configs, batched inference (bair B=512), the
#   single-rank RCCL exercise of the distributed path.
TAG="${1:-r04final}"; R="${TAG%%final*}"; R="${R:-r04}"
OUT=gpurun_out/$TAG; mkdir -p "$OUT" profiles; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
# (profiled EAGER iterations launch their weight-gradient GEMMs at the end, as a captured iteration does: with the background
# launches of an eager backward, MNK_WGRAD_BG, kernels overlap and their durations are not comparable)
EAGER_ENV="MNK_WGRAD_BG=0"
S="$OUT/summary.txt"; : > "$S"
COMMIT="$(cat .gpurun_commit 2>/dev/null || echo unknown)"
echo "== pytest -m gpu + smoke" | tee -a "$S"
timeout 900 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?" | tee -a "$S"; tail -3 "$OUT/pytest_gpu.log" | cut -c1-200 | tee -a "$S"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200 | tee -a "$S"
echo "== pmc FETCH_SIZE / WRITE_SIZE (separate passes)" | tee -a "$S"
CMD2="env $EAGER_ENV python $PWD/bench.py --steps 2 --warmup 2 --graph 0 --no-cpu-baseline --no-profile --dropin 0"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_fetch" -o f -- $CMD2 > "$OLDPWD/$OUT/pmc_fetch.log" 2>&1 ); echo "fetch rc=$?" | tee -a "$S"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_write" -o w -- $CMD2 > "$OLDPWD/$OUT/pmc_write.log" 2>&1 ); echo "write rc=$?" | tee -a "$S"
python tools/pmc_summarize.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic_moving-gif_b32.json" 2>&1 | tee -a "$S"
python - "$OUT/pmc_traffic_moving-gif_b32.json" "$COMMIT" <<'P'
import json, sys
sys.path.insert(0, ".")
try:
    import bench
    d = json.load(open(sys.argv[1])); d["_measured_on"] = "commit " + sys.argv[2]
    d["_kernel_source_stamp"] = bench.kernel_source_stamp()      # bench.py quotes the file only on these kernel sources
    d["_all_source_stamp"] = bench.all_source_stamp()            # ... and the non-GEMM kernels' traffic (roofline_hbm) only on these
    # iterations the passes saw: the soft-argmax forward runs once per training iteration
    d["_iterations"] = max([v["launches"] for k, v in d.items() if isinstance(v, dict) and k.startswith("softmax_kp_fwd")] or [0])
    json.dump(d, open(sys.argv[1], "w"), indent=1)
except Exception as e:
    print("pmc stamp:", e); sys.exit(1)
P
[ $? -eq 0 ] || { echo "EVIDENCE FAILED: the PMC traffic file could not be stamped with the kernel sources" | tee -a "$S"; }
find "$OUT" -name "*counter_collection*" -size +8M -delete; find "$OUT" -name "*kernel_trace*" -size +4M -delete
# bench.py reads roofline.traffic from profiles/: the file of THIS build
cp "$OUT/pmc_traffic_moving-gif_b32.json" "profiles/${R}_pmc_traffic_moving-gif_b32.json" 2>/dev/null
echo "== bench (default: moving-gif, roofline + cpu_baseline + hot_path_only)" | tee -a "$S"
timeout 900 python bench.py > "$OUT/bench_moving-gif_b32.json" 2> "$OUT/bench.err"; echo "rc=$?" | tee -a "$S"; cut -c1-900 "$OUT/bench_moving-gif_b32.json" | tee -a "$S"
timeout 400 python bench.py --config taichi --no-cpu-baseline > "$OUT/bench_taichi_b32.json" 2> "$OUT/bench_taichi.err"; echo "taichi rc=$?" | tee -a "$S"; cut -c1-300 "$OUT/bench_taichi_b32.json" | tee -a "$S"
timeout 300 python bench.py --config vox --size 256 --batch 8 --no-cpu-baseline > "$OUT/bench_vox256_b8.json" 2> "$OUT/bench_vox.err"; echo "vox 256 rc=$?" | tee -a "$S"; cut -c1-300 "$OUT/bench_vox256_b8.json" | tee -a "$S"
echo "== rocprofv3 kernel stats (eager iteration)" | tee -a "$S"
CMD="env $EAGER_ENV python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile --dropin 0"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$S"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/moving-gif_b32_eager_kernel_stats.csv"
t=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_groups.py "$t" --csv "$OUT/moving-gif_b32_steady_kernel_stats.csv" > "$OUT/moving-gif_b32_steady_groups.txt" 2>&1
head -45 "$OUT/moving-gif_b32_steady_groups.txt" | cut -c1-130 | tee -a "$S"
find "$OUT" -name "*kernel_trace*" -size +4M -delete
echo "== rocprofv3 kernel trace of hipGraph replays: idle time between kernels (tools/trace_gaps.py)" | tee -a "$S"
CMDG="python $PWD/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --dropin 0"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_graph -o g -- $CMDG > "$OLDPWD/$OUT/rocprof_graph.log" 2>&1 ); echo "rocprof rc=$?" | tee -a "$S"
t=$(find /tmp/prof_graph -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_gaps.py "$t" --last 6 > "$OUT/graph_replay_gaps.txt" 2>&1; head -3 "$OUT/graph_replay_gaps.txt" | cut -c1-200 | tee -a "$S"
echo "== SQ pass over the per-layer conv bench (moving-gif): MFMA busy, clock" | tee -a "$S"
CMD3="python $PWD/tools/conv_bench.py --config moving-gif --batch 32 --iters 3"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/$OUT/sq" -o s -- $CMD3 > "$OLDPWD/$OUT/sq.log" 2>&1 ); echo "sq rc=$?" | tee -a "$S"
python tools/sq_summarize.py "$OUT/sq" > "$OUT/sq_counters_conv_bench_moving-gif.txt" 2>&1; head -14 "$OUT/sq_counters_conv_bench_moving-gif.txt" | cut -c1-260 | tee -a "$S"
find "$OUT" -name "*counter_collection*" -size +8M -delete; find "$OUT" -name "*kernel_trace*" -size +4M -delete
echo "== per-layer conv bench" | tee -a "$S"
for c in moving-gif taichi; do timeout 300 python tools/conv_bench.py --config $c --batch 32 > "$OUT/conv_bench_${c}_b32.txt" 2>&1; grep TOTAL "$OUT/conv_bench_${c}_b32.txt" | tee -a "$S"; done
timeout 300 python tools/conv_bench.py --config vox --batch 8 --size 256 > "$OUT/conv_bench_vox256_b8.txt" 2>&1; grep TOTAL "$OUT/conv_bench_vox256_b8.txt" | tee -a "$S"
echo "== fp32 MFMA pipe micro-benchmark (what the peak of the roofline is worth on this chip)" | tee -a "$S"
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/microbench/mfma_f32_peak.hip 2>/dev/null && timeout 120 /tmp/mfma_peak > "$OUT/mfma_f32_peak.txt" 2>&1; head -8 "$OUT/mfma_f32_peak.txt" | tee -a "$S"
# the parity evidence pytest wrote (tests/test_fullsize_oracle.py, tests/test_kp_index.py)
mkdir -p "$OUT/parity"; cp gpurun_out/parity_*.json gpurun_out/kp_index_*.json "$OUT/parity/" 2>/dev/null
python - "$OUT" <<'P'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "parity", "kp_index_*.json"))):
    d = json.load(open(f))
    rows.append("%-34s pixel %5d / %-5d argmax %5d / %-5d max |mean - ref64| %.2e  nearest cell boundary %.2e px  listed %d" % (
        d["case"], d["pixel_equal"], d["pixel_total"], d["argmax_equal"], d["argmax_total"], d["max_abs_mean_error"],
        d["smallest_boundary_distance_px"], len(d["listed_ill_defined"])))
for f in sorted(glob.glob(os.path.join(out, "parity", "parity_*.json"))):
    d = json.load(open(f))
    rows.append("%-34s %d quantities, error / reference's own fp32 error: %s, worst error / tolerance %.2f" % (
        os.path.basename(f)[7:-5], d["n"], d.get("error_over_reference_fp32_error"), d["worst"][0]["ratio"]))
open(os.path.join(out, "parity_summary.txt"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows[-4:]))
P
echo "== batched inference (bair, B=512)" | tee -a "$S"
timeout 300 python tools/infer_bench.py > "$OUT/infer_bair_b512.json" 2> "$OUT/infer.err"; cat "$OUT/infer_bair_b512.json" | cut -c1-400 | tee -a "$S"
echo "== single-rank RCCL exercise (MNK_DIST_FORCE=1, torch.distributed.run)" | tee -a "$S"
MNK_DIST_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > "$OUT/bench_dist1.json" 2> "$OUT/bench_dist1.err"; echo "rc=$?" | tee -a "$S"
cut -c1-330 "$OUT/bench_dist1.json" | tee -a "$S"; grep -h "mnk.dist\|capture failed" "$OUT/bench_dist1.err" | head -3 | cut -c1-200 | tee -a "$S"
echo "== the same with the SyncBN sums through RCCL instead of the peer-to-peer exchange (MNK_SYNCBN_P2P=0)" | tee -a "$S"
MNK_SYNCBN_P2P=0 MNK_DIST_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > "$OUT/bench_dist1_rccl.json" 2> "$OUT/bench_dist1_rccl.err"; echo "rc=$?" | tee -a "$S"
cut -c1-200 "$OUT/bench_dist1_rccl.json" | tee -a "$S"
echo "== the same with the overlapped gradient exchange of several ranks forced on (two linear hipGraphs + host calls)" | tee -a "$S"
MNK_DIST_FORCE=1 MNK_GRAD_OVERLAP=force timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > "$OUT/bench_dist1_overlap.json" 2> "$OUT/bench_dist1_overlap.err"; echo "rc=$?" | tee -a "$S"
cut -c1-200 "$OUT/bench_dist1_overlap.json" | tee -a "$S"
# what profiles/ keeps (small files only)
for f in bench_moving-gif_b32.json bench_taichi_b32.json bench_vox256_b8.json moving-gif_b32_eager_kernel_stats.csv moving-gif_b32_steady_kernel_stats.csv moving-gif_b32_steady_groups.txt pmc_traffic_moving-gif_b32.json sq_counters_conv_bench_moving-gif.txt conv_bench_moving-gif_b32.txt conv_bench_taichi_b32.txt conv_bench_vox256_b8.txt infer_bair_b512.json mfma_f32_peak.txt parity_summary.txt; do
  [ -s "$OUT/$f" ] && cp "$OUT/$f" "$OUT/${R}_$f"
done
cp "$OUT/bench_dist1.json" "$OUT/${R}_bench_moving-gif_b32_rccl_1rank.json" 2>/dev/null
cp "$OUT/bench_dist1_overlap.json" "$OUT/${R}_bench_moving-gif_b32_rccl_1rank_overlap.json" 2>/dev/null
cp "$OUT/bench_dist1_rccl.json" "$OUT/${R}_bench_moving-gif_b32_rccl_1rank_syncbn_rccl.json" 2>/dev/null
cp gpurun_out/p2p_timing_world2.txt gpurun_out/p2p_timing_world4.txt "$OUT/" 2>/dev/null
cp "$OUT/graph_replay_gaps.txt" "$OUT/${R}_graph_replay_gaps.txt" 2>/dev/null
tail -3 "$OUT/pytest_gpu.log" > "$OUT/${R}_pytest_gpu_tail.txt"
