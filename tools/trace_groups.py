"""Group a rocprofv3 kernel trace by (kernel, grid size): calls, mean/total duration per training iteration.

Usage: trace_groups.py trace.csv [--marker KERNEL_SUBSTR] [--skip N]
The iterations are periodic, so the window between the end of the (N)th and of the last launch of a once-per-iteration
kernel (default: softmax_kp_fwd_kernel, the key-point read-out of the one detector forward) holds whole iterations only -- this drops MIOpen's find-mode
kernels (naive_conv_*, Im2d2Col, Cijk_*) that run during warm-up.  Also writes a kernel_stats-style csv of the window
when --csv PATH is given.
"""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="softmax_kp_fwd_kernel")
ap.add_argument("--skip", type=int, default=2)
ap.add_argument("--csv", default=None)
a = ap.parse_args()

rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name,
                     (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))))
rows.sort()
marks = [e for (s, e, n, g) in rows if a.marker in n]
iters = 1
if len(marks) > a.skip:
    t0, t1 = marks[a.skip - 1] if a.skip > 0 else rows[0][0], marks[-1]
    iters = len(marks) - a.skip
    rows = [r for r in rows if t0 < r[0] and r[1] <= t1]
    print(f"window: {iters} iterations, {(t1 - t0) / 1e6 / iters:.3f} ms wall per iteration (eager, profiler attached)")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:72]


g = collections.defaultdict(list)
k = collections.defaultdict(list)
for s, e, n, grid in rows:
    g[(short(n), grid)].append(e - s)
    k[short(n)].append(e - s)
tot = sum(sum(v) for v in k.values())
print(f"kernel time {tot / 1e6 / iters:.3f} ms per iteration, {sum(len(v) for v in k.values()) // iters} launches per iteration")
print("\n== by kernel (per iteration)")
for n, v in sorted(k.items(), key=lambda kv: -sum(kv[1]))[:60]:
    print(f"{sum(v) / 1e3 / iters:9.1f} us  {100 * sum(v) / tot:5.1f}%  {len(v) // iters:4d} calls  {sum(v) / len(v) / 1e3:8.1f} us avg  {n}")
print("\n== by kernel and grid (per iteration)")
for (n, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:100]:
    print(f"{sum(v) / 1e3 / iters:9.1f} us  {len(v) / iters:5.1f} calls  {sum(v) / len(v) / 1e3:8.1f} us avg  grid={'x'.join(x for x in grid if x)}  {n}")
if a.csv:
    with open(a.csv, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerIteration", "TotalNsPerIteration", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(k.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v) / iters, sum(v) / iters, sum(v) / len(v), round(100 * sum(v) / tot, 3), min(v), max(v)])
