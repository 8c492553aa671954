"""Idle time between consecutive kernels of a rocprofv3 kernel trace, per training iteration (hipGraph replay or eager):
where the GPU has nothing to run.  Usage: trace_gaps.py trace.csv [--marker KERNEL_SUBSTR] [--skip N] [--top K]
Prints the iteration period, the summed kernel time, the summed idle time, the idle time at the iteration boundary (between
the last kernel of one iteration and the first of the next) and the largest gaps with the kernels on both sides."""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="softmax_kp_fwd_kernel")
ap.add_argument("--skip", type=int, default=2)
ap.add_argument("--top", type=int, default=15)
ap.add_argument("--last", type=int, default=0, help="only the last N iterations (the hipGraph replays of a bench.py run)")
a = ap.parse_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:60]


marks = [i for i, (s, e, n) in enumerate(rows) if a.marker in n]
marks = marks[-(a.last + 1):] if a.last else marks[a.skip:]
iters = len(marks) - 1
t0, t1 = rows[marks[0]][0], rows[marks[-1]][0]
win = rows[marks[0]:marks[-1]]
busy_end = win[0][1]
idle = overlap = 0
gaps = collections.defaultdict(list)
for (s, e, n), (s2, e2, n2) in zip(win, win[1:] + [rows[marks[-1]]]):
    busy_end = max(busy_end, e)
    g = s2 - busy_end
    if g > 0:
        idle += g
        gaps[(short(n), short(n2))].append(g)
    else:
        overlap += -g
ktime = sum(e - s for s, e, n in win)
print(f"{iters} iterations: period {(t1 - t0) / 1e6 / iters:.3f} ms, kernel time {ktime / 1e6 / iters:.3f} ms, idle {idle / 1e6 / iters:.3f} ms, "
      f"{len(win) // iters} launches, mean gap {idle / max(1, len(win)) / 1e3:.2f} us")
hist = collections.Counter()
for v in gaps.values():
    for g in v:
        hist[min(int(g / 1000), 20)] += 1
print("gap histogram (us -> launches per iteration):", {k: round(v / iters, 1) for k, v in sorted(hist.items())})
print("largest gaps per iteration (us total, count, mean):")
for (n, n2), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:a.top]:
    print(f"{sum(v) / 1e3 / iters:8.1f} us {len(v) / iters:6.1f}x {sum(v) / len(v) / 1e3:8.1f} us   {n}  ->  {n2}")
