"""The kernels of one replayed training iteration in launch order (rocprofv3 kernel trace of a hipGraph-replay bench run):
offset from the iteration's first kernel, duration, idle time in front of it, name and grid -- averaged over the last N
iterations, which replay the same sequence.  Usage: trace_timeline.py trace.csv [--last 6] [--marker softmax_kp_fwd_kernel]
Ends with the totals per kernel name (time, launches, idle time in front)."""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="softmax_kp_fwd_kernel")
ap.add_argument("--last", type=int, default=6)
a = ap.parse_args()
rows = []
with open(a.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"),
                     "%sx%sx%s" % (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:64]


marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
marks = marks[-(a.last + 1):]
its = [rows[marks[k]:marks[k + 1]] for k in range(len(marks) - 1)]
n = len(its[0])
same = [it for it in its if len(it) == n and all(x[2] == y[2] for x, y in zip(it, its[0]))]
print("%d iterations of %d launches (of %d windows), period %.3f ms" % (
    len(same), n, len(its), (rows[marks[-1]][0] - rows[marks[0]][0]) / 1e6 / (len(marks) - 1)))
tot = collections.defaultdict(lambda: [0.0, 0, 0.0])
for j in range(n):
    off = sum(it[j][0] - it[0][0] for it in same) / len(same) / 1e3
    dur = sum(it[j][1] - it[j][0] for it in same) / len(same) / 1e3
    gap = sum(it[j][0] - it[j - 1][1] for it in same) / len(same) / 1e3 if j else 0.0
    name = short(same[0][j][2])
    print("%9.1f us  %7.1f us  gap %6.1f  %-64s %s" % (off, dur, gap, name, same[0][j][3]))
    t = tot[name]
    t[0] += dur
    t[1] += 1
    t[2] += gap
print("\n== per kernel: time, launches, idle time in front of its launches")
for name, (d, c, g) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%9.1f us %4d x  idle %7.1f us  %s" % (d, c, g, name))
print("total kernel %.1f us, idle %.1f us" % (sum(v[0] for v in tot.values()), sum(v[2] for v in tot.values())))
