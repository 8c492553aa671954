#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc SQ pass (counter_collection csv) per kernel: counter sums per launch and ratios.
usage: sq_summarize.py <dir>"""
import csv, glob, os, sys, collections, re

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"\(.*", "", k)[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r.get("Dispatch_Id")
        if did not in cnt[k]:
            cnt[k].add(did)
            try:
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            except Exception:
                pass
names = sorted({c for v in acc.values() for c in v})
print("counters:", names)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]:
    n = len(cnt[k])
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    line = f"{k:58s} n={n:4d}"
    if dur[k]:
        line += f" {dur[k]/n/1e3:8.1f} us"
        # GRBM_GUI_ACTIVE is summed over the 8 XCD instances of the counter; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
        if "GRBM_GUI_ACTIVE" in v:
            line += f" clk {v['GRBM_GUI_ACTIVE']/8/dur[k]:.2f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            line += f" mfma_busy_per_simd {v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['GRBM_GUI_ACTIVE']/8*1024):.3f}"
    for c in names:
        if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES",):
            line += f" {c[3:]}={v.get(c,0)/wc:.3f}"
        elif not c.startswith("SQ_") and c != "GRBM_GUI_ACTIVE":
            line += f" {c}/launch={v.get(c,0)/max(n,1):.4g}"
    print(line)
