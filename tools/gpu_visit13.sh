#!/bin/bash
# visit 13: in-situ kernel traces of the eager iteration under the 64x64-tile rule and the previous rule
OUT=gpurun_out/r02v13; mkdir -p "$OUT"; export TMPDIR=/tmp
for v in new old; do
  [ $v = old ] && export MNK_BN128_KWORK=-1
  CMD="python $PWD/bench.py --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-profile"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof_$v" -o bench -- $CMD > "$OLDPWD/$OUT/rocprof_$v.log" 2>&1 ); echo "rocprof $v rc=$?"
  t=$(find "$OUT/prof_$v" -name "*kernel_trace.csv" | head -1)
  python tools/trace_groups.py "$t" --csv "$OUT/steady_$v.csv" > "$OUT/steady_groups_$v.txt" 2>&1
  python - "$t" "$OUT/trace_$v.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    for r in rows:
        f.write("%s,%s,%s,%s,%s,%d\n" % (r["Kernel_Name"].replace(",", ";")[:90], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"],
                                     r["Start_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
P
  rm -rf "$OUT/prof_$v"
  head -3 "$OUT/steady_groups_$v.txt"
done
