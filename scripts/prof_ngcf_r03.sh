#!/bin/bash
# GPU box: kernel stats (rocprofv3 --kernel-trace --stats) of the full-size NGCF step, batch-aware and whole-graph
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03_ngcf; rm -rf $out; mkdir -p $out
for cfg in ${CFGS:-32:0.5 4096:0.5 4096:0.0}; do
  set -- ${cfg/:/ }
  tag=b$1_f$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 scratch/ngcf_step.py $1 $2 100 ${ROUTE:-fused} > $out/$tag.log 2>&1 || { tail -5 $out/$tag.log; exit 1; }
  grep "ms per step" $out/$tag.log
  f=$(find $out/$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${tag}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"  sum of kernel time {tot/110/1e3:.1f} us per step (110 steps incl. warm-up)")
rows = [r for r in rows if "yr::" in r["Name"] or "rocclr" in r["Name"]]
print(f"  engine kernels + fills: {sum(float(r['TotalDurationNs']) for r in rows)/110/1e3:.1f} us per step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print(f"  {float(r['TotalDurationNs'])/110/1e3:8.1f} us/step  calls/step {int(r['Calls'])/110:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:90]}")
PY
  # the launches of ONE step (the last adam_dense_multi to the next), in order, with the idle gap before each
  t=$(find $out/$tag -name "*kernel_trace.csv" | head -1)
  python3 - "$t" > $out/${tag}_one_step_timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [k for k, r in enumerate(rows) if "adam_dense_multi" in r["Kernel_Name"]]
a, b = ends[-3] + 1, ends[-2] + 1
t0, prev = int(rows[a]["Start_Timestamp"]), None
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev is None else (s - prev) / 1e3
    busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:8.1f} us  +gap {gap:6.1f}  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'].split('(')[0].replace('void ', '')[:80]}")
    prev = e
print(f"step span {(prev - t0) / 1e3:.1f} us, busy {busy:.1f} us, {b - a} launches")
PY
  tail -1 $out/${tag}_one_step_timeline.txt
done
rm -rf $out/b*_f*/
