#!/bin/bash
# GPU box: kernel stats (rocprofv3 --kernel-trace --stats) of the full-size NGCF step, batch-aware and whole-graph
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03_ngcf; rm -rf $out; mkdir -p $out
for cfg in "32 0.5" "4096 0.5" "4096 0.0"; do
  set -- $cfg
  tag=b$1_f$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 scratch/ngcf_step.py $1 $2 100 > $out/$tag.log 2>&1 || { tail -5 $out/$tag.log; exit 1; }
  grep "ms per step" $out/$tag.log
  f=$(find $out/$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/${tag}_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"  sum of kernel time {tot/110/1e3:.1f} us per step (110 steps incl. warm-up)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"  {float(r['TotalDurationNs'])/110/1e3:8.1f} us/step  calls/step {int(r['Calls'])/110:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:90]}")
PY
done
rm -rf $out/b*_f*/
