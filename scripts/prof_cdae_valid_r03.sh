#!/bin/bash
# GPU box: kernel stats of CDAE validate() + evaluate() over list batches at Yelp2018 size (group = $1 batches per launch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03_cdae_valid; rm -rf $out; mkdir -p $out
YR_GROUP=${1:-16} rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 scratch/cdae_valid_epoch.py 256 lists > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
grep -E "validate|evaluate" $out/log.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "yr::" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"  total {float(r['TotalDurationNs'])/1e3/2:9.1f} us per pass-pair  calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
rm -rf $out/prof
