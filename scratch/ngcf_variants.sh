#!/bin/bash
# on the GPU box: rebuild ngcf.hip with each set of flags; kernel times of the full-size NGCF step under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" ngcf.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ngcfv -- python3 bench.py --workload ngcf --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'], 'ms per step')"
   python3 - <<PY
import csv,glob,os
f=sorted(glob.glob('gpurun_out/prof_ngcfv/*/*_kernel_stats.csv'), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if 'ngcf_dense' in r['Name']: print('  ', r['Name'].split('(')[0].replace('void ',''), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
  ) || exit 1
done
