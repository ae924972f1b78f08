#!/bin/bash
# usage: scripts/prof_step.sh <tag>   (run on the GPU box through gpurun; writes gpurun_out/prof_<tag>)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$1.log 2>&1
grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*' gpurun_out/bench_$1.log | tr '\n' ' '; echo
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_$1/*/*_kernel_stats.csv')[0]
tot=0
for r in csv.reader(open(f)):
    if 'yr::' in r[0]:
        print(r[0].replace('void ','')[:60].ljust(60), r[1], round(float(r[3])/1e3,1)); tot+=float(r[3])/1e3
print('sum_us', round(tot,1))
PY
