#!/bin/bash
# usage: scripts/prof_cmd.sh <tag> <python script and args...>   (on the GPU box through gpurun)
# rocprofv3 kernel trace + stats of the command; prints the yr:: kernels (calls, average us) and their sum.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
tail -2 gpurun_out/prof_$tag.log
python3 - <<PY
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_$tag/*/*_kernel_stats.csv'))[-1]
tot=0; allt=0
for r in csv.reader(open(f)):
    if r[0]=='Name': continue
    allt+=float(r[2])/1e3
    if 'yr::' in r[0]:
        print(r[0].replace('void ','')[:70].ljust(70), r[1].rjust(6), str(round(float(r[3])/1e3,1)).rjust(9)); tot+=float(r[2])/1e3
print('yr_total_us', round(tot,1), 'all_kernels_us', round(allt,1), 'yr_share', round(tot/allt,3))
PY
