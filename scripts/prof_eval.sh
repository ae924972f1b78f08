#!/bin/bash
# usage: scripts/prof_eval.sh <tag>  (GPU box): rocprofv3 kernel stats of an end-to-end MFTrainer epoch loop at Yelp2018 size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epoch_$1 -- python scratch/epoch_time.py 65536 > gpurun_out/epoch_$1.log 2>&1 || exit 1
grep -v rocprofv3 gpurun_out/epoch_$1.log | tail -14
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_epoch_$1/*/*_kernel_stats.csv')[0]
for r in list(csv.reader(open(f)))[1:]:
    if 'yr::' in r[0]: print(r[0].replace('void ','')[:70].ljust(70), r[1].rjust(6), str(round(float(r[3])/1e3,1)).rjust(9))
PY
