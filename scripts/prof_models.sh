#!/bin/bash
# usage: scripts/prof_models.sh <tag> <model> <batch>  (on the GPU box): rocprofv3 kernel stats of full-size NGCF / CDAE steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; m=$2; b=$3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${m}_$tag -- python scratch/step_models.py $m $b 20 > gpurun_out/step_${m}_$tag.log 2>&1 || exit 1
tail -1 gpurun_out/step_${m}_$tag.log
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_${m}_$tag/*/*_kernel_stats.csv')[0]
rows=list(csv.reader(open(f)))[1:]
tot=sum(float(r[2]) for r in rows)
for r in rows[:24]:
    print(r[0].replace('void ','')[:70].ljust(70), r[1].rjust(6), str(round(float(r[3])/1e3,1)).rjust(8), str(round(100*float(r[2])/tot,1)).rjust(6))
PY
