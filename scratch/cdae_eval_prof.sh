#!/bin/bash
# usage (GPU box): scratch/cdae_eval_prof.sh [rows per evaluation batch] — rocprofv3 kernel stats of CDAE validate + evaluate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/cdae_eval; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 scratch/cdae_valid_epoch.py 256 lists ${1:-256} > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
grep -E "validate|evaluate" $out/log.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -d'(' -f1,2 | cut -c1-90 | paste - <(head -16 "$f" | awk -F'",' '{print $2}' | cut -d, -f1-4)
