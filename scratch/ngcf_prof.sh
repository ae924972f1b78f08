#!/bin/bash
# usage (GPU box): scratch/ngcf_prof.sh — rocprofv3 kernel stats of `bench.py --workload ngcf` (yr:: kernels, per-call µs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/ngcf_prof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --workload ngcf > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
grep "yr::" "$f" | sed 's/void yr:://;s/^"//' | awk -F'",' '{split($2,a,","); n=split($1,b,"("); printf "%-48s calls %5d  avg %8.1f us  %5.1f %%\n", substr(b[1],1,48), a[1], a[3]/1000, a[4]}'
