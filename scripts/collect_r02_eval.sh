#!/bin/bash
# Round-2 evaluation evidence (GPU box): bench lines of the other workloads, rocprofv3 kernel stats of
# `bench.py --workload eval` (both precisions), every precision / k / prescan combination (scratch/eval_split.py), the
# phase timeline of the sweep (-DYR_ET_STAMPS) with and without the prescan, CDAE validate / evaluate over all users.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r02e; rm -rf $out; mkdir -p $out
for w in eval ngcf cdae; do python3 bench.py --workload $w >> $out/bench_other_workloads.jsonl 2>> $out/bench_other.err || exit 1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_eval -- python3 bench.py --workload eval > $out/prof_eval.log 2>&1 || exit 1
python3 scratch/eval_split.py > $out/eval_split.txt 2>&1 || exit 1
python3 scratch/cdae_valid_epoch.py 256 lists > $out/cdae_valid_lists.txt 2>&1 || exit 1
for k in 10 16; do for ps in 0 1; do
  echo "k=$k prescan=$ps" >> $out/eval_phases.txt
  YR_K=$k YR_PRESCAN=$ps scratch/eval_phases.sh >> $out/eval_phases.txt 2>&1 || exit 1
done; done
for k in 10 16; do
  echo "k=$k hint lists (the own result)" >> $out/eval_phases.txt
  YR_K=$k YR_HINT=1 scratch/eval_phases.sh >> $out/eval_phases.txt 2>&1 || exit 1
done
echo collected
