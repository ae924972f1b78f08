#!/bin/bash
# Round-2 CDAE evidence (GPU box): bench lines of the other workloads, rocprofv3 kernel stats of the CDAE training
# step in its three forms and of a whole train epoch fed with list batches, PMC traffic of the sampled-decoder step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r02c; mkdir -p $out
for w in eval ngcf cdae; do python3 bench.py --workload $w >> $out/bench_other_workloads.jsonl 2>> $out/bench_other.err || exit 1; done
for f in sampled dense auto; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cdae_$f -- python3 scratch/cdae_step_prof.py $f > $out/prof_cdae_$f.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cdae_epoch -- python3 scratch/cdae_epoch.py 256 lists > $out/prof_cdae_epoch.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_cdae_$c -- python3 scratch/cdae_step_prof.py sampled > $out/pmc_cdae_$c.log 2>&1 || exit 1
done
python3 scratch/cdae_epoch.py 256 dense > $out/cdae_epoch_dense.txt 2>&1
python3 scratch/cdae_epoch.py 256 lists > $out/cdae_epoch_lists.txt 2>&1
echo collected
