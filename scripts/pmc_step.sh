#!/bin/bash
# usage: scripts/pmc_step.sh <tag>  (GPU box) — separate PMC passes for HBM traffic and L2 hit rate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$1_$c -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$1_$c.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_$1_L2 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$1_L2.log 2>&1
echo done
