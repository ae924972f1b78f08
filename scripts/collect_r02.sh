#!/bin/bash
# Round-2 evidence (GPU box): rocprofv3 kernel stats of the default bench command and of the step at the swept
# batch sizes, PMC traffic of the default bench command, bench lines of the other full-size workloads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r02; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --no-cpu-baseline --no-sweep > $out/prof_bench.log 2>&1 || exit 1
for b in 32 4096 65536 262144; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_b$b -- python3 scratch/step_prof.py $b auto synth 50 > $out/prof_b$b.log 2>&1 || exit 1
done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_bench_$n -- python3 bench.py --no-cpu-baseline --no-sweep --steps 20 > $out/pmc_bench_$n.log 2>&1 || exit 1
done
for w in eval ngcf cdae; do python3 bench.py --workload $w >> $out/bench_other_workloads.jsonl 2>> $out/bench_other.err || exit 1; done
echo collected
