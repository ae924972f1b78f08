#!/bin/bash
# Round-3 evidence (GPU box): the full GPU test suite, the headline-size parity report, bench lines (default, the
# other workloads, a 2-rank one-GPU rehearsal of the N > 1 line), rocprofv3 kernel stats of the default bench command
# and of the step at the swept batch sizes, PMC traffic of the default bench command, NGCF step kernel stats.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03; rm -rf $out; mkdir -p $out
YR_PARITY_REPORT=$PWD/$out/mf_full_parity.json python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 || { tail -20 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
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
YR_BENCH_REHEARSAL_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_rehearsal_2ranks_one_gpu.json 2> $out/bench_rehearsal.err || exit 1
for s in 1048576:skew 1048576:uniform 65536:skew; do
  python3 scratch/step_prof.py ${s%:*} pull ${s#*:} 100 >> $out/skewed_batches.txt 2>/dev/null || exit 1
done
python3 scratch/emul_shard.py > $out/emulated_rank_step.txt 2>/dev/null || exit 1
scripts/prof_ngcf_r03.sh > $out/ngcf_prof.txt 2>&1 || exit 1
scripts/prof_cdae_valid_r03.sh 16 > $out/cdae_valid_prof.txt 2>&1 || exit 1
echo collected
