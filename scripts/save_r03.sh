#!/bin/bash
# copy the round-3 evidence collected by scripts/collect_r03.sh (gpurun_out/r03) into profiles/
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r03
impl=$(python3 -c "
import json;d=json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1]);print(d['config']['step_impl'])")
python3 scripts/save_profile.py $o/prof_bench r03_bench_default_b524288_kernel_stats $o/prof_bench.log
for b in 32 4096 65536 262144; do python3 scripts/save_profile.py $o/prof_b$b r03_step_b${b}_kernel_stats; done
python3 scripts/pmc_traffic.py $o/pmc_bench_FETCH_SIZE $o/pmc_bench_WRITE_SIZE 524288 "$impl" r03_bench_default_pmc_traffic | head -1
cp $o/bench_default.json profiles/r03_bench_default_run.json
cp $o/bench_other_workloads.jsonl profiles/r03_bench_other_workloads.jsonl
cp $o/bench_rehearsal_2ranks_one_gpu.json profiles/r03_bench_rehearsal_2ranks_one_gpu.json
cp $o/mf_full_parity.json profiles/r03_mf_full_parity_vs_reference_run.json
cp $o/skewed_batches.txt profiles/r03_skewed_batches.txt
cp $o/emulated_rank_step.txt profiles/r03_emulated_rank_step.txt
tail -3 $o/pytest_gpu.log > profiles/r03_pytest_gpu_summary.txt
for t in b32_f0.5 b4096_f0.5 b4096_f0.0; do
  n=${t/_f0.5/}; n=${n/_f0.0/_whole_graph}
  cp gpurun_out/r03_ngcf/${t}_kernel_stats.csv profiles/r03_ngcf_step_${n}_kernel_stats.csv
  cp gpurun_out/r03_ngcf/${t}_one_step_timeline.txt profiles/r03_ngcf_step_${n}_one_step_timeline.txt
done
cp gpurun_out/r03_cdae_valid/kernel_stats.csv profiles/r03_cdae_validate_evaluate_kernel_stats.csv
grep -ho '"frac[a-z_]*": [0-9.]*' profiles/r03_*.json profiles/r03_*.jsonl | awk -F': ' '$2>1 {print "FRACTION ABOVE 1:", $0}'
echo saved
