#!/bin/bash
# usage: scripts/pmc_eval.sh <tag>  (GPU box) — instruction-mix counters of the fused evaluation kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_eval_$1_$i -- python scratch/eval_prof.py > gpurun_out/pmc_eval_$1_$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_eval_$1_*/')):
    for f in glob.glob(d+'*/*counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'mf_eval_topk_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items(): print(k, round(sum(v)/len(v)), 'per launch (', len(v), 'launches )')
PY
