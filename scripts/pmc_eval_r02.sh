#!/bin/bash
# usage: scripts/pmc_eval_r02.sh  (GPU box) — instruction-mix counters of the fused evaluation, one rocprofv3 --pmc pass per
# counter group, for the f32 instruction, the bf16-split form (prescan thresholds) and the bf16-split form with hint lists;
# per-launch averages per kernel (the sweep and the prescan are instantiations of one template: told apart by name)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_eval_r02; rm -rf $out; mkdir -p $out
for form in "f32" "bf16x3" "bf16x3 hint"; do
  tag=$(echo $form | tr ' ' '_'); i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_$i -- python3 scratch/eval_prof.py $form > $out/${tag}_$i.log 2>&1 || echo "pass $i of $form failed"
  done
done
python3 - <<PY
import csv, glob, collections, re
for tag in ("f32", "bf16x3", "bf16x3_hint"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"$out/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"mf_eval_topk_kernel<([^>]*)>", r["Kernel_Name"])
            if m:
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern, cs in sorted(agg.items()):
        what = "prescan" if kern.endswith("true") else "sweep"
        print(f"# {tag}: mf_eval_topk_kernel<{kern}> ({what}; <D, list length, bias, bf16 split, prescan>)")
        for k, v in sorted(cs.items()):
            v = v[1:] if len(v) > 1 and tag.endswith("hint") and what == "sweep" else v      # hinted run: the first call has no hint
            print(f"{k} {round(sum(v) / len(v))} per launch ({len(v)} launches)")
PY
