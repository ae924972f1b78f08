#!/bin/bash
# usage: scripts/pmc_eval_r03.sh  (GPU box) — matrix-pipe counters of the fused evaluation sweep, both forms, with hint lists,
# D = 64 and D = 128 (one rocprofv3 --pmc pass per counter group; per-launch averages of the sweep kernel).
# pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs per CU ... see the printed formula)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_eval_r03; rm -rf $out; mkdir -p $out
for cfg in "four_waves 64" "two_roles 64" "four_waves 128" "two_roles 128"; do
  tag=$(echo $cfg | tr ' ' '_'); i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_$i -- python3 scratch/eval_prof.py bf16x3 hint $cfg > $out/${tag}_$i.log 2>&1 || echo "pass $i of $cfg failed"
  done
done
python3 - <<PY  # (the last line it prints per kernel uses GRBM_GUI_ACTIVE, which is summed over XCDs: see profiles/r03_eval_pmc_both_forms.txt)
import csv, glob, collections, re
for tag in ("four_waves_64", "two_roles_64", "four_waves_128", "two_roles_128"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"$out/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(mf_eval_topk(?:_pp)?_kernel<[^>]*>)", r["Kernel_Name"])
            if m and not m.group(1).endswith("true>"):
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern, cs in sorted(agg.items()):
        print(f"# {tag}: {kern}")
        vals = {}
        for k, v in sorted(cs.items()):
            v = v[1:] if len(v) > 1 else v                  # the first call has no hint
            vals[k] = sum(v) / len(v)
            print(f"{k} {round(vals[k])} per launch ({len(v)} launches)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
            # MFMA_BUSY sums over the chip's 1,024 SIMDs, GUI_ACTIVE is the launch's duration in cycles
            print(f"matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE) = {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / vals['GRBM_GUI_ACTIVE']:.3f}")
PY
