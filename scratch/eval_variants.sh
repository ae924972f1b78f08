#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (GPU box): scratch/eval_variants.sh "<flags 1>" "<flags 2>" ...: rebuild csrc/eval_topk.hip with each flag set,
# time the full-size evaluation (bench.py --workload eval)
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && python3 bench.py --workload eval 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'ms cold', d['roofline']['with_hint_lists']['ms'], 'ms hinted')") || exit 1
done
