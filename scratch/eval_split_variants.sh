#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (GPU box): scratch/eval_split_variants.sh "<flags 1>" ...: rebuild csrc/eval_topk.hip with each flag set, print the
# phase timeline (-DYR_ET_STAMPS) and the times of scratch/eval_split.py
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip -DYR_ET_STAMPS $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && python3 scratch/eval_phases.py) || exit 1
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && python3 scratch/eval_split.py 2>&1 | grep -E "k=(4|10|16) bf16x3 [0-9n]") || exit 1
done
