#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (GPU box): scratch/eval_shard_variants.sh "<flags>" ... — rebuild eval_topk.hip per flag set, run scratch/eval_shard.py
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && timeout -k 10 120 python3 scratch/eval_shard.py) || exit 1
done
