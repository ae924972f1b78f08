#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (GPU box): scratch/cdae_variants.sh "<flags 1>" ...: rebuild csrc/cdae_sparse.hip with each flag set, time the step
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" cdae_sparse.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && python3 scratch/cdae_step_prof.py sampled 2>/dev/null | grep "fused step") || exit 1
done
