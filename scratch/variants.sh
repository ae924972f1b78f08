#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (on the GPU box): scratch/variants.sh "<flags variant 1>" "<flags variant 2>" ... ; env SIZES="32 65536 ..."
# rebuilds csrc/bpr_pull.hip with each set of extra compiler flags and times the step at every size
SIZES=${SIZES:-"32 4096 65536 262144 1048576"}
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" bpr_pull.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  for b in $SIZES; do (cd ../.. && python3 scratch/step_prof.py $b pull ${DIST:-synth} 100) || exit 1; done
done
