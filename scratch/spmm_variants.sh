#!/bin/bash
# on the GPU box: rebuild ngcf.hip with each set of flags and time both SpMM forms
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" ngcf.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  (cd ../.. && python3 scratch/spmm_time.py) || exit 1
done
