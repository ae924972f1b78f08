#!/bin/bash
# on the GPU box: rebuild bpr_pull with the given flags and run a pytest selection
flags=$1; shift
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" bpr_pull.hip $flags) || exit 1
export YR_ENGINE_LIB="$lib"
cd ../.. && python3 -m pytest "$@" -q 2>&1 | tail -4
