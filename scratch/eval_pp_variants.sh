#!/bin/bash
# usage (GPU box): scratch/eval_pp_variants.sh "<flags 1>" "<flags 2>" ...: rebuild csrc/eval_topk.hip with each flag set into a
# temp copy of the library (scratch/inst_build.sh) and time both forms of the sweep (scratch/eval_forms.py)
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip $v) || exit 1
  (cd "$GRAFT_REPO_ROOT" && YR_ENGINE_LIB="$lib" timeout -k 10 120 python3 scratch/eval_forms.py ${YR_FORMS_ARGS} 2>/dev/null | grep -v "four_waves:") || exit 1
done
