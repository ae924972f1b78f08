#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (GPU box): scratch/eval_split_prof.sh ["<flags>" ...] — for each flag set (default: none) rebuild csrc/eval_topk.hip
# and print the rocprofv3 kernel stats of scratch/eval_split.py (all evaluation variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
[ $# -eq 0 ] && set -- ""
for v in "$@"; do
  echo "=== variant: $v"
  lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip $v) || exit 1
  export YR_ENGINE_LIB="$lib"
  out=gpurun_out/evsplit; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 scratch/eval_split.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  grep -E "^k=(10|16) bf16x3 [0-9n]|^k=10 f32 [0-9]|k=4 bf16x3 no" $out/log.txt
  f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
  grep -E "mf_eval_topk_kernel<64, 4, false, true, true>" "$f" | cut -d, -f8- | cut -c1-60
done
