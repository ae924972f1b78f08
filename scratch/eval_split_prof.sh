#!/bin/bash
# usage (GPU box): scratch/eval_split_prof.sh ["<flags>" ...] — for each flag set (default: none) rebuild csrc/eval_topk.hip
# and print the rocprofv3 kernel stats of scratch/eval_split.py (all evaluation variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
[ $# -eq 0 ] && set -- ""
for v in "$@"; do
  echo "=== variant: $v"
  (cd yelprecommendation_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $v -c eval_topk.hip -o eval_topk.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o) || exit 1
  out=gpurun_out/evsplit; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 scratch/eval_split.py > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  grep -E "^k=(10|16) bf16x3 [0-9n]|^k=10 f32 [0-9]|k=4 bf16x3 no" $out/log.txt
  f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
  grep -E "mf_eval_topk_kernel<64, 4, false, true, true>" "$f" | cut -d, -f8- | cut -c1-60
done
