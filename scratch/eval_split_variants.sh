#!/bin/bash
# usage (GPU box): scratch/eval_split_variants.sh "<flags 1>" ...: rebuild csrc/eval_topk.hip with each flag set, print the
# phase timeline (-DYR_ET_STAMPS) and the times of scratch/eval_split.py
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DYR_ET_STAMPS $v -c eval_topk.hip -o eval_topk.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  (cd ../.. && python3 scratch/eval_phases.py) || exit 1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $v -c eval_topk.hip -o eval_topk.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  (cd ../.. && python3 scratch/eval_split.py 2>&1 | grep -E "k=(4|10|16) bf16x3 [0-9n]") || exit 1
done
