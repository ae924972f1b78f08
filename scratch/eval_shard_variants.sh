#!/bin/bash
# usage (GPU box): scratch/eval_shard_variants.sh "<flags>" ... — rebuild eval_topk.hip per flag set, run scratch/eval_shard.py
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $v -c eval_topk.hip -o eval_topk.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  (cd ../.. && timeout -k 10 120 python3 scratch/eval_shard.py) || exit 1
done
