#!/bin/bash
# usage (GPU box): scratch/cdae_variants.sh "<flags 1>" ...: rebuild csrc/cdae_sparse.hip with each flag set, time the step
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $v -c cdae_sparse.hip -o cdae_sparse.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  (cd ../.. && python3 scratch/cdae_step_prof.py sampled 2>/dev/null | grep "fused step") || exit 1
done
