#!/bin/bash
# usage (on the GPU box): scratch/variants.sh "<flags variant 1>" "<flags variant 2>" ... ; env SIZES="32 65536 ..."
# rebuilds csrc/bpr_pull.hip with each set of extra compiler flags and times the step at every size
SIZES=${SIZES:-"32 4096 65536 262144 1048576"}
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $v -c bpr_pull.hip -o bpr_pull.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  for b in $SIZES; do (cd ../.. && python3 scratch/step_prof.py $b pull ${DIST:-synth} 100) || exit 1; done
done
