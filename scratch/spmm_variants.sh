#!/bin/bash
# on the GPU box: rebuild ngcf.hip with each set of flags and time both SpMM forms
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $v -c ngcf.hip -o ngcf.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  (cd ../.. && python3 scratch/spmm_time.py) || exit 1
done
