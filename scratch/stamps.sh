#!/bin/bash
# on the GPU box: rebuild bpr_pull with -DYR_STAMPS into a scratch copy of the library and print WG timelines
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DYR_STAMPS $EXTRA -c bpr_pull.hip -o bpr_pull.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
cd ../.. && for b in "$@"; do echo "== B=$b"; python3 scratch/stamps.py $b; done
