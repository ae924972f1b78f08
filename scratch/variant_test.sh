#!/bin/bash
# on the GPU box: rebuild bpr_pull with the given flags and run a pytest selection
flags=$1; shift
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c bpr_pull.hip -o bpr_pull.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
cd ../.. && python3 -m pytest "$@" -q 2>&1 | tail -4
