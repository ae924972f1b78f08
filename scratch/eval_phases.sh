#!/bin/bash
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DYR_ET_STAMPS $EXTRA -c eval_topk.hip -o eval_topk.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
cd ../.. && python3 scratch/eval_phases.py
