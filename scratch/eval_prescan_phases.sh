#!/bin/bash
# usage (GPU box): scratch/eval_prescan_phases.sh — phase timeline of the sweep with and without the prescan thresholds
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DYR_ET_STAMPS $EXTRA -c eval_topk.hip -o eval_topk.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
cd ../..
for k in 10 16; do for ps in 0 1; do echo "k=$k prescan=$ps"; YR_K=$k YR_PRESCAN=$ps python3 scratch/eval_phases.py || exit 1; done; done
