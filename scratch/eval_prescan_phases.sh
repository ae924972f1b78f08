#!/bin/bash
# (variants are built into a temp copy of the library: scratch/inst_build.sh; the product .so is untouched)
# usage (GPU box): scratch/eval_prescan_phases.sh — phase timeline of the sweep with and without the prescan thresholds
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
lib=$("$GRAFT_REPO_ROOT/scratch/inst_build.sh" eval_topk.hip -DYR_ET_STAMPS $EXTRA) || exit 1
export YR_ENGINE_LIB="$lib"
cd ../..
for k in 10 16; do for ps in 0 1; do echo "k=$k prescan=$ps"; YR_K=$k YR_PRESCAN=$ps python3 scratch/eval_phases.py || exit 1; done; done
