#!/bin/bash
# phase cycles of the evaluation sweep (-DYR_ET_STAMPS) on an instrumented COPY of the library (scratch/inst_build.sh)
lib=$("$(dirname "$0")/inst_build.sh" eval_topk.hip -DYR_ET_STAMPS $EXTRA) || exit 1
export YR_ENGINE_LIB="$lib"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && python3 scratch/eval_phases.py
