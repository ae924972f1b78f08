#!/bin/bash
# on the GPU box: bpr_pull with -DYR_STAMPS in an instrumented COPY of the library (scratch/inst_build.sh; the
# product objects and the product .so are not written) and the workgroup timelines of the given batch sizes
lib=$("$(dirname "$0")/inst_build.sh" bpr_pull.hip -DYR_STAMPS $EXTRA) || exit 1
export YR_ENGINE_LIB="$lib"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && for b in "$@"; do echo "== B=$b"; python3 scratch/stamps.py $b; done
