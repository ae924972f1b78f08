#!/bin/bash
# usage (GPU box): scratch/eval_variants.sh "<flags 1>" "<flags 2>" ...: rebuild csrc/eval_topk.hip with each flag set,
# time the full-size evaluation (bench.py --workload eval)
cd "$GRAFT_REPO_ROOT/yelprecommendation_amd/csrc" || exit 1
for v in "$@"; do
  echo "=== variant: $v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $v -c eval_topk.hip -o eval_topk.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyelprec_engine.so *.o || exit 1
  (cd ../.. && python3 bench.py --workload eval 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'ms', d['roofline']['frac'])") || exit 1
done
