#!/bin/bash
# Build an INSTRUMENTED copy of the engine library without touching the product objects or the product .so:
#   scratch/inst_build.sh <source.hip> <extra hipcc flags...>   ->  prints the path of the instrumented library
# The instrumented object and library go to a temp directory; run the measurement with YR_ENGINE_LIB=<that path>
# (the only place the override is read: yelprecommendation_amd/_lib.py).  The product objects in csrc/ are only READ.
set -e
src="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
csrc="$root/yelprecommendation_amd/csrc"
out="$(mktemp -d "${TMPDIR:-/tmp}/yr_inst.XXXXXX")"
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
[ "$src" = "eval_topk.hip" ] && flags="$flags -mllvm -amdgpu-mfma-vgpr-form"
make -s -C "$csrc" >&2                                   # the other objects: the product build, up to date
/opt/rocm/bin/hipcc $flags -I"$csrc" "$@" -c "$csrc/$src" -o "$out/${src%.hip}.o" >&2
objs=""
for o in "$csrc"/*.o; do [ "$(basename "$o")" = "${src%.hip}.o" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libyelprec_engine.so" $objs "$out/${src%.hip}.o" >&2
echo "$out/libyelprec_engine.so"
