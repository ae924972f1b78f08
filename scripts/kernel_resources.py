#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel in the built objects, read from the gfx950 code objects
(llvm-objdump --offloading + llvm-readelf --notes: the amdhsa.kernels metadata), and the occupancy budgets the
design depends on.  No GPU needed: `__graft_entry__.build()` calls check() after `make`, and so does a CPU test —
an edit that pushes a kernel over its budget fails the build instead of silently halving its occupancy
(the owner passes need 8 workgroups per CU, the split evaluation kernel 3 waves per SIMD).

    python scripts/kernel_resources.py            # table of all kernels + budget check
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yelprecommendation_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"

# (regex on the demangled kernel name, max VGPRs (arch + acc), max LDS bytes (static), why)
BUDGETS = [
    (r"void yr::mf_eval_topk_kernel<64, 10, (true|false), true, false>", 168, 54 * 1024,
     "split-form sweep, 10-entry lists: three waves per SIMD (512 / 3 -> 168) and three workgroups per CU"),
    (r"void yr::mf_eval_topk_kernel<64, 4, (true|false), true, false>", 168, 54 * 1024, "as above, 4-entry lists"),
    (r"void yr::mf_eval_topk_kernel<64, 10, (true|false), false, false>", 168, 54 * 1024, "f32-instruction sweep: three workgroups per CU"),
    (r"void yr::owner_pass_kernel<64, (true|false)", 64, 20 * 1024,
     "every bucket's workgroup resident at once: 8 workgroups of 256 threads per CU = 64 VGPRs, <= 20 KB LDS"),
    (r"void yr::owner_pass_kernel<(16|32|128), (true|false)", 64, 20 * 1024, "as above for the other widths"),
    (r"void yr::spmm_csr_kernel<64, (true|false), (true|false)>", 96, 0,
     "one wave per row with 8 gather passes in flight: five waves per SIMD (full and row-subset forms)"),
    (r"void yr::ngcf_dense_fwd_kernel<64, false>", 128, 0, "one wave per workgroup, four waves per SIMD"),
    (r"void yr::ngcf_dense_bwd_data_kernel<64, false>", 128, 0, "as the forward kernel"),
    (r"void yr::ngcf_dense_(fwd|bwd_data)_kernel<64, true>", 168, 0,
     "row-list forms: three waves per SIMD (a hoisted weight tile once took the forward kernel to 252)"),
]
# Scratch (spilled registers) per lane.  The forms the benchmark and the trainers run by default — width 64, summation
# order free — must have none; the deterministic-order forms and some forms of the other widths are held at 64 VGPRs by
# __launch_bounds__ and spill a few words (measured cost in profiles/r02_deterministic_mode_cost.txt) — capped here
# so that it cannot grow unnoticed.
SCRATCH_ALLOWED = {
    r"void yr::owner_pass_kernel<(16|32|64|128), (true|false), (true|false), true, [01]>": 64,   # deterministic order
    r"void yr::owner_pass_kernel<(16|32|128), (true|false), (true|false), false, [01]>": 16,
    r"void yr::topk_masked_kernel<32, 1024>": 160,       # unfused fallback for 16 < k <= 32 (a 32-entry list per thread)
}


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.strip().split("\n")


def read_object(obj):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, lds, scratch, spill)} of one .o (gfx950 bundle)."""
    tmp = tempfile.mkdtemp(prefix="yr_kres.")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, capture_output=True, check=True)
        cos = [f for f in glob.glob(local + ".*") if "gfx950" in f]
        if not cos:
            return {}
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", cos[0]], capture_output=True, text=True, check=True).stdout
        # scratch instructions per kernel symbol (a private segment can be RESERVED without ever being accessed:
        # what matters is scratch traffic)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", cos[0]], capture_output=True, text=True,
                             check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(-\s+)?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if key == "agpr_count" and dash:                   # first key of a kernel entry
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size",
                                       "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "name"):
            cur[key] = val.strip()
    kernels = [k for k in kernels if "name" in k]
    scratch_ops, sym = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            sym = m.group(1)
            scratch_ops[sym] = 0
        elif sym and re.search(r"\bscratch_(load|store)", line):
            scratch_ops[sym] += 1
    names = _demangle([k["name"] for k in kernels]) if kernels else []
    return {n: dict(vgpr=int(k["vgpr_count"]), agpr=int(k["agpr_count"]), sgpr=int(k["sgpr_count"]),
                    lds=int(k["group_segment_fixed_size"]), scratch=int(k["private_segment_fixed_size"]),
                    spill=int(k.get("vgpr_spill_count", 0)), scratch_ops=scratch_ops.get(k["name"], 0))
            for n, k in zip(names, kernels)}


def read_all():
    out = {}
    for obj in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
        for name, r in read_object(obj).items():
            r["file"] = os.path.basename(obj)
            out[name] = r
    return out


def check(kernels=None):
    """Raise RuntimeError listing every kernel over its budget; returns the kernel table."""
    kernels = read_all() if kernels is None else kernels
    if not kernels:
        raise RuntimeError("no gfx950 kernels found in the built objects (run make first)")
    bad = []
    for pat, max_vgpr, max_lds, why in BUDGETS:
        hit = [(n, r) for n, r in kernels.items() if re.match(pat, n)]
        if not hit:
            bad.append(f"budget pattern matches no kernel (renamed?): {pat}")
        for n, r in hit:
            if r["vgpr"] > max_vgpr or (max_lds and r["lds"] > max_lds):
                bad.append(f"{n[:100]}: {r['vgpr']} VGPRs / {r['lds']} B LDS over the budget of {max_vgpr} / {max_lds} ({why})")
    for n, r in kernels.items():
        allowed = max([v for p, v in SCRATCH_ALLOWED.items() if re.match(p, n)], default=0)
        # a reserved private segment that no instruction touches (scratch_ops == 0, no spilled VGPR) is not traffic
        used = r["scratch"] if (r.get("scratch_ops", 1) > 0 or r["spill"] > 0) else 0
        if used > allowed or (allowed == 0 and r["spill"] > 0):
            bad.append(f"{n[:100]}: {r['scratch']} B of scratch per lane, {r['spill']} spilled VGPRs, {r.get('scratch_ops', '?')} scratch instructions (spills are never acceptable on this path)")
    if bad:
        raise RuntimeError("kernel resource budgets exceeded:\n  " + "\n  ".join(bad))
    return kernels


if __name__ == "__main__":
    ks = read_all()
    for n, r in sorted(ks.items(), key=lambda kv: (kv[1]["file"], kv[0])):
        print(f"{r['file']:18s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']:4d} ({r['scratch_ops']:2d} ops)  {n[:110]}")
    try:
        check(ks)
        print(f"{len(ks)} kernels, all budgets met")
    except RuntimeError as e:
        print(e)
        sys.exit(1)
