#!/usr/bin/env python3
"""Copy the engine's rows (kernel names starting with `void yr::` / `yr::`) of a rocprofv3
`*_kernel_stats.csv` into profiles/<name>.csv, followed by the bench JSON line of that run.

usage: scripts/save_profile.py gpurun_out/<dir> <name> [bench_log]
"""
import csv
import glob
import os
import sys

src, name = sys.argv[1], sys.argv[2]
log = sys.argv[3] if len(sys.argv) > 3 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = sorted(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True))
if not stats:
    raise SystemExit(f"no *_kernel_stats.csv under {src}")
out = os.path.join(root, "profiles", name + ".csv")
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    for k, path in enumerate(stats):
        with open(path, newline="") as fi:
            rows = list(csv.reader(fi))
        if k == 0:
            w.writerow(rows[0])
        for r in rows[1:]:
            if "yr::" in r[0] or "rccl" in r[0].lower() or "nccl" in r[0].lower():
                w.writerow(r)
    if log and os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith('{"metric"'):
                fo.write("# bench: " + line.strip() + "\n")
print("wrote", out)
