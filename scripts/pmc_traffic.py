#!/usr/bin/env python3
"""Sum the HBM-side traffic of one BPR-MF step from two rocprofv3 PMC passes of bench.py
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs as MI355X_MICROARCH.md prescribes) and
write profiles/traffic.json + a per-kernel CSV.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 128-byte requests as
64 bytes for wide streaming reads, so the read side is doubled (upper bound for narrow gathers);
WRITE_SIZE is taken as is.

usage: scripts/pmc_traffic.py <fetch_dir> <write_dir> <batch_per_gpu> <step_impl string> [name]
"""
import collections
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, batch, impl = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
name = sys.argv[5] if len(sys.argv) > 5 else "r01_pmc_traffic"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yr::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


# the launches of the STEP only (the input generation of bench.py — triplet_sample_kernel — runs once, before it)
import re
STEP_KERNELS = re.compile(r"tile_partition_kernel|owner_pass_kernel|bpr_fwd_bwd_kernel|adam_dual_kernel|adam_dense_kernel")
fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
fetch = {k: v for k, v in fetch.items() if STEP_KERNELS.search(k)}
write = {k: v for k, v in write.items() if STEP_KERNELS.search(k)}
rows, total = [], 0.0
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    hbm = 2 * f + w
    total += hbm
    rows.append((k, f, w, hbm))
with open(os.path.join(root, "profiles", name + ".csv"), "w", newline="") as fo:
    wr = csv.writer(fo)
    wr.writerow(["kernel", "FETCH_SIZE_bytes_raw", "WRITE_SIZE_bytes", "hbm_bytes_corrected(2*fetch+write)"])
    for r in rows:
        wr.writerow([r[0], int(r[1]), int(r[2]), int(r[3])])
    wr.writerow(["TOTAL per step", "", "", int(total)])
json.dump({"batch_per_gpu": batch, "step_impl": impl, "hbm_bytes_per_step": int(total),
           "source": f"profiles/{name}.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"},
          open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print("total HBM-side bytes per step:", int(total))
for r in rows:
    print(f"  {r[0][:60]:60s} fetch {r[1]/1e6:9.1f} MB  write {r[2]/1e6:9.1f} MB")
