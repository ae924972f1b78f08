#!/usr/bin/env python3
"""gpurun_out/r02c (scripts/collect_r02_cdae.sh) -> profiles/: kernel stats of the three CDAE step forms and of a
list-fed epoch, PMC traffic table of the sampled-decoder step (FETCH_SIZE doubled + WRITE_SIZE, separate passes),
bench lines of the other workloads, epoch timings."""
import csv, glob, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "r02c"), os.path.join(root, "profiles")
latest = lambda pat: sorted(glob.glob(os.path.join(src, pat), recursive=True), key=os.path.getmtime)[-1]
for f, name in (("sampled", "sampled"), ("dense", "dense"), ("auto", "autograd")):
    shutil.copy(latest(f"prof_cdae_{f}/**/*_kernel_stats.csv"), os.path.join(dst, f"r02_cdae_step_{name}_kernel_stats.csv"))
shutil.copy(latest("prof_cdae_epoch/**/*_kernel_stats.csv"), os.path.join(dst, "r02_cdae_epoch_lists_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_other_workloads.jsonl"), os.path.join(dst, "r02_bench_other_workloads.jsonl"))
with open(os.path.join(dst, "r02_cdae_epoch_ms.txt"), "w") as o:
    for f in ("cdae_epoch_dense.txt", "cdae_epoch_lists.txt"):
        o.write("".join(l for l in open(os.path.join(src, f)) if "amdgpu" not in l))


def per(counter):
    d = {}
    for r in csv.DictReader(open(latest(f"pmc_cdae_{counter}/**/*counter_collection.csv"))):
        k = r["Kernel_Name"]
        d.setdefault(k, [0, 0.0]); d[k][0] += 1; d[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in d.items()}


fe, wr = per("FETCH_SIZE"), per("WRITE_SIZE")
st = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(dst, "r02_cdae_step_sampled_kernel_stats.csv")))}
rows = []
for k, (n, f) in fe.items():
    if "yr::" in k and k in st:
        w = wr.get(k, (0, 0.0))[1]
        rows.append((k.replace("void ", "").split("(")[0], n, st[k], 2 * f * 1024 / 1e6, w * 1024 / 1e6))
rows.sort(key=lambda r: -(r[3] + r[4]))
with open(os.path.join(dst, "r02_cdae_step_sampled_pmc_traffic.csv"), "w") as o:
    o.write("kernel,launches,avg_us(stats run),fetch_MB(x2),write_MB,hbm_side_MB,hbm_side_GBps\n")
    for k, n, us, f, w in rows:
        o.write(f"{k},{n},{us:.1f},{f:.1f},{w:.1f},{f + w:.1f},{(f + w) / us * 1e3:.0f}\n")
print(open(os.path.join(dst, "r02_cdae_step_sampled_pmc_traffic.csv")).read())
