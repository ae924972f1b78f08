#!/usr/bin/env python3
"""Turn gpurun_out/r02 (scripts/collect_r02.sh) into the committed round-2 evidence under profiles/:
kernel stats of the default bench command and of the step at the swept batch sizes, the PMC traffic table of the
default bench command (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes, + WRITE_SIZE, L2 hit rate from the
TCC pass), profiles/traffic.json (what bench.py reports as `traffic`) and the bench line itself."""
import collections, csv, glob, json, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "r02"), os.path.join(root, "profiles")


def stats_file(d):
    return sorted(glob.glob(os.path.join(src, d, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]


bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
B = bench["config"]["batch_per_gpu"]
shutil.copy(stats_file("prof_bench"), os.path.join(dst, f"r02_bench_default_b{B}_kernel_stats.csv"))
for b in (32, 4096, 65536, 262144):
    shutil.copy(stats_file(f"prof_b{b}"), os.path.join(dst, f"r02_step_b{b}_kernel_stats.csv"))
json.dump(bench, open(os.path.join(dst, "r02_bench_default_run.json"), "w"), indent=1)


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(os.path.join(src, d, "**", "*_counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:                 # the latest pass only (gpurun_out accumulates)
        for r in csv.DictReader(open(f)):
            if "yr::" in r["Kernel_Name"] and "triplet_sample" not in r["Kernel_Name"]:   # the sampler is set-up, not the step
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


fetch, write, tcc = counters("pmc_bench_FETCH_SIZE"), counters("pmc_bench_WRITE_SIZE"), counters("pmc_bench_TCC_HIT_sum")
avg_us = {}
for r in csv.DictReader(open(stats_file("prof_bench"))):
    avg_us[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e3
rows, total = [], 0.0
for k in fetch:
    f, w = 2 * fetch[k]["FETCH_SIZE"] * 1024 / 1e6, write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
    hit, miss = tcc.get(k, {}).get("TCC_HIT_sum", 0.0), tcc.get(k, {}).get("TCC_MISS_sum", 0.0)
    us = avg_us.get(k, float("nan"))
    rows.append((k, us, f, w, f + w, (f + w) / us * 1e3 if us else float("nan"), hit / (hit + miss) if hit + miss else float("nan")))
    total += f + w
rows.sort(key=lambda r: -r[4])
name = "r02_bench_default_pmc_traffic.csv"
with open(os.path.join(dst, name), "w") as fo:
    fo.write("kernel,avg_us(kernel stats run),fetch_MB(x2),write_MB,hbm_side_MB,hbm_side_GBps,l2_hit_rate\n")
    for r in rows:
        fo.write(f"{r[0]},{r[1]:.1f},{r[2]:.1f},{r[3]:.1f},{r[4]:.1f},{r[5]:.0f},{r[6]:.3f}\n")
    fo.write(f"TOTAL per step,,,,{total:.1f},,\n")
json.dump({"batch_per_gpu": B, "step_impl": bench["config"]["step_impl"], "hbm_bytes_per_step": int(total * 1e6),
           "hbm_bytes_dominant_kernel": int(rows[0][4] * 1e6),
           "source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC passes of `bench.py --no-cpu-baseline "
                     "--no-sweep`, round 2; FETCH_SIZE doubled)"}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, name)).read())
print({k: bench[k] for k in ("value", "ms_per_step")}, bench["roofline"]["frac"], bench["roofline"]["frac_with_adam_bytes"])
print({k: v["us_per_step"] for k, v in bench["batch_sweep"].items()})
