#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per engine kernel.  usage: scripts/pmc_summary.py <dir>..."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "yr::" not in k:
                continue
            acc[k.split("(")[0][-45:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k.ljust(46), " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
