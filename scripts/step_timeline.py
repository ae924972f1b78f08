#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 kernel trace: usage scripts/step_timeline.py <prof dir> <marker kernel substring>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [k for k, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("span us", (t1 - t0) / 1e3, "busy us", busy / 1e3, "launches", len(seg))
prev = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f} {r['Kernel_Name'][:60]}")
    prev = e
