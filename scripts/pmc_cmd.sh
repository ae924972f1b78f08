#!/bin/bash
# usage: scripts/pmc_cmd.sh <tag> <python script and args...>   (GPU box) — fabric-side traffic and L2 hit rate per
# yr:: kernel: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss), FETCH_SIZE doubled + WRITE_SIZE
# as MI355X_MICROARCH.md prescribes; writes gpurun_out/pmc_<tag>_summary.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$n -- python3 "$@" > gpurun_out/pmc_${tag}_$n.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in glob.glob('gpurun_out/pmc_${tag}_*/'):
    for f in glob.glob(d + '*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'yr::' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('(')[0].replace('void ', '')
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d + '*/*kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            if 'yr::' in r['Kernel_Name']:
                dur[r['Kernel_Name'].split('(')[0].replace('void ', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
avg = lambda v: sum(v) / len(v) if v else 0.0
with open('gpurun_out/pmc_${tag}_summary.csv', 'w') as fo:
    fo.write('kernel,launches,avg_us_under_pmc,fetch_MB(x2),write_MB,hbm_side_MB,hbm_side_GBps,l2_hit_rate\n')
    for k in sorted(acc, key=lambda k: -avg(dur[k]) * len(dur[k])):
        f, w = avg(acc[k]['FETCH_SIZE']) * 1024 * 2 / 1e6, avg(acc[k]['WRITE_SIZE']) * 1024 / 1e6
        h, ms = avg(acc[k]['TCC_HIT_sum']), avg(acc[k]['TCC_MISS_sum'])
        us = avg(dur[k])
        fo.write(f"{k},{len(dur[k])//3},{us:.1f},{f:.1f},{w:.1f},{f + w:.1f},{(f + w) / us * 1e3 if us else 0:.0f},{h / (h + ms) if h + ms else 0:.3f}\n")
print(open('gpurun_out/pmc_${tag}_summary.csv').read())
PY
