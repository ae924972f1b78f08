#!/bin/bash
# usage: scripts/kres.sh file.hip [extra flags]: per-kernel SGPR/VGPR/scratch/occupancy/LDS from the compiler remarks
f=$1; shift
cd "$(dirname "$0")/../yelprecommendation_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; vals={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); vals={}; continue
    m=re.search(r'remark:\s+([A-Za-z][\w \[\]/]*?): (\S+) \[-R',line)
    if m and cur:
        key=m.group(1).strip(); vals[key]=m.group(2)
        if key.startswith('LDS Size'):
            name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip()[:70]
            print(name.ljust(71),'sgpr',vals.get('TotalSGPRs'),'vgpr',vals.get('VGPRs'),'scratch',vals.get('ScratchSize [bytes/lane]'),'occ',vals.get('Occupancy [waves/SIMD]'),'lds',vals[key])
"
