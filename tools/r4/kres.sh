#!/bin/bash
# per-kernel VGPRs / spills / scratch with the template arguments: tools/r4/kres.sh spx_group_team.hip [extra hipcc flags]
cd "$(dirname "$0")/../../shiftedproximaloperators.jl_amd/csrc"
f="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I. -c "$f" -o /tmp/kres.o "$@" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('spill',r'VGPRs Spill: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[key]=m.group(1)
for r in rows:
    name=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    m=re.search(r'(\w+(<[^(]*>)?)\(',name)
    print('%-60s vgpr=%s spill=%s scratch=%s occ=%s'%((m.group(1) if m else name)[:60],r.get('vgpr'),r.get('spill'),r.get('scratch'),r.get('occ')))
"
