#!/bin/bash
# compact per-kernel resource usage: tools/kres.sh spx_group.hip [filter]
cd "$(dirname "$0")/../shiftedproximaloperators.jl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I. -c "$1" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('sgpr',r' SGPRs: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('spill',r'VGPRs Spill: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[key]=m.group(1)
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r in rows:
    name=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    name=re.sub(r'\(.*','',name)
    if flt in name: print('%-60s vgpr=%s sgpr=%s occ=%s spill=%s scratch=%s lds=%s'%(name[:60],r.get('vgpr'),r.get('sgpr'),r.get('occ'),r.get('spill'),r.get('scratch'),r.get('lds')))
" "${2:-}"
