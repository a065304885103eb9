#!/bin/bash
# Register / spill / occupancy table of the kernels of one .hip file (compile only, no GPU needed): tools/kres.sh csrc/pt_wavefront.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 | python3 -c "
import re,sys
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'remark: +(Function Name): (\S+)',l)
    if m: cur={'name':m.group(2)}; rows.append(cur); continue
    m=re.search(r'remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip().split('(')[0]
    print('%-34s VGPR %3d  AGPR %3d  spill %3d  SGPR %3d  sspill %3d  occ %d  LDS %6d  scratch %d'%(n[-34:],r.get('VGPRs',-1),r.get('AGPRs',0),r.get('VGPRs Spill',0),r.get('TotalSGPRs',-1),r.get('SGPRs Spill',0),r.get('Occupancy',-1),r.get('LDS Size',0),r.get('ScratchSize',0)))
"
