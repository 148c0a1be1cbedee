#!/bin/bash
# usage: tools_resusage.sh file.hip  -> per-kernel VGPR / spill / LDS / occupancy summary
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I/root/repo/include -I/root/repo/pixelwiseregression_amd/csrc -ffp-contract=off -x hip -c $f -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); print(); print(cur[:110],end=' | ')
    for k in ('VGPRs:','AGPRs:','ScratchSize','Occupancy','LDS Size','SGPRs:'):
        if k in line: print(line.split('remark:')[1].split('[')[0].strip(),end='; ')
print()
"
