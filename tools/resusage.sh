#!/bin/bash
# usage: tools/resusage.sh file.hip  -> per-kernel VGPR / AGPR / scratch / LDS / occupancy summary
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I/root/repo/include -I/root/repo/pixelwiseregression_amd/csrc -ffp-contract=off -x hip -c $f -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: print(); print(m.group(1)[:100],end=' | ')
    m=re.search(r'remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',line)
    if m: print(m.group(1).split(' ')[0]+'='+m.group(2),end=' ')
print()
"
