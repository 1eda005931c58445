#!/bin/bash
# usage: tools/resource_usage.sh file.hip  -> per-kernel VGPR / scratch / occupancy summary
hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/ru.o 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); print(); print(cur[:60],end=' ')
    for key in ('VGPRs:','AGPRs:','ScratchSize','Occupancy','TotalSGPRs','LDS Size'):
        m=re.search(key+r'[^:]*:? *([0-9]+)',l)
        if m and cur: print(key.strip(':')+'='+m.group(1),end=' ')
print()
"
