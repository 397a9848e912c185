#!/bin/bash
# usage: tools/kres.sh <file.hip> [filter]   -- registers / scratch / LDS / occupancy per kernel of one source file (no GPU needed)
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Istillleben_amd/csrc \
  --cuda-device-only -c "$1" -Rpass-analysis=kernel-resource-usage -o /dev/null 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: //; s/ *\[-Rpass.*//' \
  | python3 -c "
import sys,re
rows=[];cur=None
for l in sys.stdin:
    l=l.strip()
    if 'Function Name' in l:
        m=re.search(r'(k_\w+?)E[A-Z]',l); cur=[m.group(1) if m else l[-40:]]; rows.append(cur)
    elif cur is not None: cur.append(re.sub(r'\s+',' ',l))
for r in rows: print('%-22s %s'%(r[0],' | '.join(r[1:])))
" | grep -E "${2:-.}"
