#!/bin/bash
# Developer probe (GPU box): LDS part of the solver x scenes per wave.  usage: ldsc2_probe.sh "<lds contacts> <spw>" ...
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ldsc2_probe.txt; : > $out
for v in "$@"; do
  set -- $v
  SLHIP_EXTRA_FLAGS="-DSLHIP_LDS_CONTACTS=$1" python -c "import __graft_entry__ as g; import os; os.remove(g.LIB); g.build()" >/dev/null 2>&1
  echo "lds_contacts=$1 spw=$2" >> $out
  SLHIP_SOLVE_SPW=$2 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['breakdown_ms']
print(d['value'], d['ms_per_step'], 'settle', r['settle_ms_per_batch'], 'alone', r['settle_ms_per_batch_alone'], 'solve', r['ms_per_launch'], r['ms_per_launch_alone'], 'render_ov', b['render_total_overlapped'], 'spill', d['caps']['spill_step_rate'])" >> $out 2>&1
done
cat $out
