#!/bin/bash
# Developer probe (GPU box): the solver's LDS-resident contacts (SLHIP_LDS_CONTACTS) against the whole pipeline
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ldsc_probe.txt; : > $out
for v in $@; do
  SLHIP_EXTRA_FLAGS="-DSLHIP_LDS_CONTACTS=$v" python -c "import __graft_entry__ as g; import os; os.remove(g.LIB); g.build()" >/dev/null 2>&1
  echo "lds_contacts=$v" >> $out
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['breakdown_ms']
print(d['value'], d['ms_per_step'], 'settle', r['settle_ms_per_batch'], 'alone', r['settle_ms_per_batch_alone'], 'solve', r['ms_per_launch'], r['ms_per_launch_alone'], 'render_ov', b['render_total_overlapped'], 'spill', d['caps']['spill_step_rate'])" >> $out 2>&1
done
cat $out
