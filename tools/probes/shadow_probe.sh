#!/bin/bash
# Developer probe (GPU box): the shadow raster's LDS window (SLHIP_SHADOW_WINDOW texels per side; 0 = global atomics only)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/shadow_probe.txt; : > $out
for v in "$@"; do
  SLHIP_EXTRA_FLAGS="-DSLHIP_SHADOW_WINDOW=$v" python -c "import __graft_entry__ as g; import os; os.remove(g.LIB); g.build()" >/dev/null 2>&1
  echo "window=$v" >> $out
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['breakdown_ms']; i=d['breakdown_isolated_ms']
print(d['value'], d['ms_per_step'], 'settle', r['settle_ms_per_batch'], 'render_ov', b['render_total_overlapped'], 'render_iso', b['render_total_isolated'], 'shadow iso', i['shadow_raster'], i['shadow_large'])" >> $out 2>&1
done
cat $out
