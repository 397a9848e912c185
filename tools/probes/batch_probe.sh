#!/bin/bash
# Developer probe (GPU box): step size x settle streams.  usage: batch_probe.sh "<bench args>" ...
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/batch_probe.txt; : > $out
for v in "$@"; do
  echo "$v" >> $out
  timeout 900 python bench.py --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['breakdown_ms']
print(d['value'], d['ms_per_step'], 'settle', r['settle_ms_per_batch'], 'alone', r['settle_ms_per_batch_alone'], 'render_ov', b['render_total_overlapped'])" >> $out 2>&1
done
cat $out
