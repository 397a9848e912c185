#!/bin/bash
# Developer probe (GPU box): list capacities / pair budget against the whole pipeline.  usage: cap_probe.sh "ENV=.. ENV=.." ...
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cap_probe.txt; : > $out
for v in "$@"; do
  echo "$v" >> $out
  env $v timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['breakdown_ms']; c=d['caps']
print(d['value'], d['ms_per_step'], 'settle', r['settle_ms_per_batch'], 'alone', r['settle_ms_per_batch_alone'], 'solve', r['ms_per_launch'], r['ms_per_launch_alone'], 'render_ov', b['render_total_overlapped'], 'drops', c['contact_drop_steps'], c['pair_drop_steps'], 'max', c['most_contacts_in_a_step'], c['most_hull_pairs_in_a_step'])" >> $out 2>&1
done
cat $out
