#!/bin/bash
# Developer probe (GPU box): settle streams x render streams after the caps went (the solver's launch time follows the heaviest scene)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/streams_probe.txt; : > $out
for ss in 1 2 3; do
  echo "settle_streams=$ss" >> $out
  timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --settle-streams $ss 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'settle', r['settle_ms_per_batch'], 'alone', r['settle_ms_per_batch_alone'], 'solve', r['ms_per_launch'], r['ms_per_launch_alone'])" >> $out 2>&1
done
cat $out
