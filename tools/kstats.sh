#!/bin/bash
# usage: tools/kstats.sh <command...>   -- per-kernel rocprofv3 stats of a command, printed compactly (GPU box)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- "$@" > /tmp/prof_ks.log 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob("/tmp/prof_ks/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)",r["Name"]); n=m.group(1) if m else r["Name"][:28]
    print("%-18s calls %5s total %9.2f ms avg %8.1f us %6.2f%% min %7.1f max %8.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,float(r["Percentage"]),float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
