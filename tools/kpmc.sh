#!/bin/bash
# usage: tools/kpmc.sh "<counters>" <command...>  -- per-kernel PMC averages (GPU box)
C="$1"; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_pm && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_pm -- "$@" > /tmp/prof_pm.log 2>&1
python3 - <<PY
import csv,glob,re,collections
f=glob.glob("/tmp/prof_pm/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); ids=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)",r["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"]); ids[m.group(1)].add(r["Dispatch_Id"])
for k,v in acc.items():
    n=len(ids[k]); print(k, n, {c:round(x/n) for c,x in sorted(v.items())})
PY
