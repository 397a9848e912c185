#!/usr/bin/env python3
"""Developer tool: turns rocprofv3 output directories (gpurun_out/...) into the summaries kept under
profiles/: per-kernel stats and the per-kernel HBM traffic JSON that bench.py's `roofline.traffic`
reads.  usage: summarize_profiles.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix> <tag>"""
import collections
import csv
import glob
import json
import re
import shutil
import sys


def kname(n):
    m = re.search(r"(k_\w+)\(", n)
    return m.group(1) if m else None


def main():
    stats_dir, fetch_dir, write_dir, out_prefix, tag = sys.argv[1:6]
    st = glob.glob(stats_dir + "/*/*_kernel_stats.csv")[0]
    shutil.copy(st, "%s/kernel_stats_%s.csv" % (out_prefix, tag))
    res = {}
    for c, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        f = glob.glob(d + "/*/*_counter_collection.csv")[0]
        acc = collections.defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if r["Counter_Name"] != c or not k:
                continue
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1].add(r["Dispatch_Id"])
        for k, (v, ids) in acc.items():
            res.setdefault(k, {})[c] = v * 1024 / len(ids)
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- "
                   "python bench.py --batch 256 --render-chunk 128 --steps 1 --warmup 0 --no-cpu-baseline --settle-streams 1",
        "units": "bytes per launch = counter * 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE calibrated exactly "
                 "against a 3.2 GB fill; FETCH_SIZE includes Infinity-Cache hits",
        "scenes_per_launch": {"render kernels": 128, "k_settle": 256}, "tag": tag, "kernels": {},
    }
    for k, v in sorted(res.items()):
        n = 256 if k == "k_settle" else 128
        f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        out["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w, "hbm_bytes_per_scene": (f + w) / n}
    json.dump(out, open("%s/pmc_traffic.json" % out_prefix, "w"), indent=1)
    for r in list(csv.DictReader(open(st)))[:12]:
        print("%-18s calls %4s avg %9.3f ms" % (kname(r["Name"]) or r["Name"][:18], r["Calls"], float(r["AverageNs"]) / 1e6))
    for k, v in out["kernels"].items():
        print("%-18s HBM per scene %8.2f MB (fetch %.1f MB, write %.1f MB per launch)" % (k, v["hbm_bytes_per_scene"] / 1e6, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
