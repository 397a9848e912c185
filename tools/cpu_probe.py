#!/usr/bin/env python3
"""Developer tool (GPU box): what the host really offers bench.py's cpu_baseline -- affinity mask, cgroup quota, load -- and how a
pure-compute loop scales over threads and over processes (the signature of a quota is a plateau at quota / period cores)."""
import json
import multiprocessing as mp
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_cpu_facts():
    out = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
        try:
            out[p] = open(p).read().strip()
        except OSError:
            out[p] = None
    return out


def spin(_):
    import numpy as np
    a = np.random.default_rng(0).standard_normal((256, 256))
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:
        a = a @ a
        a /= abs(a).max()
        n += 1
    return n


if __name__ == "__main__":
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    print(json.dumps(host_cpu_facts()))
    base = None
    for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if n > (os.cpu_count() or 1):
            break
        with mp.get_context("fork").Pool(n) as pool:
            w0 = time.perf_counter()
            tot = sum(pool.map(spin, range(n)))
            wall = time.perf_counter() - w0
        base = base or tot
        print("processes %4d: %8d iterations in %.2f s = %.1f x one" % (n, tot, wall, tot / base))
