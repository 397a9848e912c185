#!/usr/bin/env python3
"""Developer tool: steady-state throughput of the settle half alone (stage + settle, no render) with N launches in flight.
usage: settle_throughput.py <scenes per launch> <streams> <launches>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stillleben_amd as sl
from stillleben_amd import synthetic
B, NS, NL = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hulls = sys.argv[4] if len(sys.argv) > 4 else "vhacd"
sl.init_cuda(0)
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64, hulls=hulls))
streams = [torch.cuda.Stream() for _ in range(NS)]
batches = [sl.SceneBatch(table, B, 20, seed=7) for _ in range(NS)]
def run(n, base):
    for k in range(n):
        with torch.cuda.stream(streams[k % NS]):
            batches[k % NS].stage(scene_id_base=(base + k) * B)
            batches[k % NS].settle()
    torch.cuda.synchronize()
run(NS, 0)
t = time.perf_counter()
run(NL, 100)
dt = time.perf_counter() - t
print("B=%d streams=%d launches=%d hulls=%s impl=%s: %.1f ms per launch, %.0f scenes/s" % (B, NS, NL, hulls, os.environ.get("SLHIP_SETTLE_IMPL", "default"), dt / NL * 1e3, B * NL / dt))
