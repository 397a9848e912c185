#!/usr/bin/env python3
"""Developer tool: is the host ahead of the GPU?  Enqueues a few pipelined steps (bench.Pipeline) and reports how long the
enqueue of each step kept the host busy against how long the GPU needs per step.  A step whose enqueue takes about as long
as its GPU work means something on the host path waits for the device (that is how the blocking read-out of the render phase
timers was found: settle and render of consecutive batches then never overlap)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
timing = len(sys.argv) > 3 and sys.argv[3] == "timing"
sl.init_cuda(0)
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0))
pipe = bench.Pipeline(sl, table, B, min(1024, B), True, 2, seed=1, rank=0)
if timing:
    pipe.eng.L.slhip_timing_enable(1)
    pipe.eng.L.slhip_settle_timing_enable(1)
pipe.launch_step(0, 0)
pipe.drain()
t0 = time.perf_counter()
host = []
for k in range(1, K + 1):
    t = time.perf_counter()
    pipe.launch_step(k, k * B)
    host.append(time.perf_counter() - t)
t_enq = time.perf_counter() - t0
pipe.drain()
t_all = time.perf_counter() - t0
print("%d steps of %d scenes (phase timers %s): host enqueue per step %s ms, all enqueued after %.0f ms, GPU done after %.0f ms"
      % (K, B, "on" if timing else "off", " ".join("%.0f" % (h * 1e3) for h in host), t_enq * 1e3, t_all * 1e3))
