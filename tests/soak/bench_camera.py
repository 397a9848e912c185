#!/usr/bin/env python3
"""Developer tool: throughput of slhip_camera_model (row f4) on a batch of 640x480 images, with the
oracle timed beside it on one host core.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import camera_model as cm  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NOISE = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
sl.init_cuda(0)
rng = np.random.default_rng(0)
img = torch.rand(B, 3, 480, 640, device="cuda")
ps = [cm.make_params(rng.uniform(-0.002, 0.002, (3, 2)), rng.uniform(0.998, 1.002, 3), 1.5, 0.3, NOISE, 0.02, 0.01, 0.02, seed=i)
      for i in range(B)]
cm.process_batch(img, ps)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
REPS = 5
for _ in range(REPS):
    out = cm.process_batch(img, ps)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / REPS
alg = B * 3 * 480 * 640 * 4 * 4   # read in, write + read the intermediate, write out
import oracle  # noqa: E402

pd = [cm.make_params(rng.uniform(-0.002, 0.002, (3, 2)), rng.uniform(0.998, 1.002, 3), 1.5, 0.3, False, 0, 0, 0.02, seed=0)]
t0 = time.perf_counter()
oracle.camera_model(img[:1].cpu().numpy(), pd)
t_cpu = time.perf_counter() - t0
print(json.dumps({"metric": "camera model images/s (640x480, blur on, noise %s)" % ("on" if NOISE else "off"),
                  "value": B / (ms * 1e-3), "ms_per_batch": ms, "batch": B,
                  "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_image": alg // B},
                  "cpu_baseline": {"value": 1.0 / t_cpu, "unit": "images/s", "cores": 1, "kind": "port",
                                   "sample": "1 image, deterministic stages (oracle/camera_ref.c)"}}))
