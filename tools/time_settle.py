#!/usr/bin/env python3
"""Developer tool (GPU box): one settle of B C2 scenes with the GPU to itself, per-kernel launch averages from the
library's own HIP events (slhip_settle_timings) -- the quick loop for work on the k_w_* kernels.
    python tools/time_settle.py [B=16384] [repeats=2] [pair_contact_budget = sl.SceneBatch's default, 0]
SLHIP_BY_STEP=file.csv: the last repeat times EVERY step and writes step, ms of the five kernels (profiles/rNN/solve_by_step.csv)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sl.init_cuda(0)
table = sl.AssetTable(synthetic.ycb_like_meshes(seed=0, tex_size=64))
batch = sl.SceneBatch(table, B, 20, seed=20260929)
if len(sys.argv) > 3:
    batch.settle_params["pair_contact_budget"] = int(sys.argv[3])
L = _abi.lib()
batch.stage()
batch.settle(frames=2)
torch.cuda.synchronize()
names = ("k_w_begin", "k_w_gjk_first+rest", "k_w_manifold", "k_w_finish", "k_w_solve")
for r in range(REP):
    batch.stage(scene_id_base=(r + 1) * B)
    torch.cuda.synchronize()
    by_step = os.environ.get("SLHIP_BY_STEP") if r == REP - 1 else None
    L.slhip_settle_timing_every(1 if by_step else 8)
    L.slhip_settle_timing_enable(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    batch.settle()
    e1.record()
    torch.cuda.synchronize()
    ms, n = (C.c_float * 5)(), (C.c_uint32 * 5)()
    if by_step:
        import numpy as np

        rows, steps, nr = np.zeros((400, 5), np.float32), np.zeros(400, np.uint32), C.c_uint32(0)
        L.slhip_settle_timings_by_step(C.c_void_p(rows.ctypes.data), C.c_void_p(steps.ctypes.data), 400, C.byref(nr))
        with open(by_step, "w") as f:
            f.write("step," + ",".join(names) + "\n")
            for i in range(nr.value):
                f.write("%d," % steps[i] + ",".join("%.4f" % v for v in rows[i]) + "\n")
        for i in range(5):
            ms[i] = float(rows[:nr.value, i].mean())
    else:
        L.slhip_settle_timings(C.byref(ms), C.byref(n))
    L.slhip_settle_timing_enable(0)
    batch.check_settled()
    cp = batch.settle_caps()
    print("B=%d settle %.1f ms | " % (B, e0.elapsed_time(e1)) + "  ".join("%s %.3f" % (k, ms[i]) for i, k in enumerate(names))
          + " | sum %.3f | max contacts %d pairs %d reduced %d" % (sum(ms), cp["max_contacts"], cp["max_hull_pairs"], cp.get("reduced_steps", -1)))
