#!/usr/bin/env python3
"""Developer tool: renders one batch of C2 scenes (unsettled random heaps -- the render cost does
not depend on how the poses were produced) a few times; run under rocprofv3 to collect kernel
stats or PMC counters for the render kernels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, synthetic  # noqa: E402
from stillleben_amd._context import engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sl.init_cuda(0)
meshes = synthetic.ycb_like_meshes(seed=0)
rng = np.random.default_rng(0)
scenes = []
for i in range(B):
    s = bench.make_scene(sl, meshes, i)
    for o in s.objects:
        p = np.eye(4, dtype=np.float32)
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        from stillleben_amd._math import quat_to_matrix
        p[:3, :3] = quat_to_matrix(q)
        p[:3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.08, 0.25)]
        o.set_pose(torch.from_numpy(p))
    s.choose_random_camera_pose()
    s.choose_random_light_direction()
    scenes.append(s)
import ctypes as C  # noqa: E402

eng = engine()
MASK = int(os.environ.get("SLHIP_PROF_MASK", str(_abi.OUT_GT6)), 0)
SSAO = os.environ.get("SLHIP_PROF_SSAO", "1") != "0"
bufs = eng.render(scenes, MASK, ssao=SSAO, shadows=True, buffers=None)   # warm-up
torch.cuda.synchronize()
eng.L.slhip_timing_enable(1)
ms = (C.c_float * 8)()
eng.L.slhip_render_timings(C.byref(ms))
for _ in range(REPS):
    bufs = eng.render(scenes, MASK, ssao=SSAO, shadows=True, buffers=bufs)
torch.cuda.synchronize()
if eng.L.slhip_render_timings(C.byref(ms)) == 0:
    names = ["shadow_raster", "shadow_large", "vis_raster", "vis_large", "shade", "ssao", "ssao_apply", "tonemap"]
    print("phase ms per render of %d scenes:" % B, {n: round(v / REPS, 2) for n, v in zip(names, ms)})
inst = bufs.instance.cpu().numpy()
print("rendered", B, "scenes x", REPS, "; covered pixels per scene: %.0f" % ((inst != 0).sum() / B))
