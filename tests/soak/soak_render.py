#!/usr/bin/env python3
"""Developer tool: render parity soak -- N settled C2 scenes (seeds from BASE) at 640x480 with shadows + SSAO,
GPU vs oracle: geometric outputs bit for bit, rgb to the 8-bit tolerance of the tests."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import oracle  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, physics, synthetic  # noqa: E402
from stillleben_amd._context import engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
sl.init_cuda(0)
import test_gpu_render as T  # noqa: E402

meshes = synthetic.ycb_like_meshes(seed=0, tex_size=256)
eng = engine()
mask = _abi.OUT_GT6 | _abi.OUT_CAM_COORD
bad = 0
t0 = time.time()
for c0 in range(0, N, 4):
    scs = [bench.make_scene(sl, meshes, BASE + i) for i in range(c0, min(N, c0 + 4))]
    physics.settle_batch(scs)
    for s in scs:
        s.choose_random_camera_pose()
        s.choose_random_light_direction()
    bufs, ref = T.both(eng, oracle, scs, mask=mask)
    try:
        T.assert_geometry_equal(bufs, ref, mask=mask)
        T.assert_rgb_close(bufs, ref)
    except AssertionError as e:
        bad += 1
        print("MISMATCH scenes %d..%d: %s" % (c0, c0 + len(scs) - 1, str(e)[:300]))
print("%d scenes: %s (%.0f s)" % (N, "all outputs within the bar" if bad == 0 else "%d chunks differ" % bad, time.time() - t0))
sys.exit(1 if bad else 0)
