#!/usr/bin/env python3
"""Developer tool: re-renders soak scenes and prints every output at the pixels where GPU and oracle differ."""
import os
import sys

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

first, count = int(sys.argv[1]), int(sys.argv[2])
sl.init_cuda(0)
import test_gpu_render as T  # noqa: E402

meshes = synthetic.ycb_like_meshes(seed=0, tex_size=256)
eng = engine()
scs = [bench.make_scene(sl, meshes, first + i) for i in range(count)]
physics.settle_batch(scs)
for s in scs:
    s.choose_random_camera_pose()
    s.choose_random_light_direction()
for rep in range(3):
    bufs, ref = T.both(eng, oracle, scs, mask=_abi.OUT_ALL, ssao=False, shadows=False)
    g = bufs.instance.cpu().numpy().view(np.uint16)
    bad = np.argwhere(g != ref.instance)
    print("rep", rep, "differing instance pixels:", len(bad))
    for b in bad[:4]:
        n, y, x = int(b[0]), int(b[1]), int(b[2])
        print(" scene %d pixel (x=%d, y=%d)" % (n, x, y))
        for name in ("instance", "cls", "vertex_idx", "bary", "coord", "cam_coord"):
            ga = getattr(bufs, name).cpu().numpy()[n, y, x]
            ra = getattr(ref, name)[n, y, x]
            print("   %-10s gpu %s | oracle %s" % (name, ga, ra))
        for dy in (-1, 0, 1):
            print("   nbhd gpu", g[n, y + dy, x - 1:x + 2, 0], " oracle", ref.instance[n, y + dy, x - 1:x + 2, 0])
