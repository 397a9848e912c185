#!/usr/bin/env python3
"""Developer tool: settle parity soak -- N C2 scenes (seeds from BASE), GPU vs oracle, bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import oracle  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _settle_batch as SB, physics, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
sl.init_cuda(0)
meshes = synthetic.ycb_like_meshes(seed=0, tex_size=64)
scs = [bench.make_scene(sl, meshes, BASE + i) for i in range(N)]
se = physics.settle_engine()
planes = [(physics.prepare_tabletop(s), physics.PLANE_HALF_Z) for s in scs]
srec, bodies = SB.build_settle_batch(scs, se.pool, planes)
prm = SB.default_params(tabletop=True)
gpu = se.run(srec, bodies.copy(), prm)
hulls, verts = se.pool.arrays()
ref = bodies.copy()
t = time.time()
# the host path grows the lists when a heap needs it (SettleEngine.run): the oracle gets the capacities that run used
for key in ("max_hull_pairs_per_scene", "max_contacts_per_scene", "max_body_pairs_per_scene"):
    prm[key] = se.last_params[key]
oracle.settle(srec, ref, hulls, verts, prm)
bad = 0
for name in ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "wake_counter", "stab"):
    a, b = np.ascontiguousarray(gpu[name]), np.ascontiguousarray(ref[name])
    if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
        bad += 1
        print("MISMATCH in", name, "bodies:", np.unique(np.argwhere(a != b)[:, 0])[:10])
print("%d scenes, %d bodies: %s (oracle %.1f s)" % (N, len(ref), "bit-exact" if bad == 0 else "%d fields differ" % bad, time.time() - t))
sys.exit(1 if bad else 0)
