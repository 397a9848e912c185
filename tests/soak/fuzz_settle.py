#!/usr/bin/env python3
"""Developer tool: randomized settle parity fuzz -- batches of heaps with 1-40 bodies mixing cubes, bunnies (121
hulls: scenes beyond the pair-cache limit and the 8-per-CU LDS share) and YCB-like meshes, static bodies, plane on /
off, free-space random poses (overlaps), random frame counts / substeps / time steps, tabletop redrop on / off;
GPU vs oracle, every body bit for bit."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import scenes as S  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _settle_batch as SB, physics, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sl.init_cuda(0)


def scaled(path, diag):
    m = sl.Mesh(path)
    m.center_bbox()
    m.scale_to_bbox_diagonal(diag)
    return m


cube_s, cube_l, bunny = scaled(S.CUBE, 0.12), scaled(S.CUBE, 0.3), scaled(S.BUNNY, 0.2)
ycb = synthetic.ycb_like_meshes(seed=0, tex_size=8 if False else 64)
se = physics.settle_engine()
fields = ("pose", "lin_vel", "ang_vel", "separation", "flags", "stuck_counter", "wake_counter", "stab")
bad = 0
t0 = time.time()
for k in range(N):
    seed = BASE + k
    rng = np.random.default_rng(seed)
    scs, planes = [], []
    for b in range(int(rng.integers(1, 6))):
        scene = sl.Scene((320, 240), seed=seed * 8 + b)
        nb = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 28, 40]))
        for i in range(nb):
            r = rng.random()
            m = bunny if r < 0.08 else (cube_l if r < 0.2 else (cube_s if r < 0.5 else ycb[int(rng.integers(len(ycb)))]))
            o = sl.Object(m)
            if rng.random() < 0.05:
                o.static = True
            scene.add_object(o)
        if rng.random() < 0.6:
            on_table = physics.prepare_tabletop(scene)          # reference placement: a column above the table
            planes.append((on_table, physics.PLANE_HALF_Z))
        else:                                                   # free poses, possibly overlapping
            for o in scene._objects:
                p = np.eye(4, dtype=np.float32)
                p[:3, :3] = S.random_rotation(rng)
                p[:3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.05, 0.8)]
                o.set_pose(torch.from_numpy(p))
                if rng.random() < 0.3:
                    o.linear_velocity = torch.tensor(rng.uniform(-2, 2, 3).astype(np.float32))
            planes.append((bool(rng.integers(2)), 0.04))
        for o in scene._objects:                               # a few driven bodies (ManipulationSim's D6 joint)
            if rng.random() < 0.04 and not o._static:
                q = rng.standard_normal(4)
                o._drive = {"flags": 1 | (int(rng.integers(8)) << 1), "target": rng.uniform(-0.3, 0.3, 3).astype(np.float32),
                            "frame": (q / np.linalg.norm(q)).astype(np.float32), "stiffness": np.float32(rng.uniform(100, 900)),
                            "damping": np.float32(rng.uniform(0.05, 2.0)), "force_limit": np.float32(rng.uniform(5, 80))}
        scs.append(scene)
    prm = SB.default_params(tabletop=bool(rng.integers(2)), frames=int(rng.choice([1, 3, 10, 25, 60, 100, 250])),
                            substeps=int(rng.choice([1, 2, 4])), dt=float(rng.choice([0.002, 0.005, 0.01])))
    srec, bodies = SB.build_settle_batch(scs, se.pool, planes)
    gpu = se.run(srec, bodies.copy(), prm)
    hulls, verts = se.pool.arrays()
    ref = bodies.copy()
    # the host path grows the lists when a heap needs it (SettleEngine.run): the oracle gets the capacities that run used
    for key in ("max_hull_pairs_per_scene", "max_contacts_per_scene", "max_body_pairs_per_scene"):
        prm[key] = se.last_params[key]
    oracle.settle(srec, ref, hulls, verts, prm)
    ov_gpu = se.overlap(srec, bodies)                          # slhip_overlap_any on the initial state
    ov_ref = oracle.overlap_any(srec, bodies, hulls, verts)
    diff = [f for f in fields if not np.array_equal(np.ascontiguousarray(gpu[f]).view(np.uint8), np.ascontiguousarray(ref[f]).view(np.uint8))]
    if os.environ.get("FUZZ_VERBOSE"):
        moved = np.abs(gpu["pose"] - bodies["pose"]).max(axis=1)
        print("seed %d: %d scenes, bodies %s, frames %d x %d dt %.3f: moved>1mm %.2f, max |v| %.2f, asleep %.2f" % (
            seed, len(scs), [len(s._objects) for s in scs], prm["frames"], prm["substeps"], prm["dt"],
            float((moved > 1e-3).mean()), float(np.abs(gpu["lin_vel"]).max()), float(((gpu["flags"] & 2) != 0).mean())))
    if not np.array_equal(ov_gpu, ov_ref):
        diff.append("overlap_any")
    if diff:
        bad += 1
        print("MISMATCH seed %d (%d scenes, bodies %s, frames %d x %d, dt %.3f, tabletop %d): %s" % (
            seed, len(scs), [len(s._objects) for s in scs], prm["frames"], prm["substeps"], prm["dt"], prm["tabletop"], diff))
print("%d cases: %s (%.0f s)" % (N, "all bit-exact" if bad == 0 else "%d differ" % bad, time.time() - t0))
sys.exit(1 if bad else 0)
