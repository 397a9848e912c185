#!/usr/bin/env python3
"""Developer tool: randomized render parity fuzz -- scenes of cubes / bunnies / YCB-like meshes with random
viewport sizes, cameras (incl. very close and far ones), object counts, SSAO / shadow / exposure settings and
output masks; GPU vs oracle, geometry bit for bit, rgb to the tests' bar."""
import math
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
from stillleben_amd import _abi, synthetic  # noqa: E402
from stillleben_amd._context import engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sl.init_cuda(0)
import test_gpu_render as T  # noqa: E402

eng = engine()
ycb = synthetic.ycb_like_meshes(seed=0, tex_size=64)
# meshes are created ONCE: every sl.Mesh registers its vertices and textures in the engine's pool for good
CUBE = sl.Mesh(S.CUBE, physics=False)
CUBE.center_bbox()
CUBE.scale_to_bbox_diagonal(0.2)
BUNNY = sl.Mesh(S.BUNNY, physics=False)
BUNNY.center_bbox()
BUNNY.scale_to_bbox_diagonal(0.3)
BUNNY.class_index = 2


def clutter(seed, n_objects, size, with_bunny, plane, light):
    rng = np.random.default_rng(seed)
    scene = sl.Scene(size, seed=seed)
    meshes = [CUBE, BUNNY] if with_bunny else [CUBE]
    for i in range(n_objects):
        obj = sl.Object(meshes[i % len(meshes)])
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = S.random_rotation(rng)
        pose[:3, 3] = [rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(0.05, 0.3)]
        obj.set_pose(torch.from_numpy(pose))
        obj.metallic, obj.roughness = float(rng.uniform(0, 1)), float(rng.uniform(0, 1))
        scene.add_object(obj)
    az = rng.uniform(-math.pi, math.pi)
    cam = np.array([1.2 * math.cos(az), 1.2 * math.sin(az), 0.9], dtype=np.float32)
    scene.set_camera_look_at(torch.from_numpy(cam), torch.tensor([0.0, 0.0, 0.1]))
    if plane:
        scene.background_plane_size = torch.tensor([3.0, 3.0])
    if light:
        scene.choose_random_light_direction()
    scene.ambient_light = torch.tensor([0.1, 0.1, 0.1])
    return scene


sizes = [(320, 240), (160, 120), (333, 187), (64, 48), (640, 480), (257, 129)]
bad = 0
t0 = time.time()
for k in range(N):
    seed = BASE + k
    rng = np.random.default_rng(seed)
    size = sizes[int(rng.integers(len(sizes)))]
    batch = []
    for b in range(int(rng.integers(1, 4))):
        sc = clutter(seed * 8 + b, int(rng.integers(0, 9)), size, bool(rng.integers(2)), bool(rng.integers(4)), bool(rng.integers(5)))
        for _ in range(int(rng.integers(0, 4))):      # textured YCB-like objects on top
            o = sl.Object(ycb[int(rng.integers(len(ycb)))])
            p = np.eye(4, dtype=np.float32)
            p[:3, :3] = S.random_rotation(rng)
            p[:3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.0, 0.4)]
            o.set_pose(torch.from_numpy(p))
            o.metallic, o.roughness = float(rng.uniform(0, 1)), float(rng.uniform(0, 1))
            sc.add_object(o)
        mode = int(rng.integers(4))
        if mode == 0:      # very close camera (near-plane clipping, SSAO behind the eye)
            az = rng.uniform(-math.pi, math.pi)
            d = rng.uniform(0.12, 0.4)
            sc.set_camera_look_at(torch.tensor([d * math.cos(az), d * math.sin(az), rng.uniform(0.05, 0.4)], dtype=torch.float32),
                                  torch.tensor([0.0, 0.0, 0.1]))
        elif mode == 1:    # far camera (far plane in view)
            az = rng.uniform(-math.pi, math.pi)
            sc.set_camera_look_at(torch.tensor([6.0 * math.cos(az), 6.0 * math.sin(az), rng.uniform(0.3, 3.0)], dtype=torch.float32),
                                  torch.tensor([0.0, 0.0, 0.0]))
        if rng.integers(2):
            sc.manual_exposure = float(rng.uniform(0.3, 2.0))
        batch.append(sc)
    mask = [_abi.OUT_ALL, _abi.OUT_GT6, _abi.OUT_INSTANCE | _abi.OUT_COORD, _abi.OUT_RGB | _abi.OUT_VERTEX_IDX | _abi.OUT_BARY][int(rng.integers(4))]
    ssao, shadows = bool(rng.integers(2)), bool(rng.integers(2))
    try:
        bufs, ref = T.both(eng, oracle, batch, mask=mask, ssao=ssao, shadows=shadows)
        T.assert_geometry_equal(bufs, ref, mask=mask)
        if mask & _abi.OUT_RGB:
            T.assert_rgb_close(bufs, ref)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed %d size %s mask 0x%x ssao %d shadows %d: %s" % (seed, size, mask, ssao, shadows, str(e)[:200]))
print("%d cases: %s (%.0f s)" % (N, "all within the bar" if bad == 0 else "%d differ" % bad, time.time() - t0))
sys.exit(1 if bad else 0)
