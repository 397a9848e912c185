#!/usr/bin/env python3
"""Developer tool: randomized parity fuzz of the material / lighting inputs of the fragment stage -- random subsets
of the five material textures on UV spheres, stickers, image-based lighting on / off, background images, random
factors and exposure, depth peeling of the result; GPU vs oracle, geometry bit for bit, rgb to the tests' bar."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import stillleben_amd as sl  # noqa: E402
from stillleben_amd import _abi, _engine  # noqa: E402
from stillleben_amd._batch import HostPool, build_batch  # noqa: E402
from stillleben_amd._context import engine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sl.init()
import test_gpu_materials as TM  # noqa: E402
import test_gpu_render as T  # noqa: E402
from test_gpu_ibl import sky  # noqa: E402

eng = engine()
kinds = ["base", "normal", "mr", "occlusion", "emissive"]
# a fixed library of meshes (every sl.Mesh registers its data in the pool for good)
rng0 = np.random.default_rng(12345)
lib = []
for i in range(10):
    sub = tuple(k for k in kinds if rng0.random() < 0.5)
    d = TM.textured_sphere(100 + i, n_lat=int(rng0.integers(8, 20)), n_lon=int(rng0.integers(12, 32)), with_tex=sub)
    d.materials[0].metallic, d.materials[0].roughness = float(rng0.random()), float(rng0.uniform(0.05, 1))
    d.materials[0].emissive[:] = rng0.uniform(0, 1.5, 3)
    lib.append(sl.Mesh.from_data(d))
sizes = dict(env_size=32, env_levels=6, irr_size=4, pre_size=16, pre_levels=5, lut_size=16)
lms = []
for i in range(2):
    lm = sl.LightMap(sky(32, 64, seed=40 + i), sizes=sizes)
    lm.light_directions = [np.array([0.2, 0.3 - 0.5 * i, -0.9], np.float32)]
    lm.light_colors = [np.array([2.0, 1.5 + i, 2.0], np.float32)]
    lms.append(lm)
bad = 0
t0 = time.time()
for k in range(N):
    seed = BASE + k
    rng = np.random.default_rng(seed)
    size = [(240, 180), (160, 120), (199, 101)][int(rng.integers(3))]
    meshes = [lib[int(rng.integers(len(lib)))] for _ in range(int(rng.integers(1, 5)))]
    scene = TM.make_scene(sl, meshes, seed, size=size)
    scene.manual_exposure = float(rng.uniform(0.2, 1.5)) if rng.integers(3) else -1.0
    if rng.integers(2):
        o = scene.objects[int(rng.integers(len(scene.objects)))]
        st = rng.integers(0, 256, (int(rng.integers(4, 40)), int(rng.integers(4, 40)), 4)).astype(np.uint8)
        st[: st.shape[0] // 3, :, 3] = 0
        o.sticker_texture = sl.Texture2D(torch.from_numpy(st))
        o.sticker_range = [-0.5, -0.4, 0.5, 0.4]
        q = rng.standard_normal(4)
        o.sticker_rotation = list((q / np.linalg.norm(q)).astype(np.float32))
    host_maps = None
    if rng.integers(2):
        lm = lms[int(rng.integers(2))]
        scene.light_map = lm
        host_maps = [None] * (max(m._slot for m in lms) + 1)
        for m in lms:
            host_maps[m._slot] = ({"env": m.env.cpu().numpy(), "irradiance": m.irradiance.cpu().numpy(),
                                   "prefilter": m.prefilter.cpu().numpy(), "brdf_lut": m.brdf_lut.cpu().numpy()}, sizes)
        host_maps = [h if h is not None else host_maps[lms[0]._slot] for h in host_maps]
    if rng.integers(3) == 0:
        bg = rng.integers(0, 256, (int(rng.integers(8, 64)), int(rng.integers(8, 64)), 3)).astype(np.uint8)
        scene.background_image = sl.Texture(torch.from_numpy(bg))
    ssao, shadows = bool(rng.integers(2)), bool(rng.integers(2))
    try:
        W, H = scene.viewport
        flags = _abi.OUT_ALL | (_abi.RENDER_SSAO if ssao else 0) | (_abi.RENDER_SHADOWS if shadows else 0)
        bufs = eng.render([scene], _abi.OUT_ALL, ssao=ssao, shadows=shadows)
        torch.cuda.synchronize()
        pool = HostPool()
        srec, drec, _ = build_batch([scene], pool, with_shadows=shadows)
        ref = oracle.render(pool.arrays(), srec, drec, W, H, flags, shadow_res=_engine.SHADOW_RES, light_maps=host_maps)
        T.assert_geometry_equal(bufs, ref)
        T.assert_rgb_close(bufs, ref)
        # second layer: depth peel behind the first result
        peel = ref.coord.copy()
        bufs2 = eng.render([scene], _abi.OUT_ALL, ssao=ssao, shadows=shadows, depth_peel=torch.from_numpy(peel).to(eng.device))
        torch.cuda.synchronize()
        ref2 = oracle.render(pool.arrays(), srec, drec, W, H, flags, depth_peel=peel, shadow_res=_engine.SHADOW_RES, light_maps=host_maps)
        T.assert_geometry_equal(bufs2, ref2)
        T.assert_rgb_close(bufs2, ref2)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed %d size %s ssao %d shadows %d ibl %d: %s" % (seed, size, ssao, shadows, host_maps is not None, str(e)[:200]))
print("%d cases: %s (%.0f s)" % (N, "all within the bar" if bad == 0 else "%d differ" % bad, time.time() - t0))
sys.exit(1 if bad else 0)
