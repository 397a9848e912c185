#!/usr/bin/env python3
"""Developer tool: randomized parity fuzz of the post-render operators on freshly rendered random scenes --
camera model (bit-exact vs the oracle), sl.diff stencils / image gradients (bit-exact), pose backward (1e-5 rel.
vs the fp64 oracle) and vertex backward (bit-exact)."""
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
from stillleben_amd import camera_model as cm  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sl.init()
CUBE = sl.Mesh(S.CUBE, physics=False)
CUBE.center_bbox()
CUBE.scale_to_bbox_diagonal(0.25)
BUNNY = sl.Mesh(S.BUNNY, physics=False)
BUNNY.center_bbox()
BUNNY.scale_to_bbox_diagonal(0.3)
sizes = [(160, 120), (320, 240), (200, 150), (96, 64)]
bad = 0
t0 = time.time()
for k in range(N):
    seed = BASE + k
    rng = np.random.default_rng(seed)
    W, H = sizes[int(rng.integers(len(sizes)))]
    scene = sl.Scene((W, H), seed=seed)
    for i in range(int(rng.integers(1, 9))):
        o = sl.Object(BUNNY if rng.random() < 0.3 else CUBE)
        p = np.eye(4, dtype=np.float32)
        p[:3, :3] = S.random_rotation(rng)
        p[:3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.0, 0.3)]
        o.set_pose(torch.from_numpy(p))
        scene.add_object(o)
    az = rng.uniform(-math.pi, math.pi)
    scene.set_camera_look_at(torch.tensor([1.1 * math.cos(az), 1.1 * math.sin(az), rng.uniform(0.3, 1.0)], dtype=torch.float32),
                             torch.tensor([0.0, 0.0, 0.1]))
    scene.choose_random_light_direction()
    scene.ambient_light = torch.tensor([0.2, 0.2, 0.2])
    res = sl.RenderPass().render(scene)
    errs = []
    # ---- camera model ----
    rgb = res.rgb()[:, :, :3].permute(2, 0, 1).float().div(255.0).contiguous()
    p = cm.make_params(rng.uniform(-0.003, 0.003, (3, 2)), rng.uniform(0.997, 1.003, 3), float(rng.choice([0.0, 0.7, 2.5])),
                       float(rng.uniform(-2, 1.2)), False, 0.0, 0.0, float(rng.uniform(-0.05, 0.05)), seed=0)
    out = cm.process_batch(rgb[None].cuda(), [p])[0].cpu().numpy()
    ref = oracle.camera_model(rgb.numpy()[None], [p])[0]
    if not np.array_equal(out.view(np.uint32), ref.view(np.uint32)):
        errs.append("camera model max %g" % np.abs(out - ref).max())
    # ---- sl.diff ----
    inst = res.instance_index().cpu().numpy().reshape(H, W)
    coord = res.coordDepth().cpu().numpy()
    valid = sl.diff.generate_sobel_valid_mask(torch.from_numpy(inst), torch.from_numpy(coord[:, :, 3].copy()))
    if not np.array_equal(valid.numpy(), oracle.sobel_valid(inst, coord[:, :, 3])):
        errs.append("sobel valid")
    for o in scene.objects[:2]:
        m, c3 = sl.diff.dilate_object_mask(torch.from_numpy(inst == o.instance_index), valid, torch.from_numpy(coord[:, :, :3].copy()))
        om, oc = oracle.dilate(inst == o.instance_index, valid.numpy(), coord[:, :, :3])
        if not (np.array_equal(m.numpy(), om) and np.array_equal(c3.numpy().view(np.uint32), oc.view(np.uint32))):
            errs.append("dilate")
    gx, gy, v2 = sl.diff.compute_image_space_gradients(scene, res)
    rgb8 = res.rgb().cpu().numpy()
    ogx, ogy = oracle.image_gradients(rgb8, v2.cpu().numpy())
    if not (np.array_equal(gx.cpu().numpy(), ogx) and np.array_equal(gy.cpu().numpy(), ogy)):
        errs.append("image gradients")
    g_img = torch.from_numpy(rng.standard_normal((3, H, W)).astype(np.float32))
    g = sl.diff.backpropagate_gradient_to_poses(scene, res, g_img).cpu().numpy()
    P = scene.projection_matrix().cpu().numpy()
    poses = np.stack([o.pose().cpu().numpy() for o in scene.objects])
    oinst = np.array([o.instance_index for o in scene.objects], np.int32)
    orc = oracle.pose_backward(rgb8, coord, inst, g_img.numpy(), P, poses, oinst)
    if np.abs(g - orc).max() > 1e-5 * max(1e-6, np.abs(orc).max()):
        errs.append("pose backward %g of %g" % (np.abs(g - orc).max(), np.abs(orc).max()))
    vi, gv, gc = sl.diff.bp_to_vertices_and_colors(scene, res, g_img)
    bary = res.barycentric_coeffs().cpu().numpy()
    ogv, ogc = oracle.vertex_backward(rgb8, coord, inst, bary, g_img.numpy(), P, poses, oinst)
    gv_c, gc_c = torch.cat(gv).cpu().numpy(), torch.cat(gc).cpu().numpy()
    rgv = np.concatenate([ogv[inst == o].reshape(-1, 3) for o in oinst])
    rgc = np.concatenate([ogc[inst == o].reshape(-1, 3) for o in oinst])
    if not (np.array_equal(gv_c.view(np.uint32), rgv.view(np.uint32)) and np.array_equal(gc_c.view(np.uint32), rgc.view(np.uint32))):
        errs.append("vertex backward")
    if errs:
        bad += 1
        print("MISMATCH seed %d %dx%d: %s" % (seed, W, H, "; ".join(errs)))
print("%d cases: %s (%.0f s)" % (N, "all within the bar" if bad == 0 else "%d differ" % bad, time.time() - t0))
sys.exit(1 if bad else 0)
