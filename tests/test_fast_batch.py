"""The vectorised batch assembly (_fast_batch) must produce the same records as the per-scene
code paths (Scene.choose_random_camera_pose, _shadow.shadow_matrices, _batch.build_batch)."""
import math

import numpy as np
import torch

import scenes as S
from stillleben_amd import _fast_batch as FB
from stillleben_amd import camera_placement
from stillleben_amd._batch import HostPool, build_batch


def test_fast_batch_matches_per_scene(sl):
    scs = [S.clutter_scene(sl, 40 + i, n_objects=3 + i, size=(160, 120), with_bunny=(i % 2 == 0)) for i in range(4)]
    rng = np.random.default_rng(0)
    az = rng.uniform(-math.pi, math.pi, len(scs)).astype(np.float32)
    el = rng.uniform(math.radians(30), math.radians(60), len(scs)).astype(np.float32)
    nrm = rng.standard_normal((len(scs), 3)).astype(np.float32)
    pool = HostPool()
    t = FB.prepare(scs, pool)
    poses = np.stack([o._pose for s in scs for o in s._objects])
    cam = FB.camera_poses(t, poses, az, el)
    ld = FB.light_directions(cam, nrm)
    plane = np.stack([s._background_plane_pose for s in scs])
    srec, drec = FB.update(t, poses, cam, ld, plane)
    srec, drec = srec.copy(), drec.copy()
    # per-scene reference
    for i, s in enumerate(scs):
        s._camera_pose = camera_placement.choose_camera_pose(s, az[i], el[i])
        d = np.array([nrm[i, 0], -abs(nrm[i, 1]), -abs(nrm[i, 2])], np.float32)
        d = d / np.sqrt(np.dot(d, d)).astype(np.float32)
        d = d / np.sqrt(np.dot(d, d)).astype(np.float32)
        s.light_directions = torch.from_numpy((s._camera_pose[:3, :3] @ -d).reshape(1, 3))
    pool2 = HostPool()
    s2, d2, c2 = build_batch(scs, pool2, with_shadows=True)
    assert np.allclose(cam, np.stack([s._camera_pose for s in scs]), atol=2e-5)
    for name in ("proj", "world_to_cam", "cam_position", "light_dir", "light_color", "ambient"):
        assert np.allclose(srec[name], s2[name], atol=3e-5), name
    assert np.allclose(srec["shadow_mat"][:, 0], s2["shadow_mat"][:, 0], rtol=2e-3, atol=2e-3)
    for name in ("draw_begin", "draw_end", "n_prims"):
        assert np.array_equal(srec[name], s2[name])
    for name in ("mesh_to_object", "object_to_world", "normal_to_world", "base_color", "metallic", "roughness"):
        assert np.allclose(drec[name], d2[name], atol=1e-4, rtol=1e-4), name
    for name in ("class_index", "instance_index", "flags", "vtx_base", "idx_base", "n_tris", "prim_base"):
        assert np.array_equal(drec[name], d2[name]), name
    assert np.array_equal(t.crec, c2)
