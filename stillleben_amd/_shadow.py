"""Shadow-map matrices: computeFrustumCorners + computeShadowMapMatrix
(reference src/render_pass.cpp:69-211), in float32."""
import numpy as np

from . import _math as M
from ._math import f32

INF = f32(np.inf)


def _proj_point(P, p):
    v = P @ np.array([p[0], p[1], p[2], 1.0], dtype=np.float32)
    return (v[:3] / v[3]).astype(np.float32)


def frustum_corners(scene):  # render_pass.cpp:69-129
    P = scene._projection
    Pinv = np.linalg.inv(P.astype(np.float64)).astype(np.float32)
    cam_matrix = M.inverted_rigid(scene._camera_pose)
    near, far = f32(-1.0), f32(1.0)
    if scene._objects:
        near_obj, far_obj = INF, -INF
        for obj in scene._objects:
            bbox = obj._mesh.bbox
            c = M.transform_point((cam_matrix @ obj._pose).astype(np.float32), bbox.np_center())
            radius = bbox.np_diagonal() / f32(2.0)
            near_pt = _proj_point(P, c - np.array([0, 0, radius], np.float32))
            far_pt = _proj_point(P, c + np.array([0, 0, radius], np.float32))
            near_obj = min(near_obj, near_pt[2])
            far_obj = max(far_obj, far_pt[2])
        near = max(max(f32(-1.0), near_obj), near)
        far = min(far_obj, far)
    h = np.array([
        [-1, 1, near, 1], [1, 1, near, 1], [1, -1, near, 1], [-1, -1, near, 1],
        [-1, 1, far, 1], [1, 1, far, 1], [1, -1, far, 1], [-1, -1, far, 1],
    ], dtype=np.float32)
    cam_to_world = M.inverted_rigid(cam_matrix)
    corners = np.zeros((8, 3), np.float32)
    for i in range(8):
        p = cam_to_world @ (Pinv @ h[i])
        corners[i] = p[:3] / p[3]
    return corners


def shadow_matrix(scene, corners, light_direction):  # render_pass.cpp:131-211
    z = M.normalized(light_direction)
    x = M.normalized(np.cross(z, np.array([0, 0, 1], np.float32)).astype(np.float32))
    y = M.normalized(np.cross(z, x).astype(np.float32))
    cam_to_world = M.from_rt(np.stack([x, y, z], axis=1), np.zeros(3, np.float32))
    world_to_cam = M.inverted_rigid(cam_to_world)
    pts = np.stack([M.transform_point(world_to_cam, c) for c in corners])
    mn, mx = pts.min(axis=0), pts.max(axis=0)
    near, far = mn[2], mx[2]
    mean_z = (near + far) / f32(2.0)
    spread = far - mean_z
    far = mean_z + f32(5.0) * spread
    near = mean_z - f32(5.0) * spread
    L, R, T, B = mn[0], mx[0], mn[1], mx[1]
    if scene._objects:
        max_obj = np.full(3, INF, np.float32)   # (sic) naming follows the reference
        min_obj = np.full(3, -INF, np.float32)
        for obj in scene._objects:
            bbox = obj._mesh.bbox
            radius = bbox.np_diagonal() / f32(2.0)
            c = M.transform_point((world_to_cam @ obj._pose).astype(np.float32), bbox.np_center())
            max_obj = np.minimum(max_obj, c - radius)
            min_obj = np.maximum(min_obj, c + radius)
        L = max(L, max_obj[0]); R = min(R, min_obj[0])
        T = max(T, max_obj[1]); B = min(B, min_obj[1])
    P = np.array([
        [f32(2.0) / (R - L), 0, 0, -(R + L) / (R - L)],
        [0, f32(2.0) / (B - T), 0, -(B + T) / (B - T)],
        [0, 0, f32(2.0) / (far - near), -(far + near) / (far - near)],
        [0, 0, 0, 1],
    ], dtype=np.float32)
    return (P @ world_to_cam).astype(np.float32)


def shadow_matrices(scene):
    """The shadow matrices the renderer uses: computed by the C++ host layer (csrc/slhip_records.cpp, the same functions the
    batch path runs: one source of the bits).  frustum_corners / shadow_matrix above are the float32 numpy statement of the same
    reference code, kept as the readable mirror and compared in tests/test_host_records.py."""
    from . import _host_records

    return _host_records.shadow_matrices(scene)


def shadow_matrices_numpy(scene):
    from ._batch import effective_lights

    ld, lc, _ = effective_lights(scene)
    mats = [np.eye(4, dtype=np.float32) for _ in range(ld.shape[0])]
    corners = None
    for i in range(ld.shape[0]):
        if not lc[i].any() or not ld[i].any():
            continue
        if corners is None:
            corners = frustum_corners(scene)
        with np.errstate(all="ignore"):
            m = shadow_matrix(scene, corners, ld[i])
        if np.all(np.isfinite(m)):
            mats[i] = m
    return mats
