"""Vectorised (numpy, whole batch at once) version of the per-scene host work that sits
between the settle and the render of a batch: camera placement
(Scene::chooseRandomCameraPose, scene.cpp:472-610), random light direction (scene.cpp:453-470),
shadow matrices (render_pass.cpp:69-211) and the pose-dependent fields of the draw records.
Everything pose-independent (materials, mesh ranges, chunks) is prepared once per batch by
`prepare`.  tests/test_fast_batch.py checks it against the per-scene code paths."""
import numpy as np

from . import _math as M
from ._batch import build_batch
from .camera_placement import camera_rotation

f32 = np.float32


class BatchTemplate:
    pass


def prepare(scenes, pool):
    """Static part of a render batch; scenes must share projection & viewport handling rules."""
    t = BatchTemplate()
    srec, drec, crec = build_batch(scenes, pool, with_shadows=False)
    t.srec, t.drec, t.crec = srec, drec, crec
    t.n_scenes = len(scenes)
    # map draws -> (scene, object slot); the plane draw has object -1
    draw_scene, draw_obj = [], []
    obj_base = []
    k = 0
    for si, s in enumerate(scenes):
        obj_base.append(k)
        has_plane = float(np.dot(s._background_plane_size, s._background_plane_size)) > 0
        if has_plane:
            draw_scene.append(si)
            draw_obj.append(-1)
        for oi, o in enumerate(s._objects):
            for _ in o._mesh._data.submeshes:
                draw_scene.append(si)
                draw_obj.append(k + oi)
        k += len(s._objects)
    t.draw_scene = np.array(draw_scene, np.int64)
    t.draw_obj = np.array(draw_obj, np.int64)
    t.obj_scene = np.concatenate([np.full(len(s._objects), si, np.int64) for si, s in enumerate(scenes)])
    t.obj_base = np.array(obj_base + [k], np.int64)
    t.n_obj = k
    objs = [o for s in scenes for o in s._objects]
    t.m2o = np.stack([o._mesh._pretransform for o in objs]).astype(np.float32)
    t.bbox_corners = np.stack([o._mesh.bbox.corners() for o in objs]).astype(np.float32)       # [O,8,3]
    t.bbox_center = np.stack([o._mesh.bbox.np_center() for o in objs]).astype(np.float32)       # [O,3]
    t.bbox_radius = np.array([o._mesh.bbox.np_diagonal() / f32(2.0) for o in objs], np.float32)  # [O]
    t.proj = np.stack([s._projection for s in scenes]).astype(np.float32)
    t.proj_inv = np.linalg.inv(t.proj.astype(np.float64)).astype(np.float32)
    t.plane_size = np.stack([s._background_plane_size for s in scenes]).astype(np.float32)
    if any(s._light_map is not None for s in scenes):
        raise NotImplementedError("the vectorised batch path draws random light directions; scenes with a light map go "
                                  "through _batch.build_batch")
    t.light_colors = np.stack([s._light_colors.numpy() for s in scenes]).astype(np.float32)
    t.max_objs = max(len(s._objects) for s in scenes)
    return t


def replicate(t1, B):
    """A B-scene template whose scenes are all copies of the ONE scene of template `t1` (same objects, materials, camera
    intrinsics; poses, camera pose and lights are filled in by `update`): the records are tiled with numpy instead of being
    rebuilt scene by scene -- used for batches of pose hypotheses of one scene (sl.diff)."""
    if t1.n_scenes != 1:
        raise ValueError("replicate() needs a single-scene template")
    t = BatchTemplate()
    nd, nc, no = len(t1.drec), len(t1.crec), t1.n_obj
    t.n_scenes, t.n_obj, t.max_objs = B, no * B, t1.max_objs
    t.srec = np.tile(t1.srec, B)
    t.srec["draw_begin"] = np.arange(B, dtype=np.uint32) * nd
    t.srec["draw_end"] = t.srec["draw_begin"] + nd
    t.drec = np.tile(t1.drec, B)
    t.drec["scene"] = np.repeat(np.arange(B, dtype=np.uint32), nd)
    per_scene_clip = int(t1.drec["n_verts"].sum())
    t.drec["clip_base"] = (np.tile(t1.drec["clip_base"].astype(np.uint64), B) + np.repeat(np.arange(B, dtype=np.uint64) * per_scene_clip, nd)).astype(np.uint32)
    t.crec = np.tile(t1.crec, B)
    t.crec["scene"] = np.repeat(np.arange(B, dtype=np.uint32), nc)
    t.crec["draw"] = np.tile(t1.crec["draw"], B) + np.repeat(np.arange(B, dtype=np.uint32) * nd, nc)
    t.draw_scene = np.repeat(np.arange(B, dtype=np.int64), nd)
    obj = np.tile(t1.draw_obj, B)
    shift = np.repeat(np.arange(B, dtype=np.int64) * no, nd)
    t.draw_obj = np.where(obj >= 0, obj + shift, -1)
    t.obj_scene = np.repeat(np.arange(B, dtype=np.int64), no)
    t.obj_base = np.arange(B + 1, dtype=np.int64) * no
    t.m2o = np.tile(t1.m2o, (B, 1, 1))
    t.bbox_corners = np.tile(t1.bbox_corners, (B, 1, 1))
    t.bbox_center = np.tile(t1.bbox_center, (B, 1))
    t.bbox_radius = np.tile(t1.bbox_radius, B)
    t.proj = np.tile(t1.proj, (B, 1, 1))
    t.proj_inv = np.tile(t1.proj_inv, (B, 1, 1))
    t.plane_size = np.tile(t1.plane_size, (B, 1))
    t.light_colors = np.tile(t1.light_colors, (B, 1, 1))
    return t


def _pad_by_scene(t, arr, fill):
    """[O,...] -> [B,maxN,...] with `fill` in the unused slots."""
    out = np.full((t.n_scenes, t.max_objs) + arr.shape[1:], fill, dtype=arr.dtype)
    idx = np.arange(t.n_obj) - t.obj_base[t.obj_scene]
    out[t.obj_scene, idx] = arr
    return out


def camera_poses(t, poses, azimuth, elevation):
    """poses [O,4,4] -> camera poses [B,4,4] (scene.cpp:472-610, vectorised)."""
    B = t.n_scenes
    cam_rot = np.stack([camera_rotation(f32(a), f32(e)) for a, e in zip(azimuth, elevation)])        # [B,4,4]
    to_work = np.stack([M.inverted_rigid(c) for c in cam_rot])                                       # [B,4,4]
    trans = np.einsum("oij,ojk->oik", to_work[t.obj_scene], poses).astype(np.float32)                # [O,4,4]
    pts = (np.einsum("oij,ocj->oci", trans[:, :3, :3], t.bbox_corners) + trans[:, None, :3, 3]).astype(np.float32)
    P = t.proj
    fr = np.stack([P[:, 3] + P[:, 0], P[:, 3] - P[:, 0], P[:, 3] + P[:, 1], P[:, 3] - P[:, 1]], axis=1).astype(np.float32)
    ln = np.sqrt(np.einsum("bkj,bkj->bk", fr[:, :, :3], fr[:, :, :3])).astype(np.float32)
    fr = (fr / ln[:, :, None]).astype(np.float32)
    dots = np.einsum("ocj,okj->ock", pts, fr[t.obj_scene][:, :, :3]).astype(np.float32)             # [O,8,4]
    mins = _pad_by_scene(t, dots.min(axis=1), np.float32(np.inf)).min(axis=1)                        # [B,4]
    fr[:, :, 3] = -mins

    def intersect(a, b, ia):
        la = np.stack([a[:, ia], a[:, 2], a[:, 3]], axis=1)
        lb = np.stack([b[:, ia], b[:, 2], b[:, 3]], axis=1)
        x = np.cross(la, lb).astype(np.float32)
        bad = np.abs(x[:, 2]) < 1e-3
        x[bad] = (0.0, 0.0, 1.0)
        return x[:, 0] / x[:, 2], x[:, 1] / x[:, 2]

    lr_x, lr_z = intersect(fr[:, 0], fr[:, 1], 0)
    tb_y, tb_z = intersect(fr[:, 2], fr[:, 3], 1)
    cam_pos = np.stack([lr_x, tb_y, np.minimum(lr_z, tb_z)], axis=1).astype(np.float32)
    tr = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    tr[:, :3, 3] = cam_pos
    return np.einsum("bij,bjk->bik", cam_rot, tr).astype(np.float32)


def light_directions(cam_pose, normals3):
    """scene.cpp:453-470 given the three N(0,1) draws per scene: [B,3] -> world dirs [B,3]."""
    d = np.stack([normals3[:, 0], -np.abs(normals3[:, 1]), -np.abs(normals3[:, 2])], axis=1).astype(np.float32)
    d = d / np.sqrt((d * d).sum(axis=1, keepdims=True)).astype(np.float32)
    d = d / np.sqrt((d * d).sum(axis=1, keepdims=True)).astype(np.float32)
    return np.einsum("bij,bj->bi", cam_pose[:, :3, :3], -d).astype(np.float32)


def _tp(m, p):
    """transformPoint for batches: m [...,4,4], p [...,3]."""
    q = np.einsum("...ij,...j->...i", m[..., :3, :3], p) + m[..., :3, 3]
    w = np.einsum("...j,...j->...", m[..., 3, :3], p) + m[..., 3, 3]
    return (q / w[..., None]).astype(np.float32)


def shadow_matrices(t, poses, cam_pose, light_dir):
    """render_pass.cpp:69-211 for ONE light per scene (light index 0): [B,4,4]."""
    B = t.n_scenes
    P, Pinv = t.proj, t.proj_inv
    w2c = np.stack([M.inverted_rigid(c) for c in cam_pose])
    # frustum corners
    obj_in_cam = _tp(np.einsum("oij,ojk->oik", w2c[t.obj_scene], poses).astype(np.float32), t.bbox_center)
    near_pt = obj_in_cam - np.stack([np.zeros_like(t.bbox_radius), np.zeros_like(t.bbox_radius), t.bbox_radius], axis=1)
    far_pt = obj_in_cam + np.stack([np.zeros_like(t.bbox_radius), np.zeros_like(t.bbox_radius), t.bbox_radius], axis=1)
    near_z = _tp(P[t.obj_scene], near_pt)[:, 2]
    far_z = _tp(P[t.obj_scene], far_pt)[:, 2]
    near_obj = _pad_by_scene(t, near_z, np.float32(np.inf)).min(axis=1)
    far_obj = _pad_by_scene(t, far_z, np.float32(-np.inf)).max(axis=1)
    near = np.maximum(np.maximum(f32(-1.0), near_obj), f32(-1.0)).astype(np.float32)
    far = np.minimum(far_obj, f32(1.0)).astype(np.float32)
    sx = np.array([-1, 1, 1, -1, -1, 1, 1, -1], np.float32)
    sy = np.array([1, 1, -1, -1, 1, 1, -1, -1], np.float32)
    h = np.zeros((B, 8, 4), np.float32)
    h[:, :, 0], h[:, :, 1], h[:, :, 3] = sx, sy, 1.0
    h[:, :4, 2] = near[:, None]
    h[:, 4:, 2] = far[:, None]
    c2w = np.stack([M.inverted_rigid(w) for w in w2c])
    p = np.einsum("bij,bcj->bci", Pinv, h).astype(np.float32)
    p = np.einsum("bij,bcj->bci", c2w, p).astype(np.float32)
    corners = (p[:, :, :3] / p[:, :, 3:4]).astype(np.float32)
    # light frame
    z = light_dir / np.sqrt((light_dir * light_dir).sum(axis=1, keepdims=True)).astype(np.float32)
    x = np.cross(z, np.array([0, 0, 1], np.float32)).astype(np.float32)
    x = x / np.sqrt((x * x).sum(axis=1, keepdims=True)).astype(np.float32)
    y = np.cross(z, x).astype(np.float32)
    y = y / np.sqrt((y * y).sum(axis=1, keepdims=True)).astype(np.float32)
    l2w = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    l2w[:, :3, 0], l2w[:, :3, 1], l2w[:, :3, 2] = x, y, z
    w2l = np.stack([M.inverted_rigid(m) for m in l2w])
    cl = (np.einsum("bij,bcj->bci", w2l[:, :3, :3], corners) + w2l[:, None, :3, 3]).astype(np.float32)
    mn, mx = cl.min(axis=1), cl.max(axis=1)
    near_l, far_l = mn[:, 2], mx[:, 2]
    mean_z = (near_l + far_l) / f32(2.0)
    spread = far_l - mean_z
    far_l = mean_z + f32(5.0) * spread
    near_l = mean_z - f32(5.0) * spread
    L, R, T, Bm = mn[:, 0], mx[:, 0], mn[:, 1], mx[:, 1]
    cen = _tp(np.einsum("oij,ojk->oik", w2l[t.obj_scene], poses).astype(np.float32), t.bbox_center)
    lo = _pad_by_scene(t, cen - t.bbox_radius[:, None], np.float32(np.inf)).min(axis=1)
    hi = _pad_by_scene(t, cen + t.bbox_radius[:, None], np.float32(-np.inf)).max(axis=1)
    L = np.maximum(L, lo[:, 0]); R = np.minimum(R, hi[:, 0])
    T = np.maximum(T, lo[:, 1]); Bm = np.minimum(Bm, hi[:, 1])
    Pm = np.zeros((B, 4, 4), np.float32)
    Pm[:, 0, 0] = f32(2.0) / (R - L); Pm[:, 0, 3] = -(R + L) / (R - L)
    Pm[:, 1, 1] = f32(2.0) / (Bm - T); Pm[:, 1, 3] = -(Bm + T) / (Bm - T)
    Pm[:, 2, 2] = f32(2.0) / (far_l - near_l); Pm[:, 2, 3] = -(far_l + near_l) / (far_l - near_l)
    Pm[:, 3, 3] = 1.0
    return np.einsum("bij,bjk->bik", Pm, w2l).astype(np.float32)


def update(t, poses, cam_pose, light_dir, plane_pose, with_shadows=True):
    """Fills the pose-dependent fields of the prepared records in place; returns (srec, drec)."""
    srec, drec = t.srec, t.drec
    w2c = np.stack([M.inverted_rigid(c) for c in cam_pose])
    srec["world_to_cam"] = w2c.reshape(-1, 16)
    srec["cam_position"][:, :3] = cam_pose[:, :3, 3]
    srec["cam_position"][:, 3] = 1.0
    srec["light_dir"][:, 0, :3] = light_dir
    if with_shadows:
        sm = shadow_matrices(t, poses, cam_pose, light_dir)
        ok = np.isfinite(sm).all(axis=(1, 2))
        sm[~ok] = np.eye(4, dtype=np.float32)
        srec["shadow_mat"][:, 0] = sm.reshape(-1, 16)
    is_obj = t.draw_obj >= 0
    o2w = np.empty((len(drec), 4, 4), np.float32)
    o2w[is_obj] = poses[t.draw_obj[is_obj]]
    if (~is_obj).any():
        sc = t.draw_scene[~is_obj]
        scal = np.tile(np.eye(4, dtype=np.float32), (len(sc), 1, 1))
        scal[:, 0, 0] = t.plane_size[sc, 0] / f32(2.0)
        scal[:, 1, 1] = t.plane_size[sc, 1] / f32(2.0)
        o2w[~is_obj] = np.einsum("bij,bjk->bik", plane_pose[sc], scal).astype(np.float32)
    drec["object_to_world"] = o2w.reshape(-1, 16)
    m2w = o2w.copy()
    m2w[is_obj] = np.einsum("bij,bjk->bik", o2w[is_obj], t.m2o[t.draw_obj[is_obj]]).astype(np.float32)
    nm = np.zeros((len(drec), 3, 4), np.float32)
    nm[:, :, :3] = np.transpose(np.linalg.inv(m2w[:, :3, :3].astype(np.float64)), (0, 2, 1)).astype(np.float32)
    drec["normal_to_world"] = nm.reshape(-1, 12)
    return srec, drec
