"""Seeded scene builders shared by the CPU (oracle) tests and the GPU parity tests."""
import math
import os

import numpy as np
import torch

FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures")
CUBE = os.path.join(FIXTURES, "cube.glb")
BUNNY = os.path.join(FIXTURES, "stanford_bunny", "scene.gltf")

_mesh_cache = {}


def mesh(sl, path, physics=False):
    key = (path, physics)
    if key not in _mesh_cache:
        _mesh_cache[key] = sl.Mesh(path, physics=physics)
    return _mesh_cache[key]


def cube_lookat_scene(sl, size=(640, 480)):
    """reference tests/basic.cpp:375-453: cube seen from (4,0,0) looking at the origin."""
    m = sl.Mesh(CUBE, physics=False)
    scene = sl.Scene(size)
    obj = sl.Object(m)
    scene.add_object(obj)
    scene.set_camera_look_at(torch.tensor([4.0, 0.0, 0.0]), torch.tensor([0.0, 0.0, 0.0]))
    return scene


def random_rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ], dtype=np.float32)


def clutter_scene(sl, seed, n_objects=6, size=(320, 240), with_bunny=False, plane=True, light=True):
    """Random heap of cubes (and optionally a bunny) on a plane, camera looking down at it."""
    rng = np.random.default_rng(seed)
    scene = sl.Scene(size, seed=seed)
    cube = sl.Mesh(CUBE, physics=False)
    cube.center_bbox()
    cube.scale_to_bbox_diagonal(0.2)
    meshes = [cube]
    if with_bunny:
        b = sl.Mesh(BUNNY, physics=False)
        b.center_bbox()
        b.scale_to_bbox_diagonal(0.3)
        b.class_index = 2
        meshes.append(b)
    for i in range(n_objects):
        m = meshes[i % len(meshes)]
        obj = sl.Object(m)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = random_rotation(rng)
        pose[:3, 3] = [rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(0.05, 0.3)]
        obj.set_pose(torch.from_numpy(pose))
        obj.metallic = float(rng.uniform(0, 1))
        obj.roughness = float(rng.uniform(0, 1))
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


def delaunay_sheet(sl, n_points, seed, reverse=False, half=1.0):
    """A watertight flat sheet in the mesh's z = 0 plane: `n_points` random interior points plus a dense border, Delaunay
    triangulated (every interior edge is shared by exactly two triangles).  `reverse` keeps the geometry and submits the
    triangles in the opposite order -- on a fronto-parallel sheet every triangle has the SAME depth, so the depth test
    (LESS, earlier triangle wins) makes any pixel that two triangles both claim change its owner."""
    from scipy.spatial import Delaunay

    from stillleben_amd import _loaders

    rng = np.random.default_rng(seed)
    nb = max(8, int(math.sqrt(n_points)))
    t = np.linspace(-half, half, nb, endpoint=False)
    border = np.concatenate([np.stack([t, np.full(nb, -half)], 1), np.stack([np.full(nb, half), t], 1),
                             np.stack([-t, np.full(nb, half)], 1), np.stack([np.full(nb, -half), -t], 1)])
    pts = np.concatenate([border, rng.uniform(-half, half, (n_points, 2))]).astype(np.float32)
    tri = Delaunay(pts.astype(np.float64)).simplices.astype(np.uint32)
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    cw = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])) < 0
    tri[cw] = tri[cw][:, ::-1]                      # counter-clockwise seen from +z
    if reverse:
        tri = tri[::-1]
    cm = _loaders.ConsolidatedMesh()
    cm.positions = np.concatenate([pts, np.zeros((len(pts), 1), np.float32)], 1)
    cm.normals = np.tile(np.array([[0, 0, 1]], np.float32), (len(pts), 1))
    cm.uvs = (pts * 0.5 + 0.5).astype(np.float32)
    cm.colors = np.ones((len(pts), 4), np.float32)
    cm.indices = np.ascontiguousarray(tri).reshape(-1).astype(np.uint32)
    cm.textures = []
    cm._tex_alpha = []
    cm.materials = [_loaders.Material(base_color=(0.8, 0.8, 0.8, 1))]
    cm.submeshes = [_loaders.SubMesh(0, len(cm.indices), 0)]
    return sl.Mesh.from_data(cm, hulls=[], filename="memory://sheet%d" % seed)


def sheet_scene(sl, mesh, roll_deg=17.0, tilt_deg=0.0, distance=4.0, size=(320, 240)):
    """The sheet in front of the camera (camera at the origin looking along its optical axis): rolled about the optical
    axis so that no edge is pixel-aligned, optionally tilted (then depths differ across the sheet)."""
    scene = sl.Scene(size)
    obj = sl.Object(mesh)
    r, t = math.radians(roll_deg), math.radians(tilt_deg)
    Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]])
    Ry = np.array([[math.cos(t), 0, math.sin(t)], [0, 1, 0], [-math.sin(t), 0, math.cos(t)]])
    cam = scene.camera_pose().numpy().astype(np.float64)          # camera-to-world; the camera looks along its +z
    P = np.eye(4)
    P[:3, :3] = Ry @ Rz @ np.diag([1.0, -1.0, -1.0])              # sheet normal towards the camera
    P[:3, 3] = [0.0, 0.0, distance]
    obj.set_pose(torch.from_numpy((cam @ P).astype(np.float32)))
    scene.add_object(obj)
    return scene
