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
