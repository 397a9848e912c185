"""sl.Scene -- objects + camera + lights (reference include/stillleben/scene.h:69-241,
src/scene.cpp, python/src/py_scene.cpp:48-425)."""
import math
import warnings

import numpy as np
import torch

from . import _math as M
from ._math import as_mat4, as_vec, f32

NUM_LIGHTS = 3


class Scene:
    _phys_state = None   # physics.SceneState: what the reference's PxScene keeps between simulate calls

    def __init__(self, viewport_size, seed=None):
        from ._context import require_context

        require_context()
        self._viewport = (int(viewport_size[0]), int(viewport_size[1]))
        self._objects = []
        self._camera_pose = np.eye(4, dtype=np.float32)
        self._projection = np.eye(4, dtype=np.float32)
        self.set_camera_hfov(math.radians(58.0))  # scene.cpp:138
        # scene.h:225-232: colours {300,0,0}, directions zero => no light until one is set
        self._light_directions = torch.zeros(NUM_LIGHTS, 3)
        self._light_colors = torch.tensor([[300.0, 300.0, 300.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
        self._ambient_light = np.zeros(3, dtype=np.float32)
        self._background_color = np.zeros(4, dtype=np.float32)  # dead parameter (quirk q2)
        self._background_image = None
        self._background_plane_pose = np.eye(4, dtype=np.float32)
        self._background_plane_size = np.zeros(2, dtype=np.float32)
        self._background_plane_texture = None
        self._manual_exposure = f32(-1.0)  # scene.h:240
        self._light_map = None
        self._physics_loaded = False
        # the reference seeds std::mt19937 from random_device (scene.cpp:147-148); an explicit
        # seed is an additive extension
        self._rng = np.random.default_rng(seed)
        self._seed = seed

    # ---- camera ----------------------------------------------------------------------------
    @property
    def viewport(self):
        return self._viewport

    def camera_pose(self):
        return torch.from_numpy(self._camera_pose.copy())

    def set_camera_pose(self, pose):
        m = as_mat4(pose)
        r = m[:3, :3].astype(np.float64)
        if not (np.allclose(r @ r.T, np.eye(3), atol=1e-4) and np.allclose(m[3], [0, 0, 0, 1], atol=1e-6)):
            raise ValueError("Camera pose is not rigid:\n%s" % m)
        self._camera_pose = m

    def set_camera_look_at(self, position, look_at, up=(0.0, 0.0, 1.0)):  # scene.cpp:203-215
        position, look_at, up = as_vec(position, 3), as_vec(look_at, 3), as_vec(up, 3)
        z = M.normalized(look_at - position)
        x = M.normalized(np.cross(z, up).astype(np.float32))
        y = M.normalized(np.cross(z, x).astype(np.float32))
        self.set_camera_pose(M.from_rt(np.stack([x, y, z], axis=1), position))

    def set_camera_intrinsics(self, fx, fy, cx, cy):  # scene.cpp:222-253
        fx, fy, cx, cy = f32(fx), f32(fy), f32(cx), f32(cy)
        f, n = f32(10.0), f32(0.1)
        W, H = f32(self._viewport[0]), f32(self._viewport[1])
        L = -cx * n / fx
        R = (W - cx) * n / fx
        T = -cy * n / fy
        B = (H - cy) * n / fy
        P = np.zeros((4, 4), dtype=np.float32)  # row-major == transpose of Magnum's columns
        P[0, 0] = f32(2.0) * n / (R - L)
        P[1, 1] = f32(2.0) * n / (B - T)
        P[0, 2] = (R + L) / (L - R)
        P[1, 2] = (T + B) / (T - B)
        P[2, 2] = (f + n) / (f - n)
        P[3, 2] = f32(1.0)
        P[2, 3] = (f32(2.0) * f * n) / (n - f)
        self._projection = P

    def set_camera_hfov(self, hfov):  # scene.cpp:260-271
        W, H = f32(self._viewport[0]), f32(self._viewport[1])
        fx = f32(float(W) / (2.0 * math.tan(float(hfov) / 2.0)))
        self.set_camera_intrinsics(fx, fx, W / f32(2.0), H / f32(2.0))

    def set_camera_projection(self, P):
        self._projection = as_mat4(P)

    def projection_matrix(self):
        return torch.from_numpy(self._projection.copy())

    def camera_to_world(self, pose):  # scene.cpp:315-318
        return torch.from_numpy((self._camera_pose @ as_mat4(pose)).astype(np.float32))

    def min_dist_for_object_diameter(self, diameter):
        from . import pose_sampling

        return float(pose_sampling.minimum_distance_for_object_diameter(diameter, self._projection))

    def place_object_randomly(self, diameter, min_size_factor=0.4):
        from . import pose_sampling

        s = pose_sampling.RandomPoseSampler(pose_sampling.RandomPositionSampler(self._projection, diameter, min_size_factor))
        return torch.from_numpy(s(self._rng))

    # ---- objects ---------------------------------------------------------------------------
    @property
    def objects(self):
        return list(self._objects)

    def add_object(self, obj):  # scene.cpp:278-288
        self._objects.append(obj)
        obj._scene = self
        if obj.instance_index == 0:
            obj.instance_index = len(self._objects)

    def remove_object(self, obj):
        if obj in self._objects:
            self._objects = [o for o in self._objects if o is not obj]
            obj._scene = None

    # ---- lights ----------------------------------------------------------------------------
    @property
    def light_directions(self):
        return self._light_directions  # a view into scene memory (py_scene.cpp:284-309)

    @light_directions.setter
    def light_directions(self, v):
        v = torch.as_tensor(v, dtype=torch.float32).reshape(-1, 3)
        if v.shape[0] > NUM_LIGHTS:
            raise ValueError("Cannot support that many lights")  # scene.cpp:418-419
        self._light_directions.zero_()
        self._light_directions[: v.shape[0]].copy_(v)

    @property
    def light_colors(self):
        return self._light_colors

    @light_colors.setter
    def light_colors(self, v):
        v = torch.as_tensor(v, dtype=torch.float32).reshape(-1, 3)
        if v.shape[0] > NUM_LIGHTS:
            raise ValueError("Cannot support that many lights")
        self._light_colors.zero_()
        self._light_colors[: v.shape[0]].copy_(v)

    @property
    def light_position(self):
        warnings.warn("Scene.light_position is deprecated, use light_directions", DeprecationWarning)
        return -self._light_directions[0]

    @light_position.setter
    def light_position(self, v):
        warnings.warn("Scene.light_position is deprecated, use light_directions", DeprecationWarning)
        self._light_directions[0] = -torch.as_tensor(v, dtype=torch.float32).reshape(3)

    @property
    def ambient_light(self):
        return torch.from_numpy(self._ambient_light.copy())

    @ambient_light.setter
    def ambient_light(self, v):
        self._ambient_light = as_vec(v, 3)

    def choose_random_light_direction(self):  # scene.cpp:453-470
        g = self._rng
        d = np.array([g.standard_normal(), -abs(g.standard_normal()), -abs(g.standard_normal())], dtype=np.float32)
        light_dir_in_cam = -M.normalized(M.normalized(d))
        world = M.transform_vector(self._camera_pose, light_dir_in_cam)
        self.light_directions = world.reshape(1, 3)

    def choose_random_light_position(self):  # py_scene.cpp:350-352: no-op + warning (quirk q3)
        warnings.warn("choose_random_light_position() is deprecated and does nothing; "
                      "use choose_random_light_direction()")

    # ---- misc properties -------------------------------------------------------------------
    @property
    def background_color(self):
        return torch.from_numpy(self._background_color.copy())

    @background_color.setter
    def background_color(self, v):
        self._background_color = as_vec(v, 4)

    @property
    def background_image(self):
        return self._background_image

    @background_image.setter
    def background_image(self, v):
        """A Texture (rectangle texture of the reference, py_scene.cpp:131-140) stretched over the viewport
        behind the objects, or None."""
        self._background_image = v

    @property
    def light_map(self):
        return self._light_map

    @light_map.setter
    def light_map(self, v):
        from .light_map import LightMap

        if v is not None and not isinstance(v, LightMap):
            v = LightMap(v)          # a path, as Scene::deserialize passes it (scene.cpp:841-842)
        self._light_map = v

    @property
    def background_plane_pose(self):
        return torch.from_numpy(self._background_plane_pose.copy())

    @background_plane_pose.setter
    def background_plane_pose(self, v):
        self._background_plane_pose = as_mat4(v)

    @property
    def background_plane_size(self):
        return torch.from_numpy(self._background_plane_size.copy())

    @background_plane_size.setter
    def background_plane_size(self, v):
        self._background_plane_size = as_vec(v, 2)

    @property
    def background_plane_texture(self):
        return self._background_plane_texture

    @background_plane_texture.setter
    def background_plane_texture(self, v):
        self._background_plane_texture = v

    @property
    def manual_exposure(self):
        return float(self._manual_exposure)

    @manual_exposure.setter
    def manual_exposure(self, v):
        self._manual_exposure = f32(v)

    def load_visual(self):
        from ._context import engine

        for o in self._objects:
            engine().register_mesh(o.mesh)

    def load_physics(self):
        for o in self._objects:
            o.mesh._load_physics()
        self._physics_loaded = True

    # ---- physics (implemented in physics.py) -------------------------------------------------
    def simulate_tabletop_scene(self, vis_cb=None):
        from . import physics

        physics.simulate_tabletop_scene(self, vis_cb)

    def simulate(self, dt):
        from . import physics

        physics.simulate(self, float(dt))

    def check_collisions(self):
        from . import physics

        physics.check_collisions(self)

    def find_noncolliding_pose(self, obj, sampler="random", max_iterations=10, **kwargs):
        from . import physics

        return physics.find_noncolliding_pose(self, obj, sampler, int(max_iterations), **kwargs)

    def choose_random_camera_pose(self):
        from . import camera_placement

        az = f32(self._rng.uniform(-math.pi, math.pi))
        el = f32(self._rng.uniform(math.radians(30.0), math.pi / 2.0 - math.radians(30.0)))
        self._camera_pose = camera_placement.choose_camera_pose(self, az, el)

    # ---- (de)serialisation: 'next' row f3 ------------------------------------------------------
    def serialize(self):
        from . import serialization

        return serialization.serialize(self)

    def deserialize(self, text, cache=None):
        from . import serialization

        serialization.deserialize(self, text, cache)
