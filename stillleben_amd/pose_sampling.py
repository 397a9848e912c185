"""Pose samplers (reference include/stillleben/pose.h:25-218, src/pose.cpp:8-62).  numpy RNG
streams replace libstdc++'s; the DISTRIBUTIONS are the contract."""
import math

import numpy as np

from ._math import f32, quat_to_matrix


def random_quaternion(rng):
    """Normalised 4-vector of N(0,1) draws, [x y z w] (pose.h:25-35)."""
    q = rng.standard_normal(4).astype(np.float32)
    return (q / f32(np.sqrt(np.dot(q, q)))).astype(np.float32)


def random_rotation(rng):
    return quat_to_matrix(random_quaternion(rng))


def minimum_distance_for_object_diameter(diameter, P):
    """pose.cpp:24-34: max(P00, P11) * diameter / 2."""
    return f32(max(P[0, 0] * f32(diameter) / f32(2.0), P[1, 1] * f32(diameter) / f32(2.0)))


def cross_matrix(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float32)


def rotation_correction_for_translation(pos):
    """pose.cpp:36-60: counters the apparent rotation of an object translated in the FOV."""
    a = np.asarray(pos, np.float32)
    a = a / f32(np.sqrt(np.dot(a, a)))
    z = np.array([0, 0, 1], np.float32)
    v = np.cross(a, z).astype(np.float32)
    s = f32(np.sqrt(np.dot(v, v)))
    c = f32(np.dot(a, z))
    if abs(s) < 1e-5:
        return np.eye(3, dtype=np.float32)
    vx = cross_matrix(v)
    R = np.eye(3, dtype=np.float32) + vx + (f32(1.0) - c) / (s * s) * (vx @ vx)
    return R.T.astype(np.float32)


def perpendicular_vector(x):
    """pose.h:119-127."""
    x = np.asarray(x, np.float32)
    other = np.array([0, 1, 0], np.float32) if abs(x[0]) > 0.8 else np.array([1, 0, 0], np.float32)
    v = np.cross(x, other).astype(np.float32)
    return v / f32(np.sqrt(np.dot(v, v)))


class RandomPositionSampler:
    """pose.h:56-99: z ~ U(1.2 d, d / min_size_factor) with d = minimumDistanceForObjectDiameter;
    x ~ U(-0.8 z / P00, 0.8 z / P00), y likewise with P11."""

    def __init__(self, P, diameter, min_size_factor=0.4):
        self.P = np.asarray(P, dtype=np.float32)
        self.fully_visible = minimum_distance_for_object_diameter(diameter, self.P)
        self.min_size_factor = f32(min_size_factor)

    def __call__(self, rng):
        z = f32(rng.uniform(float(f32(1.2) * self.fully_visible), float(self.fully_visible / self.min_size_factor)))
        xr = f32(0.8) * z / self.P[0, 0]
        yr = f32(0.8) * z / self.P[1, 1]
        return np.array([rng.uniform(-xr, xr), rng.uniform(-yr, yr), z], dtype=np.float32)


def _pose(R, t):
    m = np.eye(4, dtype=np.float32)
    m[:3, :3], m[:3, 3] = R, t
    return m


class RandomPoseSampler:
    def __init__(self, position_sampler):
        self.pos = position_sampler

    def __call__(self, rng):
        R = random_rotation(rng)          # rotation first, then position (pose.h:109-113)
        return _pose(R, self.pos(rng))


class ViewPointPoseSampler:
    """pose.h:130-190: the object's `view_point` direction faces the camera, random roll."""

    def __init__(self, position_sampler, view_point=(1.0, 0.0, 0.0)):
        self.pos = position_sampler
        self.view_point = np.asarray(view_point, np.float32)

    def __call__(self, rng):
        pos = self.pos(rng)
        x = -pos / f32(np.sqrt(np.dot(pos, pos)))
        y = perpendicular_vector(x)
        xf = np.stack([x, y, np.cross(x, y)], axis=1).astype(np.float32)      # columns
        ang = rng.uniform(-math.pi, math.pi)
        c, s = f32(math.cos(ang)), f32(math.sin(ang))
        xrot = np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float32)
        r0 = self.view_point
        r1 = perpendicular_vector(r0)
        vpx = np.stack([r0, r1, np.cross(r0, r1)], axis=0).astype(np.float32)  # rows
        return _pose((xf @ xrot @ vpx).astype(np.float32), pos)


class ViewCorrectedPoseSampler:
    """pose.h:192-216: fixed orientation, corrected for the perspective of its position."""

    def __init__(self, position_sampler, orientation):
        self.pos = position_sampler
        self.orientation = np.asarray(orientation, np.float32).reshape(3, 3)

    def __call__(self, rng):
        pos = self.pos(rng)
        return _pose((rotation_correction_for_translation(pos) @ self.orientation).astype(np.float32), pos)
