"""Pose samplers (reference include/stillleben/pose.h:25-218, src/pose.cpp:24-34)."""
import numpy as np

from ._math import f32


def random_quaternion(rng):
    """Normalised 4-vector of N(0,1) draws, [x y z w] (pose.h:25-35)."""
    q = rng.standard_normal(4).astype(np.float32)
    return (q / f32(np.sqrt(np.dot(q, q)))).astype(np.float32)


def minimum_distance_for_object_diameter(diameter, P):
    """pose.cpp:24-34: distance at which an object of `diameter` fills the frame."""
    return f32(max(P[0, 0], P[1, 1])) * f32(diameter) / f32(2.0)


class RandomPositionSampler:
    """pose.h:56-99: z ~ U(1.2 d_min, d_min / min_size_factor); x,y within 80 % of the frustum."""

    def __init__(self, P, diameter, min_size_factor=0.4):
        self.P = np.asarray(P, dtype=np.float32)
        self.diameter = f32(diameter)
        self.min_size_factor = f32(min_size_factor)

    def __call__(self, rng):
        dmin = minimum_distance_for_object_diameter(self.diameter, self.P)
        z = f32(rng.uniform(float(f32(1.2) * dmin), float(dmin / self.min_size_factor)))
        # NDC x = P00 x / z + P02  =>  x = (ndc - P02) z / P00
        nx = f32(rng.uniform(-0.8, 0.8))
        ny = f32(rng.uniform(-0.8, 0.8))
        x = (nx - self.P[0, 2]) * z / self.P[0, 0]
        y = (ny - self.P[1, 2]) * z / self.P[1, 1]
        return np.array([x, y, z], dtype=np.float32)
