#!/usr/bin/env python3
"""Generates the synthetic stand-ins for <YCB-Video>/models/<class>/textured.obj used by the acceptance test
(tests/test_gpu_acceptance.py): the real models are not redistributable / not available offline.  Four classes in the
dataset's layout and units (metres): a can, a box, an L-shaped "drill" (concave -> V-HACD decomposes it) and a
flat "clamp" frame with a hole (concave).  Each: textured.obj (v / vt / vn / f with usemtl), textured.mtl (map_Kd),
texture_map.png (32x32).  Run from the repository root: python tests/fixtures/ycb_mini/make_fixtures.py"""
import math
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def box(lo, hi):
    """Closed box as 12 outward triangles; returns (verts [8,3], tris [12,3])."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    v = np.array([[x, y, z] for z in (lo[2], hi[2]) for y in (lo[1], hi[1]) for x in (lo[0], hi[0])])
    t = np.array([[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [2, 6, 3], [3, 6, 7],
                  [0, 4, 2], [2, 4, 6], [1, 3, 5], [3, 7, 5]])
    return v, t


def cylinder(r, h, n=24):
    v = [[0, 0, -h / 2], [0, 0, h / 2]]
    for i in range(n):
        a = 2 * math.pi * i / n
        v += [[r * math.cos(a), r * math.sin(a), -h / 2], [r * math.cos(a), r * math.sin(a), h / 2]]
    t = []
    for i in range(n):
        b0, t0 = 2 + 2 * i, 3 + 2 * i
        b1, t1 = 2 + 2 * ((i + 1) % n), 3 + 2 * ((i + 1) % n)
        t += [[0, b1, b0], [1, t0, t1], [b0, b1, t0], [b1, t1, t0]]
    return np.array(v, float), np.array(t)


def merge(parts):
    vs, ts, off = [], [], 0
    for v, t in parts:
        vs.append(v); ts.append(t + off); off += len(v)
    return np.concatenate(vs), np.concatenate(ts)


CLASSES = {
    "002_master_chef_can": lambda: cylinder(0.051, 0.14),
    "003_cracker_box": lambda: box((-0.03, -0.08, -0.105), (0.03, 0.08, 0.105)),
    "035_power_drill": lambda: merge([box((-0.02, -0.09, -0.025), (0.02, 0.09, 0.025)),          # body
                                      box((-0.018, 0.03, -0.15), (0.018, 0.075, -0.025))]),      # handle: an L
    "051_large_clamp": lambda: merge([box((-0.08, -0.06, -0.01), (0.08, -0.035, 0.01)), box((-0.08, 0.035, -0.01), (0.08, 0.06, 0.01)),
                                      box((-0.08, -0.035, -0.01), (-0.055, 0.035, 0.01)), box((0.055, -0.035, -0.01), (0.08, 0.035, 0.01))]),
}


def write(name, v, t, seed):
    d = os.path.join(HERE, "models", name)
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed)
    base = rng.integers(60, 230, 3)
    img = np.clip(base[None, None] + rng.integers(-40, 40, (32, 32, 3)), 0, 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(d, "texture_map.png"))
    with open(os.path.join(d, "textured.mtl"), "w") as f:
        f.write("newmtl material_0\nKa 0.2 0.2 0.2\nKd 0.8 0.8 0.8\nKs 0.1 0.1 0.1\nNs 10.0\nmap_Kd texture_map.png\n")
    # flat shading: one normal per face; planar uv from the two dominant axes of the bbox
    ext = v.max(0) - v.min(0)
    ax = np.argsort(-ext)[:2]
    uv = (v[:, ax] - v.min(0)[ax]) / ext[ax]
    with open(os.path.join(d, "textured.obj"), "w") as f:
        f.write("# synthetic stand-in for the YCB-Video model '%s' (tests/fixtures/ycb_mini/make_fixtures.py)\nmtllib textured.mtl\n" % name)
        for p in v:
            f.write("v %.6f %.6f %.6f\n" % tuple(p))
        for q in uv:
            f.write("vt %.6f %.6f\n" % tuple(q))
        for tri in t:
            n = np.cross(v[tri[1]] - v[tri[0]], v[tri[2]] - v[tri[0]])
            n = n / np.linalg.norm(n)
            f.write("vn %.6f %.6f %.6f\n" % tuple(n))
        f.write("usemtl material_0\n")
        for k, tri in enumerate(t):
            f.write("f " + " ".join("%d/%d/%d" % (i + 1, i + 1, k + 1) for i in tri) + "\n")


if __name__ == "__main__":
    for s, (name, make) in enumerate(CLASSES.items()):
        v, t = make()
        write(name, v, t, s)
        print(name, len(v), "vertices", len(t), "triangles")
