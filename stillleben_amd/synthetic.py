"""Procedural "YCB-like" object set for benchmarks and tests: the
YCB-Video models are not redistributable/available offline, so each of the 21 classes of
reference examples/ycb.py:21-30 is approximated by a compound of convex primitives
(cans = cylinders, boxes, bowl/mug = rings of wall segments, banana = bent chain, drill/clamps =
box compounds), tessellated to ~8192 vertices / ~16384 triangles with a seeded 1024^2 noise
texture.  Collision hulls: like every mesh of the reference these go through Mesh::loadPhysics' V-HACD
procedure (mesh.cpp:335-470) -- run once in the build container with the reference's own V-HACD as a tool
(oracle/ref_build/gen_hulls.py --ycb); the result for the default set (seed 0, 8192 vertices) is shipped as
data/ycb_like_hulls_seed0.npz, keyed by the meshes' geometry digests.  Other seeds / sizes: the in-tree decomposition.
`hulls="parts"` gives the by-construction hulls instead (one convex hull per primitive part)."""
import math
import os

import numpy as np

from . import _loaders
from .hulls import Hull, _qhull

YCB_CLASSES = (
    '002_master_chef_can', '003_cracker_box', '004_sugar_box', '005_tomato_soup_can', '006_mustard_bottle',
    '007_tuna_fish_can', '008_pudding_box', '009_gelatin_box', '010_potted_meat_can', '011_banana',
    '019_pitcher_base', '021_bleach_cleanser', '024_bowl', '025_mug', '035_power_drill', '036_wood_block',
    '037_scissors', '040_large_marker', '051_large_clamp', '052_extra_large_clamp', '061_foam_brick',
)


def _rot(axis, ang):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s = math.cos(ang), math.sin(ang)
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


def _grid_patch(origin, du, dv, nu, nv, normal):
    """(nu+1)x(nv+1) vertex grid spanning origin + a du + b dv."""
    a = np.linspace(0, 1, nu + 1)
    b = np.linspace(0, 1, nv + 1)
    A, B = np.meshgrid(a, b, indexing="ij")
    pos = origin[None, None] + A[..., None] * du[None, None] + B[..., None] * dv[None, None]
    uv = np.stack([A, B], axis=-1)
    idx = []
    for i in range(nu):
        for j in range(nv):
            v0 = i * (nv + 1) + j
            v1 = (i + 1) * (nv + 1) + j
            idx += [v0, v1, v1 + 1, v0, v1 + 1, v0 + 1]
    nrm = np.broadcast_to(normal, pos.shape)
    return pos.reshape(-1, 3), nrm.reshape(-1, 3).copy(), uv.reshape(-1, 2), np.array(idx, np.int64)


def box_part(size, n):
    """Axis-aligned box centred at the origin; n = grid resolution per face."""
    sx, sy, sz = [s / 2.0 for s in size]
    faces = [
        ((-sx, -sy, sz), (2 * sx, 0, 0), (0, 2 * sy, 0), (0, 0, 1)),
        ((-sx, sy, -sz), (2 * sx, 0, 0), (0, -2 * sy, 0), (0, 0, -1)),
        ((sx, -sy, -sz), (0, 2 * sy, 0), (0, 0, 2 * sz), (1, 0, 0)),
        ((-sx, sy, -sz), (0, -2 * sy, 0), (0, 0, 2 * sz), (-1, 0, 0)),
        ((sx, sy, -sz), (-2 * sx, 0, 0), (0, 0, 2 * sz), (0, 1, 0)),
        ((-sx, -sy, -sz), (2 * sx, 0, 0), (0, 0, 2 * sz), (0, -1, 0)),
    ]
    P, N, U, I = [], [], [], []
    off = 0
    for k, (o, du, dv, nn) in enumerate(faces):
        p, nr, uv, idx = _grid_patch(np.array(o, float), np.array(du, float), np.array(dv, float), n, n, np.array(nn, float))
        uv = uv * 0.3 + np.array([(k % 3) * 0.33, (k // 3) * 0.5])
        P.append(p); N.append(nr); U.append(uv); I.append(idx + off)
        off += len(p)
    corners = np.array([[x, y, z] for x in (-sx, sx) for y in (-sy, sy) for z in (-sz, sz)])
    return np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(I), corners


def cylinder_part(radius, height, n_seg, n_ring, hull_seg=28):
    """Cylinder along z centred at the origin (side wall grid + two fan caps)."""
    th = np.linspace(0, 2 * math.pi, n_seg + 1)
    zz = np.linspace(-height / 2, height / 2, n_ring + 1)
    T, Z = np.meshgrid(th, zz, indexing="ij")
    pos = np.stack([radius * np.cos(T), radius * np.sin(T), Z], axis=-1).reshape(-1, 3)
    nrm = np.stack([np.cos(T), np.sin(T), np.zeros_like(T)], axis=-1).reshape(-1, 3)
    uv = np.stack([T / (2 * math.pi), (Z + height / 2) / height * 0.8], axis=-1).reshape(-1, 2)
    idx = []
    for i in range(n_seg):
        for j in range(n_ring):
            v0 = i * (n_ring + 1) + j
            v1 = (i + 1) * (n_ring + 1) + j
            idx += [v0, v1, v1 + 1, v0, v1 + 1, v0 + 1]
    P, N, U, I = [pos], [nrm], [uv], [np.array(idx, np.int64)]
    off = len(pos)
    for s in (-1, 1):
        c = np.array([[0, 0, s * height / 2]])
        ring = np.stack([radius * np.cos(th[:-1]), radius * np.sin(th[:-1]), np.full(n_seg, s * height / 2)], axis=-1)
        p = np.concatenate([c, ring])
        n_ = np.tile([[0, 0, s]], (len(p), 1)).astype(float)
        u = np.concatenate([[[0.5, 0.9]], np.stack([0.5 + 0.08 * np.cos(th[:-1]), 0.9 + 0.08 * np.sin(th[:-1])], axis=-1)])
        ii = []
        for k in range(n_seg):
            a, b = 1 + k, 1 + (k + 1) % n_seg
            ii += [0, a, b] if s > 0 else [0, b, a]
        P.append(p); N.append(n_); U.append(u); I.append(np.array(ii, np.int64) + off)
        off += len(p)
    hth = np.linspace(0, 2 * math.pi, hull_seg, endpoint=False)
    hull = np.array([[radius * math.cos(t), radius * math.sin(t), z] for t in hth for z in (-height / 2, height / 2)])
    return np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(I), hull


def _class_parts(name, rng):
    """List of (kind, params, R(3x3), t(3)) convex parts, dimensions in metres (rough YCB sizes)."""
    I3 = np.eye(3)
    z = np.zeros(3)
    if name.endswith("_can") and "potted" not in name:
        dims = {"002_master_chef_can": (0.051, 0.14), "005_tomato_soup_can": (0.033, 0.10), "007_tuna_fish_can": (0.042, 0.033)}
        r, h = dims[name]
        return [("cyl", (r, h), I3, z)]
    if name in ("003_cracker_box", "004_sugar_box", "008_pudding_box", "009_gelatin_box", "010_potted_meat_can",
                "036_wood_block", "061_foam_brick"):
        dims = {"003_cracker_box": (0.06, 0.158, 0.21), "004_sugar_box": (0.038, 0.089, 0.175),
                "008_pudding_box": (0.035, 0.11, 0.089), "009_gelatin_box": (0.028, 0.085, 0.073),
                "010_potted_meat_can": (0.05, 0.097, 0.082), "036_wood_block": (0.085, 0.085, 0.2),
                "061_foam_brick": (0.05, 0.075, 0.05)}
        return [("box", dims[name], I3, z)]
    if name in ("006_mustard_bottle", "021_bleach_cleanser", "019_pitcher_base"):
        dims = {"006_mustard_bottle": (0.045, 0.14, 0.015, 0.05), "021_bleach_cleanser": (0.05, 0.2, 0.02, 0.05),
                "019_pitcher_base": (0.07, 0.2, 0.0, 0.0)}
        r, h, rn, hn = dims[name]
        parts = [("cyl", (r, h), I3, z)]
        if rn > 0:
            parts.append(("cyl", (rn, hn), I3, np.array([0, 0, h / 2 + hn / 2 - 0.002])))
        else:  # pitcher handle
            parts.append(("box", (0.02, 0.03, 0.12), I3, np.array([r + 0.02, 0, 0.0])))
            parts.append(("box", (0.04, 0.03, 0.02), I3, np.array([r + 0.005, 0, 0.06])))
            parts.append(("box", (0.04, 0.03, 0.02), I3, np.array([r + 0.005, 0, -0.06])))
        return parts
    if name in ("024_bowl", "025_mug"):
        r, h, wall = (0.08, 0.055, 0.008) if name == "024_bowl" else (0.042, 0.082, 0.006)
        n_wall = 12
        parts = [("cyl", (r, wall), I3, np.array([0, 0, -h / 2 + wall / 2]))]
        seg_len = 2 * math.pi * (r - wall / 2) / n_wall * 1.05
        for k in range(n_wall):
            a = 2 * math.pi * k / n_wall
            R = _rot((0, 0, 1), a)
            t = R @ np.array([r - wall / 2, 0, 0])
            parts.append(("box", (wall, seg_len, h), R, t))
        if name == "025_mug":
            parts.append(("box", (0.03, 0.012, 0.012), I3, np.array([r + 0.012, 0, 0.02])))
            parts.append(("box", (0.03, 0.012, 0.012), I3, np.array([r + 0.012, 0, -0.02])))
            parts.append(("box", (0.012, 0.012, 0.052), I3, np.array([r + 0.027, 0, 0.0])))
        return parts
    if name == "011_banana":
        parts = []
        n = 7
        rad = 0.09
        for k in range(n):
            a = (k - (n - 1) / 2) * 0.28
            R = _rot((0, 1, 0), a) @ _rot((1, 0, 0), math.pi / 2) @ _rot((0, 0, 1), 0.0)
            Rz = _rot((0, 1, 0), -a)
            t = np.array([rad * math.sin(a), 0, rad * (1 - math.cos(a))])
            parts.append(("cyl", (0.017 - 0.002 * abs(k - 3), 0.03), Rz @ _rot((0, 1, 0), math.pi / 2), t))
            del R
        return parts
    if name == "035_power_drill":
        return [("box", (0.05, 0.18, 0.06), I3, np.array([0, 0, 0.06])),
                ("cyl", (0.02, 0.05), _rot((1, 0, 0), math.pi / 2), np.array([0, 0.11, 0.06])),
                ("box", (0.04, 0.05, 0.11), _rot((1, 0, 0), 0.2), np.array([0, -0.03, -0.02])),
                ("box", (0.07, 0.1, 0.04), I3, np.array([0, -0.04, -0.09]))]
    if name == "037_scissors":
        return [("box", (0.012, 0.1, 0.004), _rot((0, 0, 1), 0.12), np.array([0.004, 0.05, 0])),
                ("box", (0.012, 0.1, 0.004), _rot((0, 0, 1), -0.12), np.array([-0.004, 0.05, 0.004])),
                ("box", (0.03, 0.06, 0.01), _rot((0, 0, 1), 0.25), np.array([0.018, -0.04, 0])),
                ("box", (0.03, 0.06, 0.01), _rot((0, 0, 1), -0.25), np.array([-0.018, -0.04, 0.004]))]
    if name == "040_large_marker":
        return [("cyl", (0.009, 0.12), I3, z)]
    if name in ("051_large_clamp", "052_extra_large_clamp"):
        s = 1.0 if name == "051_large_clamp" else 1.35
        return [("box", (0.02 * s, 0.12 * s, 0.012 * s), _rot((0, 0, 1), 0.15), np.array([0.015 * s, 0.02 * s, 0])),
                ("box", (0.02 * s, 0.12 * s, 0.012 * s), _rot((0, 0, 1), -0.15), np.array([-0.015 * s, 0.02 * s, 0])),
                ("box", (0.06 * s, 0.02 * s, 0.012 * s), I3, np.array([0, -0.035 * s, 0]))]
    raise KeyError(name)


def _noise_texture(rng, size=1024):
    base = rng.integers(40, 255, size=3)
    coarse = rng.integers(0, 255, size=(size // 32, size // 32, 3)).astype(np.float32)
    tex = np.kron(coarse, np.ones((32, 32, 1), np.float32))
    fine = rng.integers(0, 64, size=(size, size, 3)).astype(np.float32)
    rgb = np.clip(0.5 * base[None, None] + 0.35 * tex + 0.4 * fine, 0, 255).astype(np.uint8)
    return np.concatenate([rgb, np.full((size, size, 1), 255, np.uint8)], axis=2)


def make_class_mesh(name, seed=0, target_verts=8192, tex_size=1024):
    """Returns (ConsolidatedMesh, [Hull, ...]) for one YCB class."""
    rng = np.random.default_rng([seed, YCB_CLASSES.index(name)])
    parts = _class_parts(name, rng)
    per = max(64, target_verts // len(parts))
    P, N, U, I, hulls = [], [], [], [], []
    off = 0
    for kind, prm, R, t in parts:
        if kind == "box":
            n = max(1, int(round(math.sqrt(per / 6.0))) - 1)
            p, nr, uv, idx, hv = box_part(prm, n)
        else:
            n_seg = 64 if per >= 1024 else 24
            n_ring = max(1, per // (n_seg + 1) - 2)
            p, nr, uv, idx, hv = cylinder_part(prm[0], prm[1], n_seg, n_ring)
        P.append(p @ R.T + t); N.append(nr @ R.T); U.append(uv); I.append(idx + off)
        off += len(p)
        v, tr = _qhull(hv @ R.T + t)
        hulls.append(Hull(v, tr))
    cm = _loaders.ConsolidatedMesh()
    cm.positions = np.concatenate(P).astype(np.float32)
    cm.normals = np.concatenate(N).astype(np.float32)
    cm.uvs = np.concatenate(U).astype(np.float32)
    cm.colors = np.ones((len(cm.positions), 4), np.float32)
    cm.indices = np.concatenate(I).astype(np.uint32)
    cm.textures = [_noise_texture(rng, tex_size)]
    cm._tex_alpha = [False]
    cm.materials = [_loaders.Material(base_color=(1, 1, 1, 1), base_texture=0)]
    cm.submeshes = [_loaders.SubMesh(0, len(cm.indices), 0)]
    return cm, hulls


HULL_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ycb_like_hulls_seed%d.npz")


def _shipped_hulls(seed, cms):
    """The V-HACD hulls shipped for this seed, if every mesh's geometry digests match; else None."""
    from . import hulls as H

    path = HULL_DATA % seed
    if not os.path.exists(path):
        return None
    z = np.load(path)
    out = []
    for name, cm in zip(YCB_CLASSES, cms):
        key = name + "/"
        if key + "digests" not in z.files or tuple(int(x) for x in z[key + "digests"]) != H._mesh_digests(cm):
            return None
        out.append([H.Hull(z["%sv%d" % (key, i)], z["%st%d" % (key, i)]) for i in range(int(z[key + "n"]))])
    return out


def _native_hulls(cm):
    from . import hulls as H

    return H._compute_hulls(cm, False)


def native_hull_sets(seed=0, target_verts=8192):
    """The in-tree decomposition of the 21 classes (one list of hulls per class), computed in worker processes.  Needs no context:
    a caller that is about to initialise a device does this first (the workers are forked) and hands the result to
    ycb_like_meshes(hulls=<the list>)."""
    from concurrent.futures import ProcessPoolExecutor

    made = [make_class_mesh(name, seed, target_verts, 64) for name in YCB_CLASSES]   # (the texture does not enter the geometry)
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return list(ex.map(_native_hulls, [cm for cm, _ in made]))


def ycb_like_meshes(seed=0, target_verts=8192, tex_size=1024, hulls="vhacd"):
    """21 sl.Mesh objects named after the YCB-Video classes, class_index = i + 1
    (reference examples/ycb.py:46-48).  hulls: "vhacd" (the decompositions the reference's V-HACD gave this very geometry,
    shipped as data/ycb_like_hulls_seed<seed>.npz by oracle/ref_build/gen_hulls.py; for another seed / size the in-tree
    decomposition of hulls._compute_hulls computes them), "native" (always the in-tree decomposition), "parts" (by
    construction: one hull per primitive part) or a list of hull lists per class (native_hull_sets)."""
    from . import hulls as H
    from .mesh import Mesh

    made = [make_class_mesh(name, seed, target_verts, tex_size) for name in YCB_CLASSES]
    sets = [h for _, h in made]
    if isinstance(hulls, list):               # precomputed (native_hull_sets)
        if len(hulls) != len(made):
            raise ValueError("hulls: one list of hulls per class expected")
        sets = hulls
    elif hulls in ("vhacd", "native"):
        shipped = _shipped_hulls(seed, [cm for cm, _ in made]) if (hulls == "vhacd" and target_verts == 8192) else None
        if shipped is not None:
            sets = shipped
        else:
            from concurrent.futures import ProcessPoolExecutor

            with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
                sets = list(ex.map(_native_hulls, [cm for cm, _ in made]))
    elif hulls != "parts":
        raise ValueError("hulls must be 'vhacd', 'native' or 'parts'")
    out = []
    for i, (name, (cm, _), hs) in enumerate(zip(YCB_CLASSES, made, sets)):
        m = Mesh.from_data(cm, hs, "synthetic://ycb/%s" % name)
        m.class_index = i + 1
        out.append(m)
    return out
