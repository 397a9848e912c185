"""Convex collision shapes for a mesh (reference Mesh::loadPhysics, src/mesh.cpp:304-533).

The reference runs V-HACD twice (single hull, then concavity 0.002) and uses the decomposition
only when its volume is < 75 % of the hull's (mesh.cpp:426-429).  V-HACD is a vendored
third-party library of the reference and is not shipped here; this module provides
  * the single convex hull (exact, scipy/Qhull) -- what PHYSICS_FORCE_CONVEX_HULL and every
    mesh with convexity >= 0.75 uses, and
  * pre-computed decompositions: `<mesh>.hulls.npz` next to the mesh file (generated with the
    reference's V-HACD parameters by oracle/ref_build/gen_hulls.py) is picked up when present,
  * a built-in approximate decomposition (recursive plane splits of the surface along the
    longest axis while the hull volume ratio is < 0.75) for concave meshes without a cache.
Hull vertices are limited to 64 (VHACD.h:235)."""
import os

import numpy as np

MAX_HULL_VERTS = 64


class Hull:
    def __init__(self, vertices, triangles):
        self.vertices = np.asarray(vertices, dtype=np.float32)
        self.triangles = np.asarray(triangles, dtype=np.int32)

    def volume(self):
        v = self.vertices.astype(np.float64)
        t = self.triangles
        return float(np.abs(np.einsum("ij,ij->i", v[t[:, 0]], np.cross(v[t[:, 1]], v[t[:, 2]])).sum()) / 6.0)


def _qhull(points):
    from scipy.spatial import ConvexHull, QhullError

    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    try:
        h = ConvexHull(pts)
    except QhullError:
        h = ConvexHull(pts, qhull_options="QJ")
    verts = h.vertices
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[verts] = np.arange(len(verts))
    tris = remap[h.simplices]
    v = pts[verts]
    # orient outward
    c = v.mean(axis=0)
    n = np.cross(v[tris[:, 1]] - v[tris[:, 0]], v[tris[:, 2]] - v[tris[:, 0]])
    flip = np.einsum("ij,ij->i", n, v[tris[:, 0]] - c) < 0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    return v, tris


def _reduce(points, max_verts=MAX_HULL_VERTS):
    """Hull with at most max_verts vertices: greedy inside-out construction -- start from the
    axis extremes and repeatedly add the hull vertex that lies farthest outside the current
    polytope (the vertex-limited quick-hull PhysX cooking performs, [ext])."""
    from scipy.spatial import ConvexHull, QhullError

    v, t = _qhull(points)
    if len(v) <= max_verts:
        return Hull(v, t)
    sel = []
    for ax in range(3):
        for j in (int(np.argmin(v[:, ax])), int(np.argmax(v[:, ax]))):
            if j not in sel:
                sel.append(j)
    while len(sel) < max_verts:
        try:
            h = ConvexHull(v[sel])
        except QhullError:
            h = ConvexHull(v[sel], qhull_options="QJ")
        d = (v @ h.equations[:, :3].T + h.equations[:, 3]).max(axis=1)
        d[sel] = -np.inf
        j = int(np.argmax(d))
        if d[j] <= 1e-12:
            break
        sel.append(j)
    v2, t2 = _qhull(v[sel])
    return Hull(v2, t2)


def convex_hull(points):
    return _reduce(points)


def _mesh_volume(pos, idx):
    t = idx.reshape(-1, 3)
    p = pos.astype(np.float64)
    return float(np.abs(np.einsum("ij,ij->i", p[t[:, 0]], np.cross(p[t[:, 1]], p[t[:, 2]])).sum()) / 6.0)


def _decompose(pos, tris, depth, max_depth=4):
    """Recursive surface split along the longest bbox axis until each part is ~convex."""
    used = np.unique(tris)
    pts = pos[used]
    hull = _reduce(pts)
    if depth >= max_depth or len(tris) < 16:
        return [hull]
    # part "volume": sum of signed tets against the part centroid is ill-defined for open
    # parts; use the ratio of surface-sampled hull thickness instead: split while the part's
    # points are far inside their own hull on average
    c = pts.mean(axis=0)
    hv = hull.vertices.astype(np.float64)
    ht = hull.triangles
    n = np.cross(hv[ht[:, 1]] - hv[ht[:, 0]], hv[ht[:, 2]] - hv[ht[:, 0]])
    ln = np.linalg.norm(n, axis=1)
    keep = ln > 1e-20
    n = n[keep] / ln[keep, None]
    d = np.einsum("ij,ij->i", n, hv[ht[keep, 0]])
    # distance of every surface point to the closest hull plane
    dist = (d[None, :] - pts.astype(np.float64) @ n.T).min(axis=1)
    ext = pts.max(axis=0) - pts.min(axis=0)
    if dist.mean() < 0.02 * float(np.linalg.norm(ext)):
        return [hull]
    ax = int(np.argmax(ext))
    cen = pos[tris].mean(axis=1)[:, ax]
    cut = np.median(cen)
    a, b = tris[cen <= cut], tris[cen > cut]
    if len(a) == 0 or len(b) == 0:
        return [hull]
    del c
    return _decompose(pos, a, depth + 1, max_depth) + _decompose(pos, b, depth + 1, max_depth)


def hulls_for_mesh(mesh):
    from .mesh import Mesh

    data = mesh._data
    cache = mesh._filename + ".hulls.npz"
    force = bool(mesh._flags & Mesh.Flag.PHYSICS_FORCE_CONVEX_HULL)
    if not force and os.path.exists(cache):
        z = np.load(cache)
        n = int(z["n_hulls"])
        return [Hull(z["v%d" % i], z["t%d" % i]) for i in range(n)]
    single = _reduce(data.positions)
    if force:
        return [single]
    tris = data.indices.reshape(-1, 3).astype(np.int64)
    mv = _mesh_volume(data.positions, data.indices)
    hv = single.volume()
    if hv < 1e-9 or mv / max(hv, 1e-30) >= 0.75:  # mesh.cpp:373-378, :426-429
        return [single]
    parts = _decompose(data.positions, tris, 0)
    return parts if len(parts) > 1 else [single]
