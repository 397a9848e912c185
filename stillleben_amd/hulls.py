"""Convex collision shapes for a mesh (reference Mesh::loadPhysics, src/mesh.cpp:304-533).

The reference runs V-HACD twice (single hull, then concavity 0.002) and uses the decomposition
only when its volume is < 75 % of the hull's (mesh.cpp:426-429).  Here, in this order:
  * `<mesh>.hulls.npz` fixtures (tests/fixtures: cube, bunny) and the shipped set of the synthetic YCB-like classes
    (data/ycb_like_hulls_seed0.npz): decompositions made ONCE, in the build container, by the reference's own vendored
    V-HACD as a TOOL (oracle/ref_build/gen_hulls.py + vhacd_driver.cpp; nothing of it is linked into or shipped with the
    product); validated against the mesh's geometry digests;
  * `<mesh>.sl_hulls`: the cache, with the reference's invalidation keys (below);
  * the in-tree decomposition `acd.decompose` (voxelise, cut hierarchically by concavity, hulls limited to 64 vertices,
    VHACD.h:235) under the reference's selection rule -- tests/test_host_acd.py holds it against the V-HACD fixtures
    (same single-hull decisions, total hull volume within 10 %, hull counts within a factor of two)."""
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


def native_hull(points):
    """Triangles (int64 [T, 3], indices into `points`, outward) of the convex hull of float64 [N, 3] points: the quick-hull of
    libslhip.so (slhip_host_convex_hull, csrc/slhip_hull.cpp).  A cloud that spans no volume (a sheet, a rod: Qhull's QhullError)
    is joggled the way Qhull's 'QJ' does -- deterministic offsets of 1e-7 of its extent -- so that it still yields a (thin) hull;
    None when even that fails (fewer than four distinct points)."""
    import ctypes as C

    from . import _abi

    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = len(pts)
    if n < 4:
        return None
    L = _abi.lib()
    tris = np.zeros((max(4, 2 * n), 3), dtype=np.uint32)
    nt = C.c_uint32(0)
    rc = L.slhip_host_convex_hull(pts.ctypes.data, n, tris.ctypes.data, len(tris), C.byref(nt))
    if rc == 1:
        ext = float((pts.max(axis=0) - pts.min(axis=0)).max())
        if not ext > 0.0:
            return None
        k = np.arange(3 * n, dtype=np.uint64).reshape(n, 3)
        jog = ((k * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(1 << 20)).astype(np.float64) / float(1 << 20) - 0.5
        moved = np.ascontiguousarray(pts + jog * (2e-7 * ext))
        rc = L.slhip_host_convex_hull(moved.ctypes.data, n, tris.ctypes.data, len(tris), C.byref(nt))
    if rc != 0:
        if rc < 0:
            _abi.check(-1, 'host geometry')
        return None
    return tris[: nt.value].astype(np.int64)


def _hull_planes(v, tris):
    """Outward unit normals and offsets of a hull's triangles: n . x + d <= 0 inside."""
    n = np.cross(v[tris[:, 1]] - v[tris[:, 0]], v[tris[:, 2]] - v[tris[:, 0]])
    ln = np.linalg.norm(n, axis=1)
    ok = ln > 0
    n = n[ok] / ln[ok, None]
    return n, -np.einsum("ij,ij->i", n, v[tris[ok, 0]])


def hull_volume(points):
    """Volume of the convex hull of a point cloud (0 for a cloud that spans none)."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if len(pts) < 4:
        return 0.0
    import ctypes as C

    from . import _abi

    tris = np.zeros((max(4, 2 * len(pts)), 3), dtype=np.uint32)
    nt = C.c_uint32(0)
    if _abi.lib().slhip_host_convex_hull(pts.ctypes.data, len(pts), tris.ctypes.data, len(tris), C.byref(nt)) != 0:
        return 0.0
    t = tris[: nt.value].astype(np.int64)
    o = pts[t[0, 0]]
    return float(np.abs(np.einsum("ij,ij->i", pts[t[:, 0]] - o, np.cross(pts[t[:, 1]] - o, pts[t[:, 2]] - o)).sum()) / 6.0)


def _qhull(points):
    """(vertices float64 [V, 3], triangles int64 [T, 3]) of the hull of a point cloud -- vertices in the lexicographic order of their
    coordinates (what round 5's SciPy / Qhull path returned: np.unique's order, the hull's vertices in ascending index)."""
    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    tris = native_hull(pts)
    if tris is None:
        raise ValueError("convex hull of %d coincident / collinear points" % len(pts))
    verts = np.unique(tris)
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[verts] = np.arange(len(verts))
    return pts[verts], remap[tris]


def _reduce(points, max_verts=MAX_HULL_VERTS):
    """Hull with at most max_verts vertices: greedy inside-out construction -- start from the
    axis extremes and repeatedly add the hull vertex that lies farthest outside the current
    polytope (the vertex-limited quick-hull PhysX cooking performs, [ext])."""
    v, t = _qhull(points)
    if len(v) <= max_verts:
        return Hull(v, t)
    sel = []
    for ax in range(3):
        for j in (int(np.argmin(v[:, ax])), int(np.argmax(v[:, ax]))):
            if j not in sel:
                sel.append(j)
    while len(sel) < max_verts:
        tr = native_hull(v[sel])
        if tr is None:
            break
        nrm, off = _hull_planes(v[sel], tr)
        d = (v @ nrm.T + off).max(axis=1)
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


# ---- hull cache (reference mesh.cpp:94-172 readCacheFile, :490-511 write) ------------------------------
# `<mesh>.sl_mesh` of the reference holds PhysX-cooked hulls; its payload cannot be shared, but the
# INVALIDATION KEYS are the same here: format version, mesh flags, MurmurHash2 digests of the vertex
# positions and of the indices (Corrade::Utility::MurmurHash2 = MurmurHash64A on 64-bit hosts, default
# seed 23), and "cache newer than the source file".  The file is written atomically (utils/os.cpp).
CACHE_SUFFIX = ".sl_hulls"
CACHE_MAGIC = b"SLHULLS\0"
CACHE_VERSION = 1


def murmur64a(data, seed=23):
    """MurmurHash64A of a bytes-like object (the digest Corrade's MurmurHash2 yields on 64-bit hosts)."""
    b = bytes(data)
    n = len(b)
    m, r, mask = 0xC6A4A7935BD1E995, 47, (1 << 64) - 1
    h = (seed ^ ((n * m) & mask)) & mask
    nb = n // 8
    if nb:
        k = np.frombuffer(b, dtype="<u8", count=nb).copy()
        with np.errstate(over="ignore"):
            k *= np.uint64(m)
            k ^= k >> np.uint64(r)
            k *= np.uint64(m)
        for kk in k.tolist():            # the fold is inherently sequential
            h = ((h ^ kk) * m) & mask
    tail = b[nb * 8:]
    if tail:
        h ^= int.from_bytes(tail, "little")
        h = (h * m) & mask
    h ^= h >> r
    h = (h * m) & mask
    h ^= h >> r
    return h


def _mesh_digests(data):
    v = murmur64a(np.ascontiguousarray(data.positions, dtype="<f4").tobytes())
    i = murmur64a(np.ascontiguousarray(data.indices, dtype="<u4").tobytes())
    return v, i


def read_cache(cache_file, source_file, data, flags):
    """Returns the cached hulls or None (missing / stale / other version, flags or geometry)."""
    import struct

    if not os.path.exists(cache_file):
        return None
    if os.path.exists(source_file) and os.path.getmtime(cache_file) <= os.path.getmtime(source_file):
        return None   # "Cache file is stale" (mesh.cpp:133-137)
    try:
        with open(cache_file, "rb") as f:
            blob = f.read()
        if blob[:8] != CACHE_MAGIC:
            return None
        version, fl, vh, ih, n = struct.unpack_from("<IIQQI", blob, 8)
        if version != CACHE_VERSION or fl != int(flags):
            return None
        if (vh, ih) != _mesh_digests(data):
            return None
        off = 8 + struct.calcsize("<IIQQI")
        hulls = []
        for _ in range(n):
            nv, nt = struct.unpack_from("<II", blob, off)
            off += 8
            v = np.frombuffer(blob, dtype="<f4", count=3 * nv, offset=off).reshape(nv, 3).copy()
            off += 12 * nv
            t = np.frombuffer(blob, dtype="<i4", count=3 * nt, offset=off).reshape(nt, 3).copy()
            off += 12 * nt
            hulls.append(Hull(v, t))
        return hulls
    except (OSError, struct.error, ValueError):
        return None


def write_cache(cache_file, data, flags, hulls):
    import struct
    import tempfile

    vh, ih = _mesh_digests(data)
    parts = [CACHE_MAGIC, struct.pack("<IIQQI", CACHE_VERSION, int(flags), vh, ih, len(hulls))]
    for h in hulls:
        parts.append(struct.pack("<II", len(h.vertices), len(h.triangles)))
        parts.append(np.ascontiguousarray(h.vertices, dtype="<f4").tobytes())
        parts.append(np.ascontiguousarray(h.triangles, dtype="<i4").tobytes())
    try:
        fd, tmp = tempfile.mkstemp(prefix=os.path.basename(cache_file) + ".", dir=os.path.dirname(cache_file) or ".")
        with os.fdopen(fd, "wb") as f:
            f.write(b"".join(parts))
        os.replace(tmp, cache_file)       # atomic: readers see the old or the new file, never a torn one
    except OSError:
        pass                              # read-only asset directory: the cache is an optimisation


def _solid_volume(data):
    """Volume of the solid a mesh bounds.  A closed, consistently oriented surface: the divergence theorem's signed volume.
    Anything else (open at the bottom, duplicated vertices along seams): the voxel grid of the decomposition -- interior cells
    whole, the cells the surface passes through half (counted whole they overestimate a thin or small solid by the shell's
    thickness, and a mesh near the reference's 75 % rule would be classed convex where the reference decomposes it)."""
    from . import acd

    p = np.asarray(data.positions, dtype=np.float64)
    t = np.asarray(data.indices, dtype=np.int64).reshape(-1, 3)
    # closed and oriented: every directed edge has its opposite exactly once
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    n = int(p.shape[0])
    fwd = e[:, 0] * n + e[:, 1]
    bwd = e[:, 1] * n + e[:, 0]
    uf, cf = np.unique(fwd, return_counts=True)
    signed = None
    if cf.max(initial=0) == 1 and np.array_equal(uf, np.unique(bwd)):
        a, b, c = p[t[:, 0]], p[t[:, 1]], p[t[:, 2]]
        signed = abs(float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6.0)
        if _connected_components(t, n) == 1:
            return signed
        # several closed shells (a mug's body and its handle, an assembly): where they intersect the signed volumes count the
        # overlap twice -- the voxel estimate below does not; the smaller of the two stands
    occ, _, h = acd.voxelize(data.positions, data.indices, resolution=100000)
    idx = np.argwhere(occ)
    shell = len(acd._boundary(idx, occ.shape))
    vox = (float(len(idx)) - 0.5 * float(shell)) * h ** 3
    return vox if signed is None else min(signed, vox)


def _connected_components(tris, n_vertices):
    """Number of connected components of a triangle mesh's vertex graph (vertices no triangle uses do not count)."""
    parent = np.arange(n_vertices)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b, c in tris:
        ra, rb, rc = find(a), find(b), find(c)
        r = min(ra, rb, rc)
        parent[ra] = parent[rb] = parent[rc] = r
    used = np.unique(tris)
    return len({find(int(v)) for v in used})


def _compute_hulls(data, force):
    """Mesh::loadPhysics' procedure (mesh.cpp:335-470) on the in-tree decomposition: the single hull when it is forced
    (PHYSICS_FORCE_CONVEX_HULL), when the mesh has no volume (mesh.cpp:373-378: the raw vertices' hull), or when the
    decomposition would keep 75 % or more of the hull's volume (mesh.cpp:426-429)."""
    from . import acd

    single = _reduce(data.positions)
    if force:
        return [single]
    hv = single.volume()
    if hv < 1e-9:
        return [single]
    # the parts of a decomposition cover the solid: when the solid itself fills 75 % of its hull, the rule's answer is known
    # before any cut is made
    if _solid_volume(data) >= 0.80 * hv:
        return [single]
    parts = [Hull(v, t) for v, t in acd.decompose(data.positions, data.indices)]
    if len(parts) <= 1 or sum(p.volume() for p in parts) / hv >= 0.75:
        return [single]
    return parts


def hulls_for_mesh(mesh, use_cache=True):
    from .mesh import Mesh

    data = mesh._data
    force = bool(mesh._flags & Mesh.Flag.PHYSICS_FORCE_CONVEX_HULL)
    fixture = mesh._filename + ".hulls.npz"   # decompositions made with the reference's V-HACD (oracle/ref_build/gen_hulls.py)
    if not force and os.path.exists(fixture):
        z = np.load(fixture)
        # a fixture belongs to ONE geometry: stale after an asset edit (files without digests predate the check)
        if "digests" not in z.files or tuple(int(x) for x in z["digests"]) == _mesh_digests(data):
            n = int(z["n_hulls"])
            return [Hull(z["v%d" % i], z["t%d" % i]) for i in range(n)]
    on_disk = use_cache and "://" not in mesh._filename and os.path.exists(mesh._filename)   # primitive:// etc.: no cache (mesh.cpp:323)
    cache_file = mesh._filename + CACHE_SUFFIX
    if on_disk:
        cached = read_cache(cache_file, mesh._filename, data, mesh._flags)
        if cached is not None:
            return cached
    hulls = _compute_hulls(data, force)
    if on_disk:
        write_cache(cache_file, data, mesh._flags, hulls)
    return hulls
