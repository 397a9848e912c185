"""Convex collision shapes for a mesh (reference Mesh::loadPhysics, src/mesh.cpp:304-533).

The reference runs V-HACD twice (single hull, then concavity 0.002) and uses the decomposition
only when its volume is < 75 % of the hull's (mesh.cpp:426-429).  Here:
  * `lib/libslvhacd.so` -- the SAME library (the reference's vendored third-party V-HACD, built by
    `__graft_entry__.build()` from the sources where they lie, behind our C shim
    csrc/host/slvhacd.cpp) run with the same two parameter sets and the same selection rule: this
    is the decomposition behind `hulls_for_mesh` whenever the library is present;
  * `<mesh>.sl_hulls` caches the result with the reference's invalidation keys (below);
  * `<mesh>.hulls.npz` fixtures (tests/fixtures: cube, bunny) carry hulls made by the same library
    for boxes without it; they are validated against the mesh's geometry digests;
  * without the library: the exact single hull (scipy/Qhull) and, for concave meshes, a built-in
    approximate splitter -- with a RuntimeWarning, because those parts are NOT V-HACD's.
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


_VHACD = None
VHACD_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libslvhacd.so")


def vhacd_lib():
    """ctypes handle of lib/libslvhacd.so, or None when it was not built (no /root/reference at build time)."""
    global _VHACD
    if _VHACD is None:
        import ctypes as C

        if not os.path.exists(VHACD_LIB):
            _VHACD = False
        else:
            L = C.CDLL(VHACD_LIB)
            L.slvhacd_decompose.restype = C.c_void_p
            L.slvhacd_decompose.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
            L.slvhacd_free.argtypes = [C.c_void_p]
            L.slvhacd_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.slvhacd_hull_size.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
            L.slvhacd_hull_copy.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
            _VHACD = L
    return _VHACD or None


def vhacd_hulls(positions, indices, force_single=False):
    """Mesh::loadPhysics' two V-HACD passes + selection rule (mesh.cpp:335-470) through lib/libslvhacd.so.
    Returns (hulls, info) with info = dict(volume_single, volume_decomposition, used_decomposition), or None
    when V-HACD could not build a hull with volume (the reference then hands the raw vertices to PhysX,
    mesh.cpp:373-378 -- the caller falls back to the Qhull hull of the vertices)."""
    import ctypes as C

    L = vhacd_lib()
    if L is None:
        raise RuntimeError("lib/libslvhacd.so is not built")
    v = np.ascontiguousarray(positions, dtype=np.float32)
    t = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    r = L.slvhacd_decompose(v.ctypes.data, len(v), t.ctypes.data, len(t) // 3, 1 if force_single else 0)   # releases the GIL
    if not r:
        raise RuntimeError("V-HACD failed")
    try:
        info = (C.c_uint32 * 3)()
        vol = (C.c_double * 2)()
        L.slvhacd_info(r, info, vol)
        if info[2]:
            return None
        hulls = []
        for i in range(info[0]):
            nv, nt, hv = C.c_uint32(), C.c_uint32(), C.c_double()
            L.slvhacd_hull_size(r, i, C.byref(nv), C.byref(nt), C.byref(hv))
            hv_ = np.empty((nv.value, 3), np.float32)
            ht_ = np.empty((nt.value, 3), np.uint32)
            L.slvhacd_hull_copy(r, i, hv_.ctypes.data, ht_.ctypes.data)
            hulls.append(Hull(hv_, ht_.astype(np.int32)))
        return hulls, {"volume_single": vol[0], "volume_decomposition": vol[1], "used_decomposition": bool(info[1])}
    finally:
        L.slvhacd_free(r)


_warned_no_vhacd = False


def _compute_hulls(data, force):
    global _warned_no_vhacd
    if vhacd_lib() is not None:
        res = vhacd_hulls(data.positions, data.indices, force_single=force)
        if res is None:
            return [_reduce(data.positions)]      # mesh.cpp:373-378: raw vertices -> the cooker's own hull
        return res[0]
    single = _reduce(data.positions)
    if force:
        return [single]
    tris = data.indices.reshape(-1, 3).astype(np.int64)
    hv = single.volume()
    if hv < 1e-9:
        return [single]
    parts = _decompose(data.positions, tris, 0)
    # the reference's criterion on what we have: sum of the parts' hull volumes against the single hull (mesh.cpp:426-429)
    if len(parts) <= 1 or sum(p.volume() for p in parts) / hv >= 0.75:
        return [single]
    if not _warned_no_vhacd:
        import warnings

        warnings.warn("lib/libslvhacd.so is not built (run __graft_entry__.build() where /root/reference is present): concave "
                      "meshes are split by the built-in approximate splitter, NOT by V-HACD as the reference does", RuntimeWarning)
        _warned_no_vhacd = True
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
