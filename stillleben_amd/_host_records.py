"""Batch assembly of the render records of the per-object API through the C++ host layer (csrc/slhip_records.cpp): the scenes
and objects of a batch are flattened into slhip_host_scene / slhip_host_object descriptors (a few attribute sweeps over the
Python objects), the per-mesh draw templates are cached in the mesh pool, and ONE call to slhip_records_build_render fills the
slhip_scene / slhip_draw / slhip_chunk records -- shadow matrices, normal matrices, per-object overrides -- that
`_batch.build_batch` assembles scene by scene in numpy (kept for the cases below and as the cross-check of the tests).

Not covered (build_batch handles them): a `predicate`, sticker decals, background images."""
import ctypes as C

import numpy as np

from . import _abi
from ._math import f32


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


def eligible(scenes, predicate):
    if predicate is not None:
        return False
    for s in scenes:
        if s._background_image is not None:
            return False
        for o in s._objects:
            if o._sticker_texture is not None and o._sticker_range is not None:
                return False
    return True


class TemplateTable:
    """Draw templates (slhip_draw rows with everything a mesh and its materials determine) of every mesh seen so far, and of
    the background planes in use; lives with a HostPool."""

    def __init__(self, pool):
        self.pool = pool
        self.rows = []
        self.index = {}          # mesh key -> (begin, count, bbox4)
        self.plane = {}          # id(texture) or None -> row index
        self._keep = []
        self._array = None

    def mesh(self, mesh):
        from ._batch import mesh_draw_templates

        slot = self.pool.register(mesh)
        key = (id(mesh), mesh._version, mesh._class_index, mesh._pretransform.tobytes())
        hit = self.index.get(key)
        if hit is None:
            rows = mesh_draw_templates(mesh, slot)
            bbox = mesh.bbox
            b4 = np.zeros(4, np.float32)
            b4[:3] = bbox.np_center()
            b4[3] = bbox.np_diagonal() / f32(2.0)
            hit = (len(self.rows), len(rows), b4)
            self.rows.extend(rows)
            self.index[key] = hit
            self._keep.append(mesh)
            self._array = None
        return hit

    def plane_row(self, texture):
        from ._batch import plane_draw_template

        key = None if texture is None else id(texture)
        hit = self.plane.get(key)
        if hit is None:
            hit = len(self.rows)
            self.rows.append(plane_draw_template(self.pool, texture))
            self.plane[key] = hit
            if texture is not None:
                self._keep.append(texture)
            self._array = None
        return hit

    def array(self):
        if self._array is None:
            self._array = np.array(self.rows, dtype=_abi.DRAW_DTYPE) if self.rows else np.zeros(0, _abi.DRAW_DTYPE)
        return self._array


_PROJ_INV = {}


def _proj_inv(P):
    key = P.tobytes()
    r = _PROJ_INV.get(key)
    if r is None:
        if len(_PROJ_INV) > 64:
            _PROJ_INV.clear()
        r = _PROJ_INV[key] = np.linalg.inv(P.astype(np.float64)).astype(np.float32)
    return r


def describe(scenes, pool):
    """(host_scenes, host_objects, templates) of a batch."""
    from ._batch import effective_lights

    table = pool.__dict__.get("_templates")
    if table is None:
        table = pool._templates = TemplateTable(pool)
    objs = [o for s in scenes for o in s._objects]
    n = len(objs)
    ho = np.zeros(n, dtype=_abi.HOST_OBJECT_DTYPE)
    if n:
        per_mesh = {}
        tm = []
        for o in objs:
            m = o._mesh
            h = per_mesh.get(id(m))
            if h is None:
                h = per_mesh[id(m)] = table.mesh(m)
            tm.append(h)
        ho["pose"] = np.stack([o._pose for o in objs]).reshape(n, 16)
        ho["tmpl_begin"] = np.fromiter((h[0] for h in tm), np.uint32, n)
        ho["tmpl_count"] = np.fromiter((h[1] for h in tm), np.uint32, n)
        ho["bbox_center"] = np.stack([h[2] for h in tm])
        ho["instance_index"] = np.fromiter((o._instance_index for o in objs), np.uint32, n)
        ho["metallic"] = np.fromiter((o._metallic for o in objs), np.float32, n)
        ho["roughness"] = np.fromiter((o._roughness for o in objs), np.float32, n)
        ho["casts_shadows"] = np.fromiter((o._casts_shadows for o in objs), np.uint32, n)
        for k, o in enumerate(objs):
            if o._color is not None and o._force_color:
                ho["color"][k] = o._color
                ho["force_color"][k] = 1
    hs = np.zeros(len(scenes), dtype=_abi.HOST_SCENE_DTYPE)
    k = 0
    for i, s in enumerate(scenes):
        r = hs[i]
        r["proj"] = s._projection.reshape(-1)
        r["proj_inv"] = _proj_inv(s._projection).reshape(-1)
        r["camera_pose"] = s._camera_pose.reshape(-1)
        ld, lc, amb = effective_lights(s)
        r["light_dir"][:, :3] = ld
        r["light_color"][:, :3] = lc
        r["ambient"][:3] = amb
        sz = s._background_plane_size
        r["plane_size"] = sz
        r["plane_pose"] = s._background_plane_pose.reshape(-1)
        r["plane_template"] = table.plane_row(s._background_plane_texture) if float(np.dot(sz, sz)) > 0 else -1
        r["manual_exposure"] = s._manual_exposure
        r["obj_begin"] = k
        k += len(s._objects)
        r["obj_end"] = k
        r["light_map"] = 0 if s._light_map is None else s._light_map._slot + 1
    return hs, ho, table.array()


def build(scenes, pool, with_shadows=True):
    """(srec, drec, crec) of the batch: what _batch.build_batch returns, assembled in C++."""
    hs, ho, tmpl = describe(scenes, pool)
    return build_from(hs, ho, tmpl, with_shadows)


def build_from(hs, ho, tmpl, with_shadows=True):
    """slhip_records_build_render on prepared descriptors (slhip_host_scene / slhip_host_object arrays + draw templates)."""
    L = _abi.lib()
    hs, ho = np.ascontiguousarray(hs), np.ascontiguousarray(ho)
    nd, nc = C.c_uint32(), C.c_uint32()
    _abi.check(L.slhip_records_count(_ptr(hs), len(hs), _ptr(ho), _ptr(tmpl), C.byref(nd), C.byref(nc)), "slhip_records_count")
    srec = np.empty(len(hs), dtype=_abi.SCENE_DTYPE)
    drec = np.empty(nd.value, dtype=_abi.DRAW_DTYPE)
    crec = np.empty(nc.value, dtype=_abi.CHUNK_DTYPE)
    _abi.check(L.slhip_records_build_render(_ptr(hs), len(hs), _ptr(ho), _ptr(tmpl), 1 if with_shadows else 0, _ptr(srec), _ptr(drec),
                                            nd.value, _ptr(crec), nc.value), "slhip_records_build_render")
    return srec, drec, crec


def shadow_matrices(scene):
    """The NUM_LIGHTS shadow matrices of ONE scene (render_pass.cpp:69-211) -- the same C++ the batch path runs."""
    L = _abi.lib()
    hs = np.zeros(1, dtype=_abi.HOST_SCENE_DTYPE)
    from ._batch import effective_lights

    r = hs[0]
    r["proj"] = scene._projection.reshape(-1)
    r["proj_inv"] = _proj_inv(scene._projection).reshape(-1)
    r["camera_pose"] = scene._camera_pose.reshape(-1)
    ld, lc, _ = effective_lights(scene)
    r["light_dir"][:, :3] = ld
    r["light_color"][:, :3] = lc
    objs = scene._objects
    ho = np.zeros(len(objs), dtype=_abi.HOST_OBJECT_DTYPE)
    for k, o in enumerate(objs):
        ho["pose"][k] = o._pose.reshape(-1)
        bbox = o._mesh.bbox
        ho["bbox_center"][k, :3] = bbox.np_center()
        ho["bbox_center"][k, 3] = bbox.np_diagonal() / f32(2.0)
    r["obj_end"] = len(objs)
    out = np.zeros((_abi.NUM_LIGHTS, 4, 4), np.float32)
    _abi.check(L.slhip_host_shadow_matrices(_ptr(hs), _ptr(ho), _ptr(out)), "slhip_host_shadow_matrices")
    return [out[i] for i in range(_abi.NUM_LIGHTS)]
