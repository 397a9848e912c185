"""Host-side assembly of the binary scene description consumed by slhip_render (and, in the
tests, by the CPU oracle): mesh pool, per-scene camera/lights, draw list, raster chunks.

Mirrors what RenderPass::render uploads as uniforms per drawable
(reference src/render_pass.cpp:534-621, src/shaders/render_shader.cpp:233-265, :326-417)."""
import numpy as np

from . import _abi
from . import _math as M
from ._math import f32


def mip_down(img):
    """Next mip level of an RGBA8 image [H,W,4]."""
    h, w = img.shape[0], img.shape[1]
    nh, nw = max(1, h >> 1), max(1, w >> 1)
    ys0 = np.minimum(2 * np.arange(nh), h - 1)
    ys1 = np.minimum(2 * np.arange(nh) + 1, h - 1)
    xs0 = np.minimum(2 * np.arange(nw), w - 1)
    xs1 = np.minimum(2 * np.arange(nw) + 1, w - 1)
    s = (img[ys0][:, xs0].astype(np.uint16) + img[ys0][:, xs1] + img[ys1][:, xs0] + img[ys1][:, xs1] + 2) >> 2
    return s.astype(np.uint8)


class PoolSlot:
    __slots__ = ("vtx_base", "idx_base", "tex_offsets", "tex_sizes", "tex_alpha", "tex_samplers", "version", "n_vertices",
                 "n_indices")


class HostPool:
    """Growing structure-of-arrays mesh pool (host copy).  Slot 0 is the background plane of
    the reference (Primitives::planeSolid as a 2-triangle strip, render_pass.cpp:262,581)."""

    def __init__(self):
        self.pos = [np.array([[1, -1, 0, 1], [1, 1, 0, 1], [-1, -1, 0, 1], [-1, 1, 0, 1]], np.float32)]
        self.nrm = [np.array([[0, 0, 1, 0]] * 4, np.float32)]
        self.uv = [np.array([[1, 0], [1, 1], [0, 0], [0, 1]], np.float32)]
        self.col = [np.ones((4, 4), np.float32)]
        self.tan = [np.array([[1, 0, 0, 1]] * 4, np.float32)]
        self.idx = [np.array([0, 1, 2, 2, 1, 3], np.uint32)]
        self.tex = []
        self.n_vertices = 4
        self.n_indices = 6
        self.n_tex_bytes = 0
        self.dirty = True
        self._textures = {}  # (id(array), mips) -> (offset, w, h); _tex_refs pins the arrays so that ids stay unique
        self._tex_refs = []

    def add_texture(self, rgba, mips=True):
        """Appends an RGBA8 image (row 0 = top) and, for 2D textures, its mip chain (include/slhip.h:
        level l+1 = 2x2 box filter of level l, rounded to nearest; max(1, size >> 1) per axis)."""
        key = (id(rgba), bool(mips))
        if key in self._textures:
            return self._textures[key]
        off = self.n_tex_bytes
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        levels = [a]
        while mips and max(levels[-1].shape[0], levels[-1].shape[1]) > 1:
            levels.append(mip_down(levels[-1]))
        if off + sum(lv.size for lv in levels) >= 1 << 32:
            raise RuntimeError("texel pool would exceed 4 GiB (slhip_draw texture offsets are 32 bit): reuse Mesh objects "
                               "instead of loading the same asset again -- every sl.Mesh registers its textures for good")
        for lv in levels:
            self.tex.append(lv.reshape(-1))
            self.n_tex_bytes += lv.size
        self._textures[key] = (off, a.shape[1], a.shape[0])
        self._tex_refs.append(rgba)   # a collected array's id() may be handed to a new one: keep the keyed object alive
        self.dirty = True
        return self._textures[key]

    def register(self, mesh):
        slot = mesh._slot if getattr(mesh, "_slot_pool", None) is self else None
        d = mesh._data
        if slot is not None and slot.version == mesh._version:
            return slot
        nv, ni = len(d.positions), len(d.indices)
        if slot is not None and slot.n_vertices == nv and slot.n_indices == ni:
            # vertex data changed in place (update_positions & friends): rewrite the slices
            self._rewrite(slot, d)
            slot.version = mesh._version
            self.dirty = True
            return slot
        slot = PoolSlot()
        slot.vtx_base, slot.idx_base = self.n_vertices, self.n_indices
        slot.n_vertices, slot.n_indices = nv, ni
        self.pos.append(np.concatenate([d.positions, np.ones((nv, 1), np.float32)], axis=1))
        self.nrm.append(np.concatenate([d.normals, np.zeros((nv, 1), np.float32)], axis=1))
        self.uv.append(d.uvs.astype(np.float32))
        self.col.append(d.colors.astype(np.float32))
        tan = getattr(d, "tangents", None)
        self.tan.append(np.zeros((nv, 4), np.float32) if tan is None else np.ascontiguousarray(tan, dtype=np.float32))
        self.idx.append(d.indices.astype(np.uint32))
        self.n_vertices += nv
        self.n_indices += ni
        slot.tex_offsets, slot.tex_sizes = [], []
        for t in d.textures:
            off, w, h = self.add_texture(t)
            slot.tex_offsets.append(off)
            slot.tex_sizes.append((w, h))
        slot.tex_alpha = list(getattr(d, "_tex_alpha", [False] * len(d.textures)))
        # sampler state of each texture as the asset file gives it (mesh.cpp:656-663); default: repeat, trilinear
        slot.tex_samplers = list(getattr(d, "tex_samplers", None) or [_abi.SAMPLER_DEFAULT] * len(d.textures))
        slot.version = mesh._version
        mesh._slot = slot
        mesh._slot_pool = self
        self.dirty = True
        return slot

    def _rewrite(self, slot, d):
        self._flatten()
        v0, nv = slot.vtx_base, slot.n_vertices
        self.pos[0][v0:v0 + nv, :3] = d.positions
        self.nrm[0][v0:v0 + nv, :3] = d.normals
        self.col[0][v0:v0 + nv] = d.colors

    def _flatten(self):
        for name in ("pos", "nrm", "uv", "col", "tan", "idx", "tex"):
            lst = getattr(self, name)
            if len(lst) > 1:
                setattr(self, name, [np.concatenate(lst)])

    def arrays(self):
        self._flatten()
        tex = self.tex[0] if self.tex else np.zeros(4, np.uint8)
        return self.pos[0], self.nrm[0], self.uv[0], self.col[0], self.idx[0], tex, self.tan[0]


def _effective_material(mat, obj):
    # RenderShader::setMaterial (render_shader.cpp:355-377)
    metallic = f32(0.04) if mat.metallic is None else f32(mat.metallic)
    roughness = f32(0.5) if mat.roughness is None else f32(mat.roughness)
    if obj is not None:
        if obj._metallic >= 0.0:
            metallic = f32(obj._metallic)
        if obj._roughness >= 0.0:
            roughness = f32(obj._roughness)
    return metallic, roughness


def effective_lights(scene):
    """(directions [3,3], colours [3,3], ambient [3]) the renderer uses: with a light map bound, the
    directional lights are the map's (Sun / Light1 / Light2, at most NUM_LIGHTS) and the ambient term is
    off (render_pass.cpp:412-418, RenderShader::setLightMap render_shader.cpp:270-296)."""
    lm = scene._light_map
    if lm is None:
        return (scene._light_directions.detach().cpu().numpy().astype(np.float32),
                scene._light_colors.detach().cpu().numpy().astype(np.float32),
                np.asarray(scene._ambient_light, dtype=np.float32))
    ld = np.zeros((_abi.NUM_LIGHTS, 3), np.float32)
    lc = np.zeros((_abi.NUM_LIGHTS, 3), np.float32)
    for i, (d, c) in enumerate(list(zip(lm.light_directions, lm.light_colors))[:_abi.NUM_LIGHTS]):
        ld[i], lc[i] = d, c
    return ld, lc, np.zeros(3, np.float32)


def scene_record(scene, rec, shadow_mats=None):
    rec["proj"] = scene._projection.reshape(-1)
    w2c = M.inverted_rigid(scene._camera_pose)
    rec["world_to_cam"] = w2c.reshape(-1)
    # camPosition = worldToCam.invertedRigid().translation() (render_shader.cpp:246)
    rec["cam_position"][:3] = M.inverted_rigid(w2c)[:3, 3]
    rec["cam_position"][3] = 1.0
    ld, lc, amb = effective_lights(scene)
    rec["light_dir"][:, :3] = ld
    rec["light_color"][:, :3] = lc
    rec["ambient"][:3] = amb
    rec["light_map"] = 0 if scene._light_map is None else scene._light_map._slot + 1
    rec["manual_exposure"] = scene._manual_exposure
    if shadow_mats is not None:
        for i in range(_abi.NUM_LIGHTS):
            rec["shadow_mat"][i] = shadow_mats[i].reshape(-1)


def mesh_draw_templates(mesh, slot):
    """One slhip_draw per sub-mesh with everything the MESH and its materials determine (RenderShader::setMaterial,
    render_shader.cpp:326-417); scene, prim_base, clip_base, the object's pose and its overrides are filled in per use."""
    out = []
    m2o = mesh._pretransform
    for sm in mesh._data.submeshes:
        mat = mesh._data.materials[sm.material]
        d = np.zeros((), dtype=_abi.DRAW_DTYPE)
        d["mesh_to_object"] = m2o.reshape(-1)
        d["base_color"] = mat.base_color
        d["emissive"][:3] = mat.emissive
        d["alpha_cutoff"] = 0.5  # render_shader.cpp:382
        d["metallic"], d["roughness"] = _effective_material(mat, None)
        d["class_index"] = mesh._class_index
        flags = 0
        if mat.base_texture is not None:
            flags |= _abi.DRAW_HAS_BASE_TEX
            if slot.tex_alpha[mat.base_texture]:
                flags |= _abi.DRAW_ALPHA_TEST
            d["tex_offset"] = slot.tex_offsets[mat.base_texture]
            d["tex_w"], d["tex_h"] = slot.tex_sizes[mat.base_texture]
            d["tex_sampler"][0] = slot.tex_samplers[mat.base_texture]
        for k, (attr, field, bit) in enumerate((("normal_texture", "normal_tex", _abi.DRAW_HAS_NORMAL_TEX),
                                                ("mr_texture", "mr_tex", _abi.DRAW_HAS_MR_TEX),
                                                ("occlusion_texture", "occlusion_tex", _abi.DRAW_HAS_OCCLUSION_TEX),
                                                ("emissive_texture", "emissive_tex", _abi.DRAW_HAS_EMISSIVE_TEX))):
            ti = getattr(mat, attr, None)
            if ti is not None:
                flags |= bit
                d[field] = (slot.tex_offsets[ti],) + tuple(slot.tex_sizes[ti])
                d["tex_sampler"][k + 1] = slot.tex_samplers[ti]
        d["flags"] = flags
        d["vtx_base"] = slot.vtx_base
        d["idx_base"] = slot.idx_base + sm.first_index
        d["n_tris"] = sm.n_indices // 3
        d["n_verts"] = slot.n_vertices
        out.append(d)
    return out


def plane_draw_template(pool, tex):
    """The background plane's draw (render_pass.cpp:545-582) without its pose."""
    d = np.zeros((), dtype=_abi.DRAW_DTYPE)
    flags = _abi.DRAW_NO_VERTEX_ID
    if tex is not None:
        off, w, h = pool.add_texture(tex._rgba)
        d["base_color"] = (1.0, 1.0, 1.0, 1.0)
        d["tex_offset"], d["tex_w"], d["tex_h"] = off, w, h
        d["tex_sampler"][0] = _abi.SAMPLER_DEFAULT
        flags |= _abi.DRAW_HAS_BASE_TEX
    else:
        d["base_color"] = (0.0, 0.8, 0.0, 1.0)
    d["alpha_cutoff"] = 0.5
    d["metallic"], d["roughness"] = 0.04, 0.5
    d["flags"] = flags
    d["vtx_base"], d["idx_base"], d["n_tris"] = 0, 0, 2
    d["n_verts"] = 4
    return d


def object_draws(obj, pool):
    """One slhip_draw per sub-mesh of the object: what RenderShader::setTransformations / setMaterial /
    setClassIndex / setInstanceIndex upload for it (render_pass.cpp:583-621, render_shader.cpp:233-265,
    :326-417).  `scene` and `prim_base` are left to the caller."""
    out = []
    mesh = obj._mesh
    slot = pool.register(mesh)
    m2o = mesh._pretransform
    o2w = obj._pose
    nm = np.zeros((3, 4), np.float32)
    nm[:, :3] = M.normal_matrix(M.mul44(o2w, m2o))
    for d, sm in zip(mesh_draw_templates(mesh, slot), mesh._data.submeshes):
        mat = mesh._data.materials[sm.material]
        d["object_to_world"] = o2w.reshape(-1)
        d["normal_to_world"] = nm.reshape(-1)
        if obj._color is not None and obj._force_color:
            d["base_color"] = obj._color
        d["metallic"], d["roughness"] = _effective_material(mat, obj)
        d["instance_index"] = obj._instance_index
        flags = int(d["flags"]) | (_abi.DRAW_CASTS_SHADOW if obj._casts_shadows else 0)
        st = obj._sticker_texture
        if st is not None and obj._sticker_range is not None:
            # render_pass.cpp:601-606: projection + range per object, the rectangle texture if one is set
            off, w, h = pool.add_texture(st._rgba, mips=False)    # rectangle texture: one level
            flags |= _abi.DRAW_HAS_STICKER
            d["sticker_tex"] = (off, w, h)
            d["sticker_projection"] = obj.sticker_view_projection().reshape(-1)
            r = np.asarray(obj._sticker_range, dtype=np.float32)   # min.x, min.y, max.x, max.y
            d["sticker_range"] = (r[0], r[1], max(f32(1e-6), r[2] - r[0]), max(f32(1e-6), r[3] - r[1]))
        d["flags"] = flags
        out.append(d)
    return out


def build_batch(scenes, pool, predicate=None, with_shadows=True):
    """Returns (scene_records, draw_records, chunk_records) as numpy structured arrays.  Batches without a predicate, sticker
    decals or background images are assembled by the C++ host layer in one call (_host_records.build); the per-scene numpy
    path below covers the rest and produces the same bits (tests/test_host_records.py)."""
    from . import _host_records

    if len(scenes) and _host_records.eligible(scenes, predicate):
        return _host_records.build(scenes, pool, with_shadows)
    return build_batch_per_scene(scenes, pool, predicate, with_shadows)


def build_batch_per_scene(scenes, pool, predicate=None, with_shadows=True):
    from . import _shadow

    n_scenes = len(scenes)
    srec = np.zeros(n_scenes, dtype=_abi.SCENE_DTYPE)
    draws = []
    chunks = []
    for si, scene in enumerate(scenes):
        objs = [o for o in scene._objects if predicate is None or predicate(o)]
        mats = _shadow.shadow_matrices(scene) if with_shadows else None
        scene_record(scene, srec[si], mats)
        if scene._background_image is not None:
            srec[si]["bg_tex"] = pool.add_texture(scene._background_image._rgba, mips=False)
        srec[si]["draw_begin"] = len(draws)
        prim = 0
        # background plane first (render_pass.cpp:545-582)
        sz = scene._background_plane_size
        if float(np.dot(sz, sz)) > 0:
            d = np.zeros((), dtype=_abi.DRAW_DTYPE)
            scaling = np.diag([sz[0] / f32(2.0), sz[1] / f32(2.0), f32(1.0), f32(1.0)]).astype(np.float32)
            o2w = M.mul44(scene._background_plane_pose, scaling)
            d["mesh_to_object"] = np.eye(4, dtype=np.float32).reshape(-1)
            d["object_to_world"] = o2w.reshape(-1)
            nm = np.zeros((3, 4), np.float32)
            nm[:, :3] = M.normal_matrix(o2w)
            d["normal_to_world"] = nm.reshape(-1)
            flags = _abi.DRAW_NO_VERTEX_ID
            tex = scene._background_plane_texture
            if tex is not None:
                off, w, h = pool.add_texture(tex._rgba)
                d["base_color"] = (1.0, 1.0, 1.0, 1.0)
                d["tex_offset"], d["tex_w"], d["tex_h"] = off, w, h
                d["tex_sampler"][0] = _abi.SAMPLER_DEFAULT
                flags |= _abi.DRAW_HAS_BASE_TEX
            else:
                d["base_color"] = (0.0, 0.8, 0.0, 1.0)
            d["alpha_cutoff"] = 0.5
            d["metallic"], d["roughness"] = 0.04, 0.5
            d["flags"] = flags
            d["vtx_base"], d["idx_base"], d["n_tris"], d["prim_base"] = 0, 0, 2, prim
            d["scene"], d["n_verts"] = si, 4
            prim += 2
            draws.append(d)
        for obj in objs:
            for d in object_draws(obj, pool):
                d["prim_base"] = prim
                d["scene"] = si
                prim += int(d["n_tris"])
                draws.append(d)
        srec[si]["draw_end"] = len(draws)
        srec[si]["n_prims"] = prim
        for di in range(int(srec[si]["draw_begin"]), len(draws)):
            nt = int(draws[di]["n_tris"])
            for first in range(0, nt, _abi.CHUNK_TRIS):
                chunks.append((si, di, first, min(_abi.CHUNK_TRIS, nt - first)))
    drec = np.array(draws, dtype=_abi.DRAW_DTYPE) if draws else np.zeros(0, dtype=_abi.DRAW_DTYPE)
    if len(drec):
        drec["clip_base"] = np.concatenate([[0], np.cumsum(drec["n_verts"][:-1], dtype=np.uint64)]).astype(np.uint32)
    crec = np.array(chunks, dtype=_abi.CHUNK_DTYPE) if chunks else np.zeros(0, dtype=_abi.CHUNK_DTYPE)
    return srec, drec, crec
