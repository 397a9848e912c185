"""Mesh file readers producing the *consolidated* representation of the reference:
one concatenated vertex array + one u32 index array (indices global over the file) + a list of
sub-meshes, node transforms baked in (reference src/mesh_tools/consolidate.cpp:51-338).

Supported: glTF 2.0 (.gltf + external buffers/images, .glb), ``primitive://cube``
(reference src/utils/primitive_importer.cpp), Wavefront OBJ (+MTL, map_Kd).
"""
import base64
import io
import json
import os
import struct

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class SubMesh:
    __slots__ = ("first_index", "n_indices", "material")

    def __init__(self, first_index, n_indices, material):
        self.first_index = first_index
        self.n_indices = n_indices
        self.material = material


class Material:
    """PBR metallic-roughness material as RenderShader::setMaterial consumes it
    (reference src/shaders/render_shader.cpp:326-417)."""

    def __init__(self, base_color=(1.0, 1.0, 1.0, 1.0), metallic=None, roughness=None,
                 emissive=(0.0, 0.0, 0.0), base_texture=None, normal_texture=None, mr_texture=None,
                 occlusion_texture=None, emissive_texture=None):
        self.base_color = np.asarray(base_color, dtype=np.float32)
        # None == attribute absent in the file => shader defaults 0.04 / 0.5
        self.metallic = metallic
        self.roughness = roughness
        self.emissive = np.asarray(emissive, dtype=np.float32)
        self.base_texture = base_texture  # index into ConsolidatedMesh.textures or None
        # further inputs of RenderShader::setMaterial (render_shader.cpp:395-415), same indexing
        self.normal_texture = normal_texture
        self.mr_texture = mr_texture              # roughness in G, metallic in B
        self.occlusion_texture = occlusion_texture
        self.emissive_texture = emissive_texture


class ConsolidatedMesh:
    def __init__(self):
        self.positions = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.uvs = np.zeros((0, 2), np.float32)
        self.colors = np.zeros((0, 4), np.float32)
        self.tangents = None   # f32 [V,4] (xyz, bitangent sign) or None = zeros
        self.has_vertex_colors = False
        self.indices = np.zeros((0,), np.uint32)
        self.submeshes = []
        self.materials = []
        self.textures = []  # list of HxWx4 uint8 arrays, row 0 = top of the image
        self.tex_samplers = []  # one SLHIP_SAMPLER_* byte per texture (empty = defaults)


def _smooth_normals(pos, idx):
    tri = idx.reshape(-1, 3)
    p0, p1, p2 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0)
    n = np.zeros_like(pos)
    for k in range(3):
        np.add.at(n, tri[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return (n / ln).astype(np.float32)


def compute_tangents(pos, nrm, uv, idx):
    """mesh_tools::computeTangents (src/mesh_tools/compute_tangents.cpp:26-137): per-triangle tangent /
    bitangent from the UV parametrisation, averaged per vertex, bitangent sign in w.  float32 like the
    reference; a degenerate UV triangle poisons its vertices with inf / nan there as well."""
    nv = len(pos)
    tan = np.zeros((nv, 3), np.float32)
    bit = np.zeros((nv, 3), np.float32)
    deg = np.zeros(nv, np.float32)
    f = idx.reshape(-1, 3).astype(np.int64)
    with np.errstate(all="ignore"):
        d1, d2 = pos[f[:, 1]] - pos[f[:, 0]], pos[f[:, 2]] - pos[f[:, 0]]
        u1, u2 = uv[f[:, 1]] - uv[f[:, 0]], uv[f[:, 2]] - uv[f[:, 0]]
        r = (np.float32(1.0) / (u1[:, 0] * u2[:, 1] - u1[:, 1] * u2[:, 0])).astype(np.float32)
        t = ((d1 * u2[:, 1:2] - d2 * u1[:, 1:2]) * r[:, None]).astype(np.float32)
        b = ((d2 * u1[:, 0:1] - d1 * u2[:, 0:1]) * r[:, None]).astype(np.float32)
        for k in range(3):   # sequential accumulation order = face order, as in the reference loop
            np.add.at(tan, f[:, k], t)
            np.add.at(bit, f[:, k], b)
            np.add.at(deg, f[:, k], 1.0)
        tan = tan / deg[:, None]
        bit = bit / deg[:, None]
        tan = tan / np.sqrt((tan * tan).sum(axis=1, keepdims=True))
        bit = bit / np.sqrt((bit * bit).sum(axis=1, keepdims=True))
        sign = np.sign((np.cross(nrm, tan) * bit).sum(axis=1)).astype(np.float32)
    return np.concatenate([tan.astype(np.float32), sign[:, None]], axis=1).astype(np.float32)


def _gltf_sampler(doc, sampler_id):
    """glTF sampler -> SLHIP_SAMPLER_* byte.  Unspecified filters: linear / linear-mipmap-linear (what Magnum's
    importer hands to mesh.cpp:656-663); wrap defaults to repeat (10497)."""
    from . import _abi

    if sampler_id is None:
        return _abi.SAMPLER_DEFAULT
    sp = doc["samplers"][sampler_id]
    wrap = {10497: 0, 33071: 1, 33648: 2}
    m = wrap.get(sp.get("wrapS", 10497), 0) | (wrap.get(sp.get("wrapT", 10497), 0) << 2)
    if sp.get("magFilter", 9729) == 9729:
        m |= 0x10
    # minFilter: 9728 NEAREST, 9729 LINEAR, 9984 NEAREST_MIPMAP_NEAREST, 9985 LINEAR_MIPMAP_NEAREST,
    #            9986 NEAREST_MIPMAP_LINEAR, 9987 LINEAR_MIPMAP_LINEAR
    mn = sp.get("minFilter", 9987)
    if mn in (9729, 9985, 9987):
        m |= 0x20
    m |= {9728: 0, 9729: 0, 9984: 1, 9985: 1, 9986: 2, 9987: 2}.get(mn, 2) << 6
    return m


def _load_image(data):
    from PIL import Image

    img = Image.open(io.BytesIO(data))
    has_alpha = img.mode in ("RGBA", "LA", "PA") or ("transparency" in img.info)
    arr = np.asarray(img.convert("RGBA"), dtype=np.uint8).copy()
    return arr, has_alpha


# ------------------------------------------------------------------------------------------
# glTF
# ------------------------------------------------------------------------------------------
def _node_matrix(node):
    if "matrix" in node:
        return np.asarray(node["matrix"], dtype=np.float64).reshape(4, 4).T  # column-major in file
    m = np.eye(4)
    if "scale" in node:
        m = np.diag(list(node["scale"]) + [1.0]) @ m
    if "rotation" in node:
        x, y, z, w = node["rotation"]
        r = np.array([
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0],
            [0, 0, 0, 1],
        ])
        m = r @ m
    if "translation" in node:
        t = np.eye(4)
        t[:3, 3] = node["translation"]
        m = t @ m
    return m


def load_gltf(path):
    path = str(path)
    base_dir = os.path.dirname(path)
    with open(path, "rb") as f:
        raw = f.read()
    bin_chunk = None
    if raw[:4] == b"glTF":
        _, _, length = struct.unpack("<III", raw[:12])
        off = 12
        doc = None
        while off < length:
            clen, ctype = struct.unpack("<II", raw[off:off + 8])
            cdata = raw[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                doc = json.loads(cdata.decode("utf-8"))
            elif ctype == 0x004E4942:
                bin_chunk = cdata
            off += 8 + clen
    else:
        doc = json.loads(raw.decode("utf-8"))

    def read_uri(uri):
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        with open(os.path.join(base_dir, uri), "rb") as f:
            return f.read()

    buffers = []
    for b in doc.get("buffers", []):
        buffers.append(read_uri(b["uri"]) if "uri" in b else bin_chunk)

    def accessor(i):
        acc = doc["accessors"][i]
        dt = np.dtype(_COMPONENT[acc["componentType"]])
        nc = _NCOMP[acc["type"]]
        count = acc["count"]
        if "bufferView" not in acc:
            return np.zeros((count, nc), dt)
        bv = doc["bufferViews"][acc["bufferView"]]
        buf = buffers[bv["buffer"]]
        start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        if stride == dt.itemsize * nc:
            arr = np.frombuffer(buf, dtype=dt, count=count * nc, offset=start).reshape(count, nc)
        else:
            arr = np.lib.stride_tricks.as_strided(
                np.frombuffer(buf, dtype=dt, offset=start, count=((count - 1) * stride) // dt.itemsize + nc),
                shape=(count, nc), strides=(stride, dt.itemsize))
        arr = np.array(arr)
        if acc.get("normalized", False) and dt.kind in "iu":
            arr = arr.astype(np.float32) / float(np.iinfo(dt).max)
        return arr

    out = ConsolidatedMesh()

    # textures/images
    image_cache = {}

    def texture_index(tex_id):
        src = doc["textures"][tex_id].get("source")
        if src is None:
            return None, False
        if src not in image_cache:
            img = doc["images"][src]
            if "uri" in img:
                data = read_uri(img["uri"])
            else:
                bv = doc["bufferViews"][img["bufferView"]]
                b = buffers[bv["buffer"]]
                data = b[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
            arr, has_alpha = _load_image(data)
            out.textures.append(arr)
            out.tex_samplers.append(_gltf_sampler(doc, doc["textures"][tex_id].get("sampler")))
            image_cache[src] = (len(out.textures) - 1, has_alpha)
        return image_cache[src]

    mat_cache = {}

    def material_index(mid):
        if mid in mat_cache:
            return mat_cache[mid]
        if mid is None:
            # Magnum falls back to a default material (reference src/context.cpp:382-384)
            m = Material(base_color=_srgba(0x3bd267ff))
        else:
            md = doc["materials"][mid]
            pbr = md.get("pbrMetallicRoughness", {})
            tex = None
            if "baseColorTexture" in pbr:
                tex, _ = texture_index(pbr["baseColorTexture"]["index"])
            # Magnum's glTF importer drops attributes that have the glTF default value (1.0),
            # so RenderShader::setMaterial then uses ITS defaults 0.04 / 0.5
            # (render_shader.cpp:355-369)
            metallic = pbr.get("metallicFactor", 1.0)
            roughness = pbr.get("roughnessFactor", 1.0)
            has_mr_tex = "metallicRoughnessTexture" in pbr

            def opt_tex(container, key):
                return texture_index(container[key]["index"])[0] if key in container else None

            m = Material(
                normal_texture=opt_tex(md, "normalTexture"), mr_texture=opt_tex(pbr, "metallicRoughnessTexture"),
                occlusion_texture=opt_tex(md, "occlusionTexture"), emissive_texture=opt_tex(md, "emissiveTexture"),
                base_color=pbr.get("baseColorFactor", (1.0, 1.0, 1.0, 1.0)),
                metallic=None if (metallic == 1.0 and not has_mr_tex) else metallic,
                roughness=None if (roughness == 1.0 and not has_mr_tex) else roughness,
                emissive=md.get("emissiveFactor", (0.0, 0.0, 0.0)),
                base_texture=tex,
            )
            if has_mr_tex:
                if m.metallic is None:
                    m.metallic = 1.0
                if m.roughness is None:
                    m.roughness = 1.0
        out.materials.append(m)
        mat_cache[mid] = len(out.materials) - 1
        return mat_cache[mid]

    pos_l, nrm_l, uv_l, col_l, idx_l, tan_l = [], [], [], [], [], []
    v_off = 0
    i_off = 0

    def add_primitive(prim, transform):
        nonlocal v_off, i_off
        if prim.get("mode", 4) != 4:
            return
        attrs = prim["attributes"]
        pos = accessor(attrs["POSITION"]).astype(np.float32)
        n = len(pos)
        if "indices" in prim:
            idx = accessor(prim["indices"]).astype(np.uint32).reshape(-1)
        else:
            idx = np.arange(n, dtype=np.uint32)
        if "NORMAL" in attrs:
            nrm = accessor(attrs["NORMAL"]).astype(np.float32)
        else:
            nrm = _smooth_normals(pos, idx)
        uv = accessor(attrs["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in attrs else np.zeros((n, 2), np.float32)
        if "COLOR_0" in attrs:
            c = accessor(attrs["COLOR_0"]).astype(np.float32)
            if c.shape[1] == 3:
                c = np.concatenate([c, np.ones((n, 1), np.float32)], axis=1)
            out.has_vertex_colors = True
        else:
            c = np.ones((n, 4), np.float32)
        # tangents: the file's, else computed from the UVs (consolidate.cpp:90-94); a mesh without UVs gets zeros
        if "TANGENT" in attrs:
            tan = accessor(attrs["TANGENT"]).astype(np.float32)
        elif "TEXCOORD_0" in attrs:
            tan = compute_tangents(pos, nrm, uv, idx)
        else:
            tan = np.zeros((n, 4), np.float32)
        T = transform.astype(np.float32)
        # transformPoint / transformVector of the baked node transform (consolidate.cpp:252-294)
        pos_t = (pos @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        nrm_t = (nrm @ T[:3, :3].T).astype(np.float32)
        # quirk kept: consolidate.cpp:275-279 writes Vector4{transformVector(tangent.xyz), 1.0f} -- the bitangent sign is lost
        tan_l.append(np.concatenate([(tan[:, :3] @ T[:3, :3].T).astype(np.float32), np.ones((n, 1), np.float32)], axis=1))
        pos_l.append(pos_t)
        nrm_l.append(nrm_t)
        uv_l.append(uv)
        col_l.append(c)
        idx_l.append(idx + np.uint32(v_off))
        out.submeshes.append(SubMesh(i_off, len(idx), material_index(prim.get("material"))))
        v_off += n
        i_off += len(idx)

    def recurse(node_id, parent):
        node = doc["nodes"][node_id]
        T = parent @ _node_matrix(node)
        if "mesh" in node:
            for prim in doc["meshes"][node["mesh"]]["primitives"]:
                add_primitive(prim, T)
        for ch in node.get("children", []):
            recurse(ch, T)

    scene_id = doc.get("scene", 0)
    scenes = doc.get("scenes")
    roots = scenes[scene_id]["nodes"] if scenes else list(range(len(doc.get("nodes", []))))
    for r in roots:
        recurse(r, np.eye(4))

    if not pos_l:
        raise RuntimeError("no triangle meshes in %s" % path)
    out.positions = np.concatenate(pos_l)
    out.normals = np.concatenate(nrm_l)
    out.uvs = np.concatenate(uv_l)
    out.colors = np.concatenate(col_l)
    out.tangents = np.concatenate(tan_l)
    out.indices = np.concatenate(idx_l)
    out._tex_alpha = [a for (_, a) in sorted(image_cache.values())]
    return out


def _srgba(rgba):
    """Magnum's ``0x..._srgbaf`` literal: 8-bit sRGB + alpha -> linear float (piecewise curve)."""
    c = np.array([(rgba >> 24) & 255, (rgba >> 16) & 255, (rgba >> 8) & 255, rgba & 255], np.float32) / 255.0
    c[:3] = np.where(c[:3] <= 0.04045, c[:3] / 12.92, np.power((c[:3] + 0.055) / 1.055, 2.4))
    return c.astype(np.float32)


# ------------------------------------------------------------------------------------------
# primitive://cube  (Magnum::Primitives::cubeSolid: 24 vertices, 12 triangles, +-1)
# ------------------------------------------------------------------------------------------
def load_primitive(name):
    if name != "cube":
        raise ValueError("Unknown primitive %s" % name)
    faces = [
        ((0, 0, 1), (1, 0, 0), (0, 1, 0)),    # +Z
        ((1, 0, 0), (0, 0, -1), (0, 1, 0)),   # +X
        ((0, 0, -1), (-1, 0, 0), (0, 1, 0)),  # -Z
        ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),   # -X
        ((0, 1, 0), (1, 0, 0), (0, 0, -1)),   # +Y
        ((0, -1, 0), (1, 0, 0), (0, 0, 1)),   # -Y
    ]
    pos, nrm, uv, idx = [], [], [], []
    for n, u, v in faces:
        n, u, v = np.array(n, np.float32), np.array(u, np.float32), np.array(v, np.float32)
        b = len(pos)
        for (su, sv) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append(n + su * u + sv * v)
            nrm.append(n)
            uv.append(((su + 1) / 2, (sv + 1) / 2))
        idx += [b, b + 1, b + 2, b, b + 2, b + 3]
    out = ConsolidatedMesh()
    out.positions = np.array(pos, np.float32)
    out.normals = np.array(nrm, np.float32)
    out.uvs = np.array(uv, np.float32)
    out.colors = np.ones((24, 4), np.float32)
    out.indices = np.array(idx, np.uint32)
    out.materials = [Material(base_color=_srgba(0x3bd267ff))]
    out.submeshes = [SubMesh(0, 36, 0)]
    out._tex_alpha = []
    return out


# ------------------------------------------------------------------------------------------
# Wavefront OBJ (the YCB `textured.obj` flavour: v / vt / vn / f, one mtl with map_Kd)
# ------------------------------------------------------------------------------------------
def load_obj(path):
    path = str(path)
    base_dir = os.path.dirname(path)
    vs, vts, vns = [], [], []
    groups = []  # (material name, list of corner-key triples)
    cur = [None, []]
    mtllibs = []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                vs.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vt":
                vts.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif t[0] == "vn":
                vns.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "f":
                corners = []
                for c in t[1:]:
                    p = c.split("/")
                    vi = int(p[0])
                    ti = int(p[1]) if len(p) > 1 and p[1] else 0
                    ni = int(p[2]) if len(p) > 2 and p[2] else 0
                    vi = vi - 1 if vi > 0 else len(vs) + vi
                    ti = ti - 1 if ti > 0 else (len(vts) + ti if ti < 0 else -1)
                    ni = ni - 1 if ni > 0 else (len(vns) + ni if ni < 0 else -1)
                    corners.append((vi, ti, ni))
                for k in range(1, len(corners) - 1):
                    cur[1].append((corners[0], corners[k], corners[k + 1]))
            elif t[0] == "usemtl":
                if cur[1]:
                    groups.append(tuple(cur))
                cur = [t[1], []]
            elif t[0] == "mtllib":
                mtllibs.append(" ".join(t[1:]))
    if cur[1]:
        groups.append(tuple(cur))
    out = ConsolidatedMesh()
    mats = {}
    for lib in mtllibs:
        p = os.path.join(base_dir, lib)
        if not os.path.exists(p):
            continue
        name = None
        with open(p, "r", errors="replace") as f:
            for line in f:
                t = line.split()
                if not t:
                    continue
                if t[0] == "newmtl":
                    name = t[1]
                    mats[name] = {"Kd": (1.0, 1.0, 1.0), "map_Kd": None}
                elif name and t[0] == "Kd":
                    mats[name]["Kd"] = tuple(float(x) for x in t[1:4])
                elif name and t[0] == "map_Kd":
                    mats[name]["map_Kd"] = t[-1]
    vs = np.array(vs, np.float32).reshape(-1, 3)
    vts = np.array(vts, np.float32).reshape(-1, 2)
    vns = np.array(vns, np.float32).reshape(-1, 3)
    key_to_index = {}
    pos, uv, nrm, idx = [], [], [], []
    tex_alpha = []
    i_off = 0
    for mname, tris in groups:
        md = mats.get(mname, {"Kd": (1.0, 1.0, 1.0), "map_Kd": None})
        tex = None
        if md["map_Kd"]:
            tp = os.path.join(base_dir, md["map_Kd"])
            if os.path.exists(tp):
                with open(tp, "rb") as f:
                    arr, has_alpha = _load_image(f.read())
                out.textures.append(arr)
                tex_alpha.append(has_alpha)
                tex = len(out.textures) - 1
        # Assimp-imported materials carry a diffuse colour/texture; the reference maps
        # DiffuseTexture to the base colour slot (render_shader.cpp:432-433)
        base = (1.0, 1.0, 1.0, 1.0) if tex is not None else tuple(md["Kd"]) + (1.0,)
        out.materials.append(Material(base_color=base, base_texture=tex))
        start = len(idx)
        for tri in tris:
            for key in tri:
                j = key_to_index.get(key)
                if j is None:  # aiProcess_JoinIdenticalVertices
                    j = len(pos)
                    key_to_index[key] = j
                    pos.append(vs[key[0]])
                    uv.append(vts[key[1]] if key[1] >= 0 and len(vts) else (0.0, 0.0))
                    nrm.append(vns[key[2]] if key[2] >= 0 and len(vns) else (0.0, 0.0, 0.0))
                idx.append(j)
        out.submeshes.append(SubMesh(start, len(idx) - start, len(out.materials) - 1))
        i_off = len(idx)
    del i_off
    out.positions = np.array(pos, np.float32).reshape(-1, 3)
    out.uvs = np.array(uv, np.float32).reshape(-1, 2)
    out.uvs[:, 1] = 1.0 - out.uvs[:, 1]  # OBJ v is bottom-up; our textures are stored top-down
    out.normals = np.array(nrm, np.float32).reshape(-1, 3)
    out.indices = np.array(idx, np.uint32)
    if not len(vns):
        out.normals = _smooth_normals(out.positions, out.indices)  # aiProcess_GenSmoothNormals
    out.colors = np.ones((len(out.positions), 4), np.float32)
    out._tex_alpha = tex_alpha
    return out


def load_any(filename):
    filename = str(filename)
    if filename.startswith("primitive://"):
        return load_primitive(filename[len("primitive://"):])
    ext = os.path.splitext(filename)[1].lower()
    if ext in (".gltf", ".glb"):
        return load_gltf(filename)
    if ext == ".obj":
        return load_obj(filename)
    raise RuntimeError("Unsupported mesh format: %s (supported: .gltf .glb .obj primitive://cube)" % filename)
